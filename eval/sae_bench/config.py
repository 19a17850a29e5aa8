"""Import-path shim (eval/sae_bench/config.py)."""
from matryoshka_transcoder_amd.sae_bench import AutoInterpSAEConfig, NPZTranscoderSpec  # noqa: F401
