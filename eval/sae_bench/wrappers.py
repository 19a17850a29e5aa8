"""Import-path shim (eval/sae_bench/wrappers.py)."""
from matryoshka_transcoder_amd.sae_bench import (  # noqa: F401
    MatryoshkaAutoInterpAdapter,
    NPZTranscoderAdapter,
    ReconstructionReport,
    build_matryoshka_adapter,
    build_npz_transcoder_adapter,
)
