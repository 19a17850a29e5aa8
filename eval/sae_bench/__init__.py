"""Import-path shim for the reference's SAEBench adapters (eval/sae_bench/{wrappers,config}.py)."""
from matryoshka_transcoder_amd.sae_bench import (  # noqa: F401
    AutoInterpSAEConfig,
    MatryoshkaAutoInterpAdapter,
    NPZTranscoderAdapter,
    NPZTranscoderSpec,
    ReconstructionReport,
    build_matryoshka_adapter,
    build_npz_transcoder_adapter,
)
