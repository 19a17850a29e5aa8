"""Import-path shim for the reference's activation-store module."""
from matryoshka_transcoder_amd.activation_store import (  # noqa: F401
    SyntheticSource, TensorPairSource, TranscoderActivationsStore, create_transcoder_config)
