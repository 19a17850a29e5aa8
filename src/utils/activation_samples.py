"""Import-path shim for the reference's sample collector (src/utils/activation_samples.py)."""
from matryoshka_transcoder_amd.activation_samples import ActivationSample, ActivationSampleCollector  # noqa: F401
