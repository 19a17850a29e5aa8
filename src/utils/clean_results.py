"""Import-path shim for the reference's results manager."""
from matryoshka_transcoder_amd.results import CleanResultsManager  # noqa: F401
