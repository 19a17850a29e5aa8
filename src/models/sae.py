"""Import-path shim: `from src.models.sae import MatryoshkaTranscoder` (or `models.sae` with src/ on
sys.path) resolves to the MI355X implementation."""
from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder  # noqa: F401
from matryoshka_transcoder_amd.config import compute_fvu  # noqa: F401
