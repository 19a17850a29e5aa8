"""Import-path shim for the reference's training module."""
from matryoshka_transcoder_amd.training import (  # noqa: F401
    FusedAdam, create_scheduler, linear_warmup, log_transcoder_performance, train_transcoder, warmup_cosine_decay)
