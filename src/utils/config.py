"""Import-path shim for the reference's config module."""
from matryoshka_transcoder_amd.config import *  # noqa: F401,F403
from matryoshka_transcoder_amd.config import (  # noqa: F401
    build_autointerp_eval_cfg, compute_fvu, create_gemma_mlp_transcoder_config, get_default_cfg, get_hook_name,
    get_model_config, post_init_cfg)
