"""Configuration helpers with the reference's names and defaults (src/utils/config.py), restated
without the `transformer_lens` import the reference needs only for hook *names*."""
from __future__ import annotations

import copy
from typing import List

import torch

_MODELS = {
    "gpt2-medium": {"act_size": 1024, "n_layers": 24},
    "gpt2-large": {"act_size": 1280, "n_layers": 36},
    "gpt2-xl": {"act_size": 1600, "n_layers": 48},
    "gpt2": {"act_size": 768, "n_layers": 12},
    "gemma-2-2b": {"act_size": 2304, "n_layers": 26},
    "gemma-2-9b": {"act_size": 3584, "n_layers": 42},
    "gemma-2-27b": {"act_size": 4608, "n_layers": 46},
}
_ATTN_SITES = {"k", "q", "v", "z", "pattern", "attn_scores", "result", "rot_k", "rot_q"}
_MLP_SITES = {"pre", "post", "pre_linear"}


def get_model_config(model_name: str) -> dict:
    """config.py:7-30.  The reference scans its table in insertion order with `key in name`, so
    "gpt2" wins for every gpt2-* name; that order is kept."""
    name = model_name.lower()
    for key in ("gpt2", "gpt2-medium", "gpt2-large", "gpt2-xl", "gemma-2-2b", "gemma-2-9b", "gemma-2-27b"):
        if key in name:
            return dict(_MODELS[key])
    return {"act_size": 768, "n_layers": 12}


def get_act_name(site: str, layer=None) -> str:
    """The subset of transformer_lens.utils.get_act_name the reference relies on."""
    if layer is None:
        return f"hook_{site}"
    if site in _ATTN_SITES:
        return f"blocks.{layer}.attn.hook_{site}"
    if site in _MLP_SITES:
        return f"blocks.{layer}.mlp.hook_{site}"
    return f"blocks.{layer}.hook_{site}"


def get_hook_name(site: str, layer: int, model_name: str) -> str:
    """config.py:33-58."""
    if "gemma" in model_name.lower():
        if site == "ln2_normalized":
            return f"blocks.{layer}.ln2.hook_normalized"
        if site == "mlp_in":
            return f"blocks.{layer}.hook_mlp_in"
    return get_act_name(site, layer)


def get_default_cfg() -> dict:
    """config.py:61-111 (same keys and values)."""
    cfg = {
        "seed": 49, "batch_size": 4096, "lr": 3e-4, "num_tokens": int(1e9), "l1_coeff": 0, "beta1": 0.9,
        "beta2": 0.99, "max_grad_norm": 100000, "seq_len": 128, "dtype": torch.float32,
        "model_dtype": torch.float32, "model_name": "gpt2-small", "site": "resid_pre", "layer": 8,
        "act_size": 768, "dict_size": 12288, "device": "cuda:0", "model_batch_size": 512,
        "num_batches_in_buffer": 10, "dataset_path": "Skylion007/openwebtext",
        "wandb_project": "sparse_autoencoders", "input_unit_norm": True, "perf_log_freq": 1000,
        "sae_type": "topk", "checkpoint_freq": 10000, "n_batches_to_dead": 20,
        "top_k": 32, "top_k_aux": 512, "aux_penalty": (1 / 32), "bandwidth": 0.001,
        "save_activation_samples": False, "sample_collection_freq": 1000, "max_samples_per_feature": 100,
        "sample_context_size": 20, "sample_activation_threshold": 0.1, "top_features_to_save": 100,
        "samples_per_feature_to_save": 10, "store_tokens_for_samples": False,
    }
    return post_init_cfg(cfg)


def post_init_cfg(cfg: dict) -> dict:
    """config.py:113-131."""
    if cfg.get("act_size") is None:
        cfg["act_size"] = get_model_config(cfg["model_name"])["act_size"]
    cfg["hook_point"] = get_hook_name(cfg["site"], cfg["layer"], cfg["model_name"])
    cfg["name"] = (f"{cfg['model_name']}_{cfg['hook_point']}_{cfg['dict_size']}_{cfg['sae_type']}_"
                   f"{cfg['top_k']}_{cfg['lr']}")
    return cfg


def create_transcoder_config(base_cfg, source_layer, target_layer, source_site, target_site) -> dict:
    """transcoder_activation_store.py:217-260."""
    cfg = base_cfg.copy()
    cfg.update(source_layer=source_layer, target_layer=target_layer, source_site=source_site, target_site=target_site)
    cfg["source_hook_point"] = get_hook_name(source_site, source_layer, cfg["model_name"])
    cfg["target_hook_point"] = get_hook_name(target_site, target_layer, cfg["model_name"])
    cfg["layer"] = source_layer
    cfg["hook_point"] = cfg["source_hook_point"]
    cfg["source_act_size"] = cfg.get("source_act_size", cfg.get("act_size"))
    cfg["target_act_size"] = cfg.get("target_act_size", cfg.get("act_size"))
    cfg["name"] = (f"{cfg['model_name']}_{cfg['source_hook_point']}_to_{cfg['target_hook_point']}_"
                   f"{cfg['dict_size']}_{cfg['sae_type']}_{cfg['top_k']}_{cfg['lr']}")
    return cfg


def create_gemma_mlp_transcoder_config(base_cfg, layer) -> dict:
    """config.py:167-202: mlp_in -> mlp_out of one Gemma-2 layer."""
    cfg = base_cfg.copy()
    cfg.update(source_layer=layer, target_layer=layer, layer=layer, source_site="mlp_in", target_site="mlp_out",
               source_hook_point=f"blocks.{layer}.hook_mlp_in", target_hook_point=f"blocks.{layer}.hook_mlp_out")
    cfg["name"] = (f"{cfg['model_name']}_mlpin_to_mlpout_{layer}_{cfg['dict_size']}_{cfg['sae_type']}_"
                   f"{cfg['top_k']}_{cfg['lr']}")
    cfg["hook_point"] = cfg["source_hook_point"]
    return cfg


def build_autointerp_eval_cfg(base_cfg, *, layer, total_tokens=2_000_000, context_size=128, description="auto_interp",
                              source_site="mlp_in", target_site="mlp_out") -> dict:
    """config.py:134-164 (the reference's version raises NameError: it forgot to import
    create_transcoder_config; this one works)."""
    cfg = copy.deepcopy(base_cfg)
    cfg["num_tokens"] = int(total_tokens)
    cfg["seq_len"] = int(context_size)
    cfg.setdefault("model_batch_size", 32)
    cfg.setdefault("num_batches_in_buffer", 4)
    cfg["experiment_description"] = description
    cfg.setdefault("device", "cuda")
    cfg.setdefault("dtype", torch.float32)
    cfg.setdefault("results_root", "results")
    return create_transcoder_config(cfg, source_layer=layer, target_layer=layer, source_site=source_site,
                                    target_site=target_site)


def group_sizes_to_prefix_sizes(group_sizes: List[int]) -> List[int]:
    """README / BASELINE lists such as [2304, 4608, 9216, 20736] are GROUP sizes; the model consumes
    CUMULATIVE prefix sizes whose last entry is the dictionary size (sae.py:94-105)."""
    out, run = [], 0
    for g in group_sizes:
        run += int(g)
        out.append(run)
    return out


def compute_fvu(x_original: torch.Tensor, x_reconstruct: torch.Tensor) -> torch.Tensor:
    """config.py:205-234: mean((x - x_hat)^2) / (var_unbiased(all elements of x) + 1e-8), fp32.
    Convenience for callers; the training forward computes all FVUs inside the decode kernel."""
    resid = x_original.float() - x_reconstruct.float()
    return resid.pow(2).mean() / (x_original.float().var() + 1e-8)
