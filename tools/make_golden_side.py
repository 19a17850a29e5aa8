#!/usr/bin/env python3
"""Generate tests/golden/side_*.pt for the SURVEY 8f rows 3 and 4 by running the REFERENCE itself
(read-only, from /root/reference): the top-activating-sample collector
(src/utils/activation_samples.py) and the SAEBench adapters (eval/sae_bench/wrappers.py).

Runs only in the build container.  The two absent third-party names on the import path
(`transformer_lens.utils.get_act_name`, `transformer_lens.hook_points.HookedRootModule`, used for hook
names / a type annotation) are satisfied by in-memory stub modules.  Everything written is data.

    python tools/make_golden_side.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True


def import_reference():
    tl = types.ModuleType("transformer_lens")
    tlu = types.ModuleType("transformer_lens.utils")
    thp = types.ModuleType("transformer_lens.hook_points")
    tlu.get_act_name = lambda site, layer=None: f"blocks.{layer}.hook_{site}"
    thp.HookedRootModule = object
    tl.utils, tl.hook_points = tlu, thp
    sys.modules.setdefault("transformer_lens", tl)
    sys.modules.setdefault("transformer_lens.utils", tlu)
    sys.modules.setdefault("transformer_lens.hook_points", thp)
    sys.path.insert(0, "/root/reference")
    sys.path.insert(0, "/root/reference/src")
    import utils.activation_samples as samples
    import eval.sae_bench.wrappers as wrappers
    from eval.sae_bench.config import NPZTranscoderSpec

    # As shipped, NPZTranscoderAdapter.__init__ (wrappers.py:176-204) builds a cfg without "seed" and
    # "act_size", and MatryoshkaTranscoder.__init__ (sae.py:89,111) raises KeyError on both.  To get
    # vectors out of the rest of that code, the two keys are supplied on the way in; nothing else is touched.
    inner = wrappers.MatryoshkaAutoInterpAdapter.__init__

    def with_missing_keys(self, cfg, *, sae_cfg):
        cfg = dict(cfg)
        cfg.setdefault("seed", 0)
        cfg.setdefault("act_size", cfg.get("source_act_size"))
        inner(self, cfg, sae_cfg=sae_cfg)

    wrappers.MatryoshkaAutoInterpAdapter.__init__ = with_missing_keys
    return samples, wrappers, NPZTranscoderSpec


class WordTokenizer:
    """decode(ids) -> 't<id>' words; the same stand-in is used by the tests."""

    def decode(self, ids):
        return " ".join(f"t{int(i)}" for i in ids)


def sparse_acts(seed, batch, seq, n_feat, dt, density, quant=None):
    g = torch.Generator("cpu").manual_seed(seed)
    a = torch.rand(batch, seq, n_feat, generator=g) * 3.0
    if quant:                                   # few distinct values -> many exact ties
        a = torch.round(a * quant) / quant
    keep = torch.rand(batch, seq, n_feat, generator=g) < density
    a = torch.where(keep, a, torch.zeros(()))
    a[:, :, 3] = 0.0                            # a feature that never fires
    a[:, :, 5] = a[:, :, 5].clamp(max=0.09)     # a feature that never passes the storage threshold
    toks = torch.randint(0, 50000, (batch, seq), generator=g)
    return a.to(dt), toks


def collector_case(samples, name, dt, max_samples, context_size, thr, quant, flat, n_feat=24, tracked=None):
    col = samples.ActivationSampleCollector(max_samples_per_feature=max_samples, context_size=context_size,
                                            storage_threshold=thr)
    tok = WordTokenizer()
    rounds = []
    for r in range(3):
        acts, toks = sparse_acts(300 + r, 4, 16, n_feat, dt, 0.35, quant)
        given = acts.reshape(-1, n_feat) if flat else acts
        col.collect_batch_samples(given, toks, tok, feature_indices=tracked)
        rounds.append({"acts": given.clone(), "tokens": toks.clone()})
    feats = {}
    for f in range(n_feat):
        top = col.get_top_samples(f)
        feats[f] = {"samples": [s.to_dict() for s in top],
                    "statistics": {k: (float(v) if isinstance(v, (float, np.floating)) else v)
                                   for k, v in col.get_feature_statistics(f).items()}}
    with tempfile.TemporaryDirectory() as tmp:
        col.save_samples(tmp, top_k_features=6, samples_per_feature=3)
        saved = sorted(os.listdir(tmp))
    fx = {"max_samples": max_samples, "context_size": context_size, "storage_threshold": thr,
          "tracked": tracked, "rounds": rounds, "features": feats, "overall": col.get_all_statistics(),
          "saved_files_top6": saved,
          "counts": {int(k): int(v) for k, v in col.feature_activation_counts.items()},
          "max_activations": {int(k): float(v) for k, v in col.feature_max_activations.items()}}
    torch.save(fx, os.path.join(OUT, f"side_collector_{name}.pt"))
    print("wrote collector", name)


def adapter_case(wrappers, NPZTranscoderSpec, name, dt, transposed, prefixes):
    d_in, d_out, d_sae, n = 48, 32, 256, 24
    g = torch.Generator("cpu").manual_seed(77)
    W_enc = torch.randn(d_in, d_sae, generator=g) / d_in ** 0.5
    W_dec = torch.randn(d_sae, d_out, generator=g) * 0.3
    b_enc = torch.randn(d_sae, generator=g) * 0.05
    b_dec = torch.randn(d_out, generator=g) * 0.05
    arrays = {"W_enc": W_enc.numpy(), "W_dec": W_dec.numpy(), "b_enc": b_enc.numpy(), "b_dec": b_dec.numpy()}
    if transposed:                              # decoder-major / column-vector layouts the loader must coerce
        arrays = {"encoder.W": W_enc.t().contiguous().numpy(), "decoder.W": W_dec.t().contiguous().numpy(),
                  "encoder.b": b_enc.reshape(-1, 1).numpy(), "decoder.b": b_dec.reshape(1, -1).numpy()}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "ckpt.npz")
        np.savez(path, **arrays)
        spec = NPZTranscoderSpec(path=path, model_name="gemma-2-2b", source_layer=8,
                                 source_hook_point="blocks.8.hook_resid_post",
                                 target_hook_point="blocks.9.hook_resid_pre", d_in=d_in, d_out=d_out, d_sae=d_sae,
                                 target_layer=9, dtype=str(dt).split(".")[-1], device="cpu", top_k=8,
                                 prefix_sizes=prefixes)
        ad = wrappers.build_npz_transcoder_adapter(spec)
    xs = torch.randn(n, d_in, generator=g).to(dt)
    xt = torch.randn(n, d_out, generator=g).to(dt)
    fx = {"arrays": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in arrays.items()},
          "spec": {"d_in": d_in, "d_out": d_out, "d_sae": d_sae, "top_k": 8, "prefix_sizes": prefixes,
                   "dtype": str(dt).split(".")[-1], "source_layer": 8, "target_layer": 9},
          "state_after_load": {k: v.detach().clone() for k, v in ad.state_dict().items()},
          "prefix_sizes": list(ad.prefix_sizes), "top_k": ad.config["top_k"], "top_k_aux": ad.config["top_k_aux"],
          "cfg_fields": {k: getattr(ad.cfg, k) for k in ("model_name", "hook_layer", "hook_name", "d_in", "d_sae",
                                                        "dtype", "device", "architecture", "activation_fn_str",
                                                        "prepend_bos", "normalize_activations", "training_tokens")},
          "row_norms": ad.get_decoder_row_norms().clone(), "x_source": xs, "x_target": xt}
    for mode in ("train", "eval"):
        ad.train(mode == "train")
        with torch.no_grad():
            if mode == "eval":
                ad.threshold.fill_(0.25)
            thr0 = ad.threshold.detach().clone()
            rep = ad.validate_reconstruction(xs, xt)
            enc = ad.encode(xs)
        fx[mode] = {"threshold_before": thr0, "fvu": float(rep.fvu), "mse": float(rep.mse),
                    "mean_activation": float(rep.mean_activation), "threshold_after": ad.threshold.detach().clone(),
                    "encoded_after": enc}
    torch.save(fx, os.path.join(OUT, f"side_adapter_{name}.pt"))
    print("wrote adapter", name)


def main():
    os.makedirs(OUT, exist_ok=True)
    samples, wrappers, NPZTranscoderSpec = import_reference()
    f32, bf = torch.float32, torch.bfloat16
    collector_case(samples, "fp32", f32, 5, 6, 0.1, None, flat=False)
    collector_case(samples, "fp32_flat_tracked", f32, 4, 1, 0.5, None, flat=True, tracked=[0, 2, 3, 7, 11, 23])
    collector_case(samples, "bf16_ties", bf, 6, 20, 0.1, 4, flat=True)
    adapter_case(wrappers, NPZTranscoderSpec, "fp32", f32, False, [32, 96, 256])
    adapter_case(wrappers, NPZTranscoderSpec, "fp32_transposed", f32, True, [64, 128])   # 256 gets appended
    adapter_case(wrappers, NPZTranscoderSpec, "bf16", bf, False, None)


if __name__ == "__main__":
    main()
