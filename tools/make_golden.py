#!/usr/bin/env python3
"""Generate tests/golden/*.pt by running the REFERENCE itself (read-only, from /root/reference).

Runs only in the build container (the reference does not travel to the GPU box).  The
reference class is imported unmodified; the one absent third-party import on its path
(`transformer_lens.utils.get_act_name`, used by config.py only for hook *names*) is satisfied
by an in-memory stub module.  Everything written is data: inputs, weights, and the outputs the
reference produced for them.

    python tools/make_golden.py            # rewrites every fixture
"""
import copy
import os
import sys
import types

import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    tl = types.ModuleType("transformer_lens")
    tlu = types.ModuleType("transformer_lens.utils")
    tlu.get_act_name = lambda site, layer=None: f"blocks.{layer}.hook_{site}"
    tl.utils = tlu
    sys.modules.setdefault("transformer_lens", tl)
    sys.modules.setdefault("transformer_lens.utils", tlu)
    sys.path.insert(0, REF)
    import models.sae as sae  # noqa
    return sae


def batch(i, B, a_s, a_t, dt):
    g = torch.Generator("cpu").manual_seed(1000 + i)
    return torch.randn(B, a_s, generator=g).to(dt), torch.randn(B, a_t, generator=g).to(dt)


def sd(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def lr_lambda(kind, warmup, total):
    import math
    if kind == "warmup_decay":
        def f(step):
            if step < warmup:
                return step / warmup
            p = (step - warmup) / (total - warmup)
            return 0.5 * (1 + math.cos(math.pi * p))
        return f
    return None


def make_case(sae, name, dt, unit, a_s, a_t, aux, l1_coeff=0.0, sched=None):
    cfg = dict(seed=49, prefix_sizes=[16, 48, 112, 256], act_size=a_s, source_act_size=a_s,
               target_act_size=a_t, top_k=4, top_k_aux=8, aux_penalty=1 / 32, n_batches_to_dead=20,
               l1_coeff=l1_coeff, dtype=dt, device="cpu", input_unit_norm=unit,
               lr=3e-4, beta1=0.9, beta2=0.99)
    B, D = 16, 256
    m = sae.MatryoshkaTranscoder(cfg)
    if aux:
        # force a subset dead: tiny encoder columns never win the global TopK
        dead_cols = torch.arange(5, D, 7)
        with torch.no_grad():
            m.W_enc[:, dead_cols] *= 0.01
        m.num_batches_not_active[dead_cols] = 20.0
    fx = {"cfg": {k: (str(v) if isinstance(v, torch.dtype) else v) for k, v in cfg.items()},
          "B": B, "init": sd(m), "cnt0": m.num_batches_not_active.clone()}

    # --- intermediates of step 0, taken from a scratch copy so the real run is untouched
    xs, xt = batch(0, B, a_s, a_t, dt)
    fx["x_source0"], fx["x_target0"] = xs, xt
    probe = copy.deepcopy(m)
    with torch.no_grad():
        xn, mean, std = probe.preprocess_input(xs)
        acts, acts_topk = probe.compute_activations(xn)
        top = torch.topk(acts.flatten(), cfg["top_k"] * B, dim=-1)
    fx["x_norm"], fx["mean"], fx["std"] = xn, mean, std
    fx["acts"] = acts
    fx["topk_indices"] = top.indices.clone()
    fx["topk_values"] = top.values.clone()
    fx["acts_topk_nonzero"] = torch.nonzero(acts_topk.flatten()).flatten()

    # --- capture every prefix reconstruction through the module-level compute_fvu hook
    seen = []
    orig_fvu = sae.compute_fvu

    def spy(x_original, x_reconstruct):
        seen.append(x_reconstruct.detach().clone())
        return orig_fvu(x_original, x_reconstruct)

    sae.compute_fvu = spy
    opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    scheduler = None
    if sched:
        lam = lr_lambda(sched, 2, 8)
        scheduler = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lam)
    steps = []
    for i in range(3):
        xs, xt = batch(i, B, a_s, a_t, dt)
        seen.clear()
        out = m(xs, xt)
        rec = {"x_source": xs, "x_target": xt,
               "out": {k: v.detach().clone() for k, v in out.items()},
               "prefix_recons": [t.clone() for t in seen[:len(cfg["prefix_sizes"])]],
               "threshold_after_fwd": m.threshold.detach().clone(),
               "cnt_after_fwd": m.num_batches_not_active.clone(),
               "lr": opt.param_groups[0]["lr"]}
        out["loss"].backward(retain_graph=True)
        rec["grads"] = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
        m.make_decoder_weights_and_grad_unit_norm()
        rec["W_dec_grad_projected"] = m.W_dec.grad.detach().clone()
        rec["W_dec_normed"] = m.W_dec.detach().clone()
        opt.step()
        rec["params_after"] = sd(m)
        rec["exp_avg"] = {n: opt.state[p]["exp_avg"].clone() for n, p in m.named_parameters()}
        rec["exp_avg_sq"] = {n: opt.state[p]["exp_avg_sq"].clone() for n, p in m.named_parameters()}
        opt.zero_grad()
        if scheduler is not None:
            scheduler.step()
        steps.append(rec)
    sae.compute_fvu = orig_fvu
    fx["steps"] = steps

    # --- encode / decode, 2-D and 3-D, train mode then eval (threshold) mode
    x2, _ = batch(7, B, a_s, a_t, dt)
    x3 = x2.reshape(2, B // 2, a_s)
    encdec = {"x2": x2}
    for mode in ("train", "eval"):
        mm = copy.deepcopy(m)
        mm.train(mode == "train")
        with torch.no_grad():
            e2 = mm.encode(x2)
            d2 = mm.decode(e2)
            thr2 = mm.threshold.detach().clone()
            e3 = mm.encode(x3)
            # with input_unit_norm the reference's decode cannot consume a 3-D encode's stats
            # (sae.py:152 broadcasts [2,8,1] stats against [16,A]); record dec3 only without it
            d3 = None if unit else mm.decode(e3.reshape(-1, e3.shape[-1]))
        encdec[mode] = {"enc2": e2, "dec2": d2, "thr_after_enc2": thr2, "enc3": e3}
        if d3 is not None:
            encdec[mode]["dec3"] = d3
    # eval-mode forward (threshold branch, still mutates counters)
    mm = copy.deepcopy(m)
    mm.eval()
    with torch.no_grad():
        xs, xt = batch(5, B, a_s, a_t, dt)
        out = mm(xs, xt)
    encdec["eval_forward"] = {"x_source": xs, "x_target": xt,
                              "out": {k: v.detach().clone() for k, v in out.items()},
                              "cnt_after": mm.num_batches_not_active.clone(),
                              "threshold": mm.threshold.detach().clone(),
                              "cnt_before": m.num_batches_not_active.clone(),
                              "state": sd(m)}
    fx["encdec"] = encdec
    torch.save(fx, os.path.join(OUT, f"{name}.pt"))
    print("wrote", name)


def make_medium(sae):
    """A=768, D=12288, B=64, k=64: indices / scalars / checksums only."""
    for dt, tag in ((torch.float32, "fp32"), (torch.bfloat16, "bf16")):
        cfg = dict(seed=49, prefix_sizes=[768, 2304, 5376, 12288], act_size=768, top_k=64,
                   top_k_aux=512, aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=0.0, dtype=dt,
                   device="cpu", input_unit_norm=False, lr=3e-4, beta1=0.9, beta2=0.99)
        B = 64
        m = sae.MatryoshkaTranscoder(cfg)
        xs, xt = batch(0, B, 768, 768, dt)
        with torch.no_grad():
            acts, _ = copy.deepcopy(m).compute_activations(xs)
            top = torch.topk(acts.flatten(), cfg["top_k"] * B, dim=-1)
        out = m(xs, xt)
        out["loss"].backward()
        fx = {"cfg": {k: (str(v) if isinstance(v, torch.dtype) else v) for k, v in cfg.items()},
              "B": B,
              "topk_indices_sorted": torch.sort(top.indices).values,
              "tau": top.values.min(),
              "n_gt_tau": (acts.flatten() > top.values.min()).sum(),
              "n_eq_tau": (acts.flatten() == top.values.min()).sum(),
              "out": {k: v.detach().clone() for k, v in out.items() if v.dim() == 0},
              "threshold_after": m.threshold.detach().clone(),
              "init_checksum": {k: v.double().sum() for k, v in m.state_dict().items()},
              "init_abs_checksum": {k: v.double().abs().sum() for k, v in m.state_dict().items()},
              "out_checksum": out["transcoder_out"].double().sum(),
              "grad_abs_checksum": {n: p.grad.double().abs().sum() for n, p in m.named_parameters()},
              "grad_row_nonzero": {"W_dec": (m.W_dec.grad.abs().sum(-1) > 0),
                                   "W_enc": (m.W_enc.grad.abs().sum(0) > 0)},
              "db_dec": m.b_dec.grad.detach().clone(),
              "db_enc": m.b_enc.grad.detach().clone()}
        torch.save(fx, os.path.join(OUT, f"medium_{tag}.pt"))
        print("wrote medium", tag)


def main():
    os.makedirs(OUT, exist_ok=True)
    sae = import_reference()
    f32, bf = torch.float32, torch.bfloat16
    make_case(sae, "fp32_plain", f32, False, 32, 32, False)
    make_case(sae, "fp32_unit", f32, True, 32, 32, False)
    make_case(sae, "fp32_cross", f32, False, 48, 32, False, l1_coeff=0.01)
    make_case(sae, "fp32_aux", f32, False, 32, 32, True, l1_coeff=0.01)
    make_case(sae, "fp32_aux_unit", f32, True, 48, 32, True)
    make_case(sae, "fp32_sched", f32, False, 32, 32, False, sched="warmup_decay")
    make_case(sae, "bf16_plain", bf, False, 32, 32, False)
    make_case(sae, "bf16_unit", bf, True, 32, 32, False)
    make_case(sae, "bf16_aux", bf, False, 32, 32, True)
    make_medium(sae)


if __name__ == "__main__":
    main()
