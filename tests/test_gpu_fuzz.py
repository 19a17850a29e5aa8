"""Seeded random configurations through the whole fused training step (FusedAdam.train_step: a1-a15 of SURVEY 8)
against the oracle's train step -- shapes nobody picked by hand: odd batch sizes, widths that are a multiple of 8
only, tiny first prefix groups, k from 1 to a quarter of the dictionary, the auxiliary branch switching on by itself
after a few steps (n_batches_to_dead = 2 or 3 with some encoder columns scaled down), with and without unit-norm.

Reference: sae.py:171-347, training.py:172-179 through oracle/ref_cpu.py (pinned to the reference's fixtures in
tests/test_oracle_vs_golden.py).  The bar per step: selection (row pointers, columns, values) and dead counters BIT-EXACT
given the GPU's activations; scalars and parameters within the float tolerance of conftest.tol (fp32 1e-4 relative;
bf16 2e-2 relative on scalars, a few lr of drift per step on parameters: tie classes)."""
import random

import pytest
import torch

from conftest import tol
from oracle import ref_cpu as R
from test_gpu_kernels import _close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _draw(seed):
    r = random.Random(1000 + seed)
    dt = torch.bfloat16 if seed % 2 == 0 else torch.float32
    A = r.choice([8, 16, 24, 40, 64, 72, 128, 136, 264, 520, 1032, 1544])
    G = r.randint(1, 5)
    D = r.choice([64, 96, 200, 512, 1000, 1208, 2048, 3000, 6000])
    cuts = sorted(r.sample(range(1, D), G - 1)) + [D] if G > 1 else [D]
    B = r.choice([1, 2, 3, 17, 37, 64, 100, 130, 200, 300])
    k = r.randint(1, max(1, min(64, D // 4)))
    cfg = dict(seed=seed, prefix_sizes=cuts, act_size=A, top_k=k, top_k_aux=r.choice([4, 16, 64]), aux_penalty=1 / 32,
               n_batches_to_dead=r.choice([2, 3]), l1_coeff=r.choice([0.0, 1e-3]), dtype=dt, device=DEV,
               input_unit_norm=r.random() < 0.4, lr=3e-4, beta1=0.9, beta2=0.99)
    return cfg, B, r


@pytest.mark.parametrize("seed", range(64))
def test_random_configuration_against_oracle(seed, gpu_lib):
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    cfg, B, r = _draw(seed)
    dt, A, D = cfg["dtype"], cfg["act_size"], cfg["prefix_sizes"][-1]
    f32 = dt == torch.float32
    m = MatryoshkaTranscoder(cfg)
    ocfg = dict(cfg, device="cpu")
    params, state = R.init_params(ocfg), R.init_state(ocfg)
    # a slice of the dictionary hardly ever wins the global TopK: those features die after n_batches_to_dead steps and
    # switch the auxiliary branch on (where they do fire, they come back: the dead set changes from step to step)
    tiny = torch.tensor(sorted(r.sample(range(D), max(1, D // r.choice([4, 9, 30])))))
    with torch.no_grad():
        params["W_enc"][:, tiny] *= 0.02
        params["b_enc"] = (0.01 * torch.randn(D, generator=torch.Generator().manual_seed(seed))).to(dt)
        m.load_state_dict({k_: v.to(DEV) for k_, v in params.items()})
    opt = FusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    oopt = R.new_opt(params)
    hp = m.hot_path()
    aux_steps = 0
    for i in range(5):
        g = torch.Generator("cpu").manual_seed(seed * 100 + i)
        xs = (torch.randn(B, A, generator=g) * (1.0 + 0.5 * i)).to(dt)
        xt = torch.randn(B, A, generator=g).to(dt)
        xs_d, xt_d = xs.to(DEV), xt.to(DEV)
        with torch.no_grad():
            xn = xs_d
            if cfg["input_unit_norm"]:
                xn = hp.normalize(hp.desc(B), xs_d)[0]
            acts_gpu = hp.encode(hp.desc(B), xn, m.W_enc.t(), m.b_enc).cpu()
        out = opt.train_step(xs_d, xt_d)
        o, inter, _ = R.train_step(params, state, oopt, xs, xt, ocfg, acts_override=acts_gpu)
        aux_steps += int(inter["n_dead"] > 0)
        # a3: the selection, bit for bit
        sel = m._last_res.sel
        nnz = int(inter["flat_idx"].numel())
        rp = sel.row_ptr.cpu()
        assert torch.equal(rp, inter["row_ptr"]), (seed, i)
        rows = torch.repeat_interleave(torch.arange(B), (rp[1:] - rp[:-1]).long())
        assert torch.equal(rows * D + sel.col_idx.cpu()[:nnz].long(), inter["flat_idx"]), (seed, i)
        assert torch.equal(sel.vals.cpu()[:nnz], inter["vals"]), (seed, i)
        # a7
        assert torch.equal(m._cnt.cpu(), state["num_batches_not_active"]), (seed, i)
        rt, at = tol(dt)
        for name in ("loss", "l2_loss", "l1_loss", "aux_loss", "l0_norm", "fvu"):
            _close(out[name].float(), o[name].float(), rt, at if name != "aux_loss" else at * 0.01 + 1e-7, f"seed {seed} step {i} {name}")
            assert out[name].dtype == o[name].dtype, (seed, i, name)
        ref_out = o["transcoder_out"].float()
        # (elements that cancel to ~0 carry the absolute error of the row's largest terms: atol scales with the output)
        _close(out["transcoder_out"].float().cpu(), ref_out, rt, (2e-5 if f32 else 4e-3) * max(1.0, float(ref_out.abs().max())),
               f"seed {seed} step {i} transcoder_out")
        for name in ("W_dec", "W_enc", "b_enc", "b_dec"):
            got, ref = getattr(m, name).detach().float().cpu(), params[name].float()
            if f32:
                # (Adam's update is lr * m / (sqrt(v) + eps): where a gradient element nearly cancels, fp32 summation order
                # moves the update by a few per cent of lr -- seen: 1 element of 9.3 M at 2.2 % after 5 steps)
                _close(got, ref, 0, 5e-2 * cfg["lr"] + 1e-7, f"seed {seed}: {name} after step {i}")
            else:
                assert float((got - ref).abs().max()) <= 2.0 * (i + 1) * cfg["lr"] + 2.0 ** -8 * float(ref.abs().max()), (seed, i, name)
    if cfg["n_batches_to_dead"] <= 3 and D >= 200:
        assert aux_steps > 0, "the scenario is meant to reach the auxiliary branch"
    # a4 / a15 on the trained weights: eval-mode forward (threshold branch) and encode() / decode() on a 3-D batch,
    # against the oracle's eval forward on the model's own parameters (sae.py:189-191, 363-406)
    m.eval()
    p_now = {k_: v.detach().cpu() for k_, v in m.state_dict().items()}
    g = torch.Generator("cpu").manual_seed(seed * 100 + 99)
    S = max(1, min(4, B))
    xs3 = torch.randn(S, 3, A, generator=g).to(dt)
    xs2, xt2 = xs3.reshape(-1, A), torch.randn(S * 3, A, generator=g).to(dt)
    with torch.no_grad():
        xn = hp.normalize(hp.desc(S * 3), xs2.to(DEV))[0] if cfg["input_unit_norm"] else xs2.to(DEV)
        acts_gpu = hp.encode(hp.desc(S * 3), xn, m.W_enc.t(), m.b_enc).cpu()
        out = m(xs2.to(DEV), xt2.to(DEV))
        feats = m.encode(xs3.to(DEV))
        recon = m.decode(feats.reshape(-1, D))
    o, _, inter = R.forward(p_now, {"num_batches_not_active": m._cnt.cpu().clone()}, xs2, xt2, ocfg, training=False,
                            acts_override=acts_gpu)
    assert feats.shape == (S, 3, D)
    dense = torch.zeros(S * 3 * D, dtype=dt)
    dense[inter["flat_idx"]] = inter["vals"]
    assert torch.equal(feats.reshape(-1).cpu(), dense), f"seed {seed}: eval encode()"
    rt, at = tol(dt)
    ref_out = o["transcoder_out"].float()
    bar = (2e-5 if f32 else 4e-3) * max(1.0, float(ref_out.abs().max()))
    _close(out["transcoder_out"].float().cpu(), ref_out, rt, bar, f"seed {seed}: eval forward")
    _close(recon.float().cpu(), ref_out, rt, bar, f"seed {seed}: decode(encode(x))")
    for name in ("loss", "l2_loss", "l0_norm", "fvu"):
        _close(out[name].float(), o[name].float(), rt, at, f"seed {seed} eval {name}")
