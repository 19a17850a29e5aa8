"""GPU parity for the kernel instantiations BASELINE's larger configurations use and the small fixtures never reach:

  * row widths A = 1536 / 3584 / 4608 / 8192 -> NC = 3 / 7 / 9 / 16 register chunks per lane in the row-gather kernels
    (decode, fused gradient + Adam, unit-norm rows): Gemma-2-9B is NC = 7, Gemma-2-27B NC = 9 (rowops.hpp nc_for);
  * the 256 x 256 MFMA encoder (the headline kernel; picked automatically only for >= 512 big tiles) forced at small
    ragged shapes through mtc_encode_set_tile(2), against the oracle's fp32 contraction and the 128 x 128 kernel;
  * BASELINE config #2 at FULL size: sampled activation rows against an fp32 CPU contraction, and the GPU selection
    bit-exact against the oracle's select run on the host over the GPU's own activations;
  * the reference-made medium fixtures (A 768, D 12288) through the HIP path;
  * three bf16 steps against the oracle fed the GPU's activations (same tie policy, same selection): weights within
    one bf16 ulp for >= 99.9 % of the entries -- kernel error separated from tie ambiguity.

Bars as everywhere: indices bit-exact given identical activations; fp32 rel 1e-4; bf16 rel 2e-2 / 1 ulp."""
import math

import pytest
import torch

from conftest import load_golden, tol
from oracle import ref_cpu as R
from test_gpu_kernels import _close, _hot, check_against_oracle, run_forward

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def force_tile(gpu_lib):
    """Force an encoder tile for one test; always restored."""
    prev = []

    def set_tile(t):
        prev.append(gpu_lib.mtc_encode_set_tile(int(t)))

    yield set_tile
    if prev:
        gpu_lib.mtc_encode_set_tile(prev[0])


# ------------------------------------------------------------------ (a) NC = 3 / 7 / 9 / 16
WIDE = [(1536, 3), (3584, 7), (4608, 9), (8192, 16)]


@pytest.mark.parametrize("A,nc", WIDE)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("unit", [False, True])
def test_wide_rows_against_oracle(A, nc, dt, unit, gpu_lib):
    """The whole forward + backward kernel chain at the row widths of the larger models (5 prefix groups, one of them
    tiny), against the oracle: select / CSR / CSC / counters bit-exact, everything else to the stated tolerances."""
    assert (A // 8 + 63) // 64 == nc
    D = 2048
    cfg = dict(seed=49, prefix_sizes=[8, 256, 512, 1024, D], act_size=A, top_k=16, top_k_aux=64,
               aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=0.0 if unit else 1e-3, dtype=dt, device="cpu",
               input_unit_norm=unit, lr=3e-4, beta1=0.9, beta2=0.99)
    B = 96
    params = R.init_params(cfg)
    params["b_enc"] = (0.01 * torch.randn(D, generator=torch.Generator().manual_seed(3))).to(dt)
    params["b_dec"] = (0.01 * torch.randn(A, generator=torch.Generator().manual_seed(4))).to(dt)
    g = torch.Generator("cpu").manual_seed(1000 + A)
    xs = torch.randn(B, A, generator=g).to(dt)
    xt = torch.randn(B, A, generator=g).to(dt)
    hp = _hot(cfg)
    cnt = torch.zeros(D)
    res = run_forward(hp, cfg, params, cnt, xs, xt)
    check_against_oracle(cfg, params, cnt, xs, xt, res, hp)


@pytest.mark.parametrize("A,nc", WIDE)
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_wide_rows_fused_step_vs_oracle(A, nc, dt, gpu_lib):
    """FusedAdam.train_step (decode with deferred d a_bj -> fused gradient + projection + Adam per row) at the wide
    row widths: two steps against the oracle fed the GPU's activations."""
    _steps_vs_oracle(A, 1024, [64, 192, 448, 1024], B=64, k=12, dt=dt, unit=True, steps=2, min_frac=0.999)


# ------------------------------------------------------------------ (f) multi-step, oracle shares the activations
def _exact_ulp(ref: torch.Tensor, floor_exp: int) -> torch.Tensor:
    """One unit in the last place of each bf16 reference value (2^(e-8) for m * 2^e, m in [0.5, 1)), floored."""
    _, e = torch.frexp(ref.float())
    return torch.ldexp(torch.ones_like(ref, dtype=torch.float32), (e - 8).clamp_min(floor_exp))


def _steps_vs_oracle(A, D, prefix, B, k, dt, unit, steps, min_frac, aux=False):
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    f32 = dt == torch.float32
    cfg = dict(seed=49, prefix_sizes=prefix, act_size=A, top_k=k, top_k_aux=32, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=dt, device=DEV, input_unit_norm=unit, lr=3e-4, beta1=0.9,
               beta2=0.99)
    m = MatryoshkaTranscoder(cfg)
    opt = FusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    ocfg = dict(cfg, device="cpu")
    params, state = R.init_params(ocfg), R.init_state(ocfg)
    oopt = R.new_opt(params)
    hp = m.hot_path()
    report = {}
    for i in range(steps):
        g = torch.Generator("cpu").manual_seed(1000 + i)
        xs = torch.randn(B, A, generator=g).to(dt)
        xt = torch.randn(B, A, generator=g).to(dt)
        xs_d, xt_d = xs.to(DEV), xt.to(DEV)
        with torch.no_grad():
            d = hp.desc(B)
            xn = hp.normalize(d, xs_d)[0] if unit else xs_d
            acts_gpu = hp.encode(d, xn, m.W_enc.t(), m.b_enc).cpu()          # activations of THIS step's GPU weights
        out = opt.train_step(xs_d, xt_d)
        o, inter, _ = R.train_step(params, state, oopt, xs, xt, ocfg, acts_override=acts_gpu)
        # identical selection (same activations, same tie policy)
        sel = m._last_res.sel
        rp = sel.row_ptr.cpu().long()
        nnz = int(rp[-1])
        rows = torch.repeat_interleave(torch.arange(B), rp[1:] - rp[:-1])
        assert torch.equal(rows * D + sel.col_idx.cpu()[:nnz].long(), inter["flat_idx"]), f"step {i}: selection"
        rt, at = tol(dt)
        for key in ("loss", "l2_loss", "l0_norm", "l1_norm", "fvu", "fvu_min", "fvu_max", "fvu_mean"):
            _close(out[key].float(), o[key].float(), rt, at, f"step {i} {key}")
        assert torch.equal(m.num_batches_not_active.cpu(), state["num_batches_not_active"])
        for name in ("W_dec", "W_enc", "b_enc", "b_dec"):
            got, ref = getattr(m, name).detach().float().cpu(), params[name].float()
            if f32:
                _close(got, ref, 0, 1e-2 * cfg["lr"] + 1e-7, f"{name} after step {i}")
                continue
            if name.startswith("W_"):
                frac = float(((got - ref).abs() <= _exact_ulp(ref, -24)).float().mean())
                assert frac >= min_frac, f"{name} after step {i}: only {frac:.5f} within 1 bf16 ulp of the oracle"
            else:
                # biases start at zero, so they ARE the sum of the Adam updates: the three bf16 roundings inside one
                # update (m, sqrt(v) / sqrt(bc2), m / denom) show directly -- the bulk must agree within 4 ulps (about
                # 1 % of an lr-sized update); a near-zero gradient whose sign flips under rounding turns into a full
                # +-lr step, so single entries may differ by up to 2 lr per step
                frac = float(((got - ref).abs() <= 4 * _exact_ulp(ref, -24)).float().mean())
                assert frac >= 0.98, f"{name} after step {i}: {frac:.5f} within 4 ulps"
                assert float((got - ref).abs().max()) <= 2.0 * (i + 1) * cfg["lr"] + 1e-6, name
            report[(i, name)] = frac
    return report


@pytest.mark.parametrize("unit", [False, True])
def test_three_steps_bf16_vs_oracle_shared_activations(unit, gpu_lib):
    """SURVEY section 9's bf16 bar (weights within 1 ulp for >= 99.9 % of entries) over three fused steps.  Against the
    REFERENCE fixtures the bar is 97 % because torch.topk's tie choice moves whole rows; the oracle shares our tie
    policy and, fed the GPU's activations, our selection -- so what is left here is kernel arithmetic only."""
    rep = _steps_vs_oracle(256, 2048, [128, 384, 896, 2048], B=256, k=16, dt=torch.bfloat16, unit=unit, steps=3,
                           min_frac=0.999)
    print("fraction of entries within 1 bf16 ulp:", {f"{k[1]}@{k[0]}": round(v, 5) for k, v in rep.items()})


# ------------------------------------------------------------------ (b) the 256 x 256 MFMA encoder, forced
@pytest.mark.parametrize("B,A,D", [(300, 768, 1000), (520, 2304, 776), (64, 2304, 264),
                                   # 3 x 101 = 303 tiles on 256 persistent workgroups: some walk two tiles, some one;
                                   # 1800 x 64 tiles: every workgroup walks 7 (K = 64: one k-step per tile)
                                   (520, 128, 25608), (2048, 64, 57344)])
def test_encoder_256_tile_against_oracle(B, A, D, gpu_lib, force_tile):
    """encode_bf16_mfma256_kernel at ragged shapes (partial batch tiles, partial feature tiles, D % 256 != 0), K = 768
    (12 k-tiles) and K = 2304 (36 k-tiles: the production staging schedule), against the oracle's fp32 contraction,
    bit for bit against the 128 x 128 kernel (same MFMA, same k order), and with the eval threshold in the epilogue."""
    from matryoshka_transcoder_amd.engine import HotPath
    dt = torch.bfloat16
    g = torch.Generator("cpu").manual_seed(B + A)
    hp = HotPath(A, A, [D], dt, False)
    d = hp.desc(B)
    x = torch.randn(B, A, generator=g).to(dt)
    W = (torch.randn(A, D, generator=g) / A ** 0.5).to(dt)              # reference layout [A, D]
    b = (torch.randn(D, generator=g) * 0.1).to(dt)
    xd, Wt, bd = x.to(DEV), W.t().contiguous().to(DEV), b.to(DEV)
    force_tile(1)
    small = hp.encode(d, xd, Wt, bd)
    force_tile(2)
    big = hp.encode(d, xd, Wt, bd)
    torch.cuda.synchronize()
    assert torch.equal(big, small), "256 x 256 and 128 x 128 encoders must agree bit for bit"
    want = R.encode_acts(x, W, b)
    # two roundings (GEMM result, + b_enc): every entry within 2 bf16 ulps of the fp32 contraction, 99.9 % within 1
    err = (big.cpu().double() - want.double()).abs()
    mag = want.double().abs() + b.double().abs()[None, :]
    assert bool((err <= 1e-3 + 1.6e-2 * mag).all()), f"acts (256 x 256 kernel) vs fp32 contraction: max {float(err.max()):.3e}"
    assert float((err <= 1e-3 + 8e-3 * mag).double().mean()) >= 0.999
    assert float((big.cpu() > 0).float().mean()) > 0.3          # not a degenerate all-zero comparison
    for thr in (0.4, 0.0, float("nan")):
        t = torch.tensor(thr, dtype=dt, device=DEV)
        fused = hp.encode_threshold(d, xd, Wt, bd, t)
        assert torch.equal(fused, torch.where(big > t, big, torch.zeros_like(big))), thr


@pytest.mark.parametrize("unit", [False, True])
def test_chain_on_256_tile_encoder(unit, gpu_lib, force_tile):
    """The whole kernel chain fed by the forced 256 x 256 encoder at A = 2304, ragged B, against the oracle."""
    A, D = 2304, 1208
    cfg = dict(seed=49, prefix_sizes=[8, 200, D], act_size=A, top_k=24, top_k_aux=16, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=1e-3, dtype=torch.bfloat16, device="cpu", input_unit_norm=unit, lr=3e-4,
               beta1=0.9, beta2=0.99)
    B = 300
    params = R.init_params(cfg)
    params["b_enc"] = (0.01 * torch.randn(D, generator=torch.Generator().manual_seed(3))).to(torch.bfloat16)
    g = torch.Generator("cpu").manual_seed(77)
    xs = torch.randn(B, A, generator=g).to(torch.bfloat16)
    xt = torch.randn(B, A, generator=g).to(torch.bfloat16)
    hp = _hot(cfg)
    cnt = torch.zeros(D)
    force_tile(2)
    res = run_forward(hp, cfg, params, cnt, xs, xt)
    check_against_oracle(cfg, params, cnt, xs, xt, res, hp)


# ------------------------------------------------------------------ (c) BASELINE config #2 at full size
def test_config2_full_size_encoder_and_select_against_oracle(gpu_lib):
    """B 4096, A 2304, D 36864, k 96 (what bench.py times).  (1) 320 sampled rows of the GPU's activations (every
    32nd row of the first and last batch tiles + a spread) against an fp32 CPU contraction of the same operands;
    (2) the oracle's select on the host over the GPU's own activations (kthvalue over 1.5e8 values) against the
    GPU selection, full three-pass form and the carried-bound one-pass form: indices / values / tau / tie counts
    bit-exact."""
    import bench
    from matryoshka_transcoder_amd.engine import HotPath
    dt = torch.bfloat16
    cfg, B = bench.make_cfg("gemma2-2b", dt, "cpu")
    A, D, k = cfg["act_size"], cfg["prefix_sizes"][-1], cfg["top_k"]
    params = R.init_params(cfg)
    params["b_enc"] = (0.02 * torch.randn(D, generator=torch.Generator().manual_seed(3))).to(dt)
    g = torch.Generator("cpu").manual_seed(1000)
    xs = torch.randn(B, A, generator=g).to(dt)
    hp = HotPath(A, A, cfg["prefix_sizes"], dt, False)
    d = hp.desc(B)
    Wt = params["W_enc"].t().contiguous().to(DEV)
    acts = hp.encode(d, xs.to(DEV), Wt, params["b_enc"].to(DEV))
    full = hp.select_topk(d, acts, k * B)
    car1 = hp.select_topk(d, acts, k * B, carry=True)       # first call: no bound yet (full passes, bound stored)
    car2 = hp.select_topk(d, acts, k * B, carry=True)       # second call: the one-pass compact path
    torch.cuda.synchronize()
    assert int(hp.carry_state(torch.device(DEV))[4]) >= 1, "the carried-bound path did not serve the second call"
    acts_cpu = acts.cpu()
    rows = torch.unique(torch.cat([torch.arange(0, 256, 8), torch.arange(B - 256, B, 8),
                                   torch.randint(0, B, (256,), generator=g)]))
    assert rows.numel() >= 256
    want = R.encode_acts(xs[rows], params["W_enc"], params["b_enc"])
    # one bf16 ulp of the PRE-bias value (the first of the two roundings, sae.py:173): where pre and bias partly
    # cancel, that ulp is larger than the result's own
    err = (acts_cpu[rows].double() - want.double()).abs()
    mag = want.double().abs() + params["b_enc"].double().abs()[None, :]
    # two roundings (GEMM result, then + b_enc), each may flip by one ulp (2^-7 relative at worst) when the fp32 sums
    # differ in their last bits: every entry within 2 ulps, 99.9 % within 1
    assert bool((err <= 1e-3 + 1.6e-2 * mag).all()), f"config #2 activations: max {float(err.max()):.3e}"
    assert float((err <= 1e-3 + 8e-3 * mag).double().mean()) >= 0.999
    # the bf16 results agree exactly except where the fp32 sums straddle a rounding boundary
    same = float((acts_cpu[rows] == want).float().mean())
    assert same > 0.98, same
    tau, flat_idx, n_gt, n_tie = R.global_topk_select(acts_cpu, k * B)
    vals = acts_cpu.flatten()[flat_idx]
    for name, sel in (("full", full), ("carry#1", car1), ("carry#2", car2)):
        st = sel.state.cpu()
        rp = sel.row_ptr.cpu().long()
        nnz = int(rp[-1])
        assert nnz == k * B == flat_idx.numel(), name
        r = torch.repeat_interleave(torch.arange(B), rp[1:] - rp[:-1])
        assert torch.equal(r * D + sel.col_idx.cpu()[:nnz].long(), flat_idx), f"{name}: flat indices differ from the oracle's"
        assert torch.equal(sel.vals.cpu()[:nnz], vals), name
        assert (int(st[4]) & 0xffffffff) == (int(tau.float().view(torch.int32)) & 0xffffffff), name
        assert int(st[3]) == n_gt and int(st[5]) == n_tie, name
    assert n_tie > 0          # bf16 at this size: hundreds of values share tau


# ------------------------------------------------------------------ (d) the reference-made medium fixtures
@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_medium_fixture_through_hip_path(tag, gpu_lib):
    """tests/golden/medium_*.pt (A 768, D 12288, B 64, k 64; produced by the reference itself): selected indices
    outside the tau tie class, tau, counts above tau, the output scalars, gradient checksums and the zero-row
    pattern of the gradients, from the HIP path."""
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    fx, cfg = load_golden(f"medium_{tag}")
    dt = cfg["dtype"]
    f32 = dt == torch.float32
    B, D = int(fx["B"]), cfg["prefix_sizes"][-1]
    m = MatryoshkaTranscoder(dict(cfg, device=DEV))
    for key, v in m.state_dict().items():
        if key != "threshold":
            assert math.isclose(float(v.double().sum()), float(fx["init_checksum"][key]), rel_tol=1e-9, abs_tol=1e-9)
    g = torch.Generator("cpu").manual_seed(1000)
    xs = torch.randn(B, 768, generator=g).to(dt)
    xt = torch.randn(B, 768, generator=g).to(dt)
    out = m(xs.to(DEV), xt.to(DEV))
    sel = m._last_res.sel
    rp = sel.row_ptr.cpu().long()
    nnz = int(rp[-1])
    ref_idx = fx["topk_indices_sorted"]
    assert nnz == ref_idx.numel() == cfg["top_k"] * B
    rows = torch.repeat_interleave(torch.arange(B), rp[1:] - rp[:-1])
    ours = rows * D + sel.col_idx.cpu()[:nnz].long()
    vals = dict(zip(ours.tolist(), sel.vals.cpu()[:nnz].float().tolist()))
    diff = set(ours.tolist()) ^ set(ref_idx.tolist())
    tau = float(fx["tau"])
    if f32:
        assert len(diff) <= 2, diff            # fp32: tie class of size 1; allow one pair swapped by summation order
    else:
        assert len(diff) <= 0.1 * nnz
    hp = m.hot_path()
    acts = hp.encode(m._last_res.desc, xs.to(DEV), m.W_enc.t(), m.b_enc).flatten().float().cpu()
    for i in diff:                              # every disagreement sits within one ulp of tau
        assert abs(float(acts[i]) - tau) <= tau * (2 ** -7 if not f32 else 1e-5), (i, float(acts[i]), tau)
    tau_gpu = torch.tensor([int(sel.state.cpu()[4])], dtype=torch.int32).view(torch.float32)
    assert abs(float(tau_gpu) - tau) <= tau * (2 ** -7 if not f32 else 1e-6)
    if f32:
        assert int(sel.state.cpu()[3]) == int(fx["n_gt_tau"])
    rtol, atol = tol(dt)
    for key in ("loss", "l2_loss", "l0_norm", "fvu", "fvu_min", "fvu_max", "fvu_mean", "l1_norm"):
        _close(out[key].float(), fx["out"][key].float(), rtol, atol, key)
        assert out[key].dtype == fx["out"][key].dtype, key
    _close(m.threshold.float(), fx["threshold_after"].float(), rtol, atol, "threshold")
    a, b = float(out["transcoder_out"].detach().double().sum()), float(fx["out_checksum"])
    assert abs(a - b) <= (1e-3 if f32 else 2.0) + 1e-4 * abs(b), ("transcoder_out checksum", a, b)
    out["loss"].backward()
    for n, p in m.named_parameters():
        a, b = float(p.grad.double().abs().sum()), float(fx["grad_abs_checksum"][n])
        assert math.isclose(a, b, rel_tol=1e-4 if f32 else 2e-2), (n, a, b)
    if f32:
        _close(m.b_dec.grad, fx["db_dec"], 1e-4, 1e-7, "db_dec")
    else:
        # bf16: ~1 % of the selections differ inside the tau tie class (torch.topk's pick is not lowest-index), which
        # moves whole reconstruction rows; the column sums are compared in norm
        a, b = m.b_dec.grad.double().cpu(), fx["db_dec"].double()
        assert float((a - b).norm() / b.norm()) <= 5e-2, float((a - b).norm() / b.norm())
    if f32:
        _close(m.b_enc.grad, fx["db_enc"], 1e-4, 1e-7, "db_enc")
        assert torch.equal((m.W_dec.grad.abs().sum(-1) > 0).cpu(), fx["grad_row_nonzero"]["W_dec"])
        assert torch.equal((m.W_enc.grad.abs().sum(0) > 0).cpu(), fx["grad_row_nonzero"]["W_enc"])


# ------------------------------------------------------------------ (e) BASELINE configs #4 / #5 at full size
@pytest.mark.parametrize("workload", ["gemma2-9b", "gemma2-27b", "gemma2-27b-2x"])
def test_full_size_properties_large_models(workload, gpu_lib):
    """BASELINE configs #4 (A 3584, D 57344, k 128, B 8192, 6 groups) and #5 (A 4608, D 73728, k 184, B 16384; and its
    2x stress dictionary D 147456: 15 GB of parameters + moments + activations) at full size on one GPU: size-independent invariants of the select / decode / step through the NC = 7 / 9 kernels, and the
    fused step against the autograd path (bit-identical weights)."""
    import bench
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    dt = torch.bfloat16
    cfg, B = bench.make_cfg(workload, dt, DEV)
    D, k, A = cfg["prefix_sizes"][-1], cfg["top_k"], cfg["act_size"]
    m = MatryoshkaTranscoder(cfg)
    opt = FusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    pool = bench.batch_pool(B, A, dt, DEV, n=2)
    losses = []
    for i in range(3):
        xs, xt = pool[i % 2]
        if i == 1:
            # keep a copy of the state, take this step through the fused path, replay it through autograd below
            snap = {n: p.detach().clone() for n, p in m.named_parameters()}
            snap_opt = {n: (opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone(), opt.state[p]["step"])
                        for n, p in m.named_parameters()}
            snap_thr, snap_cnt = m.threshold.clone(), m.num_batches_not_active.clone()
            out = opt.train_step(xs, xt)
            fused = {n: p.detach().clone() for n, p in m.named_parameters()}
            with torch.no_grad():
                for n, p in m.named_parameters():
                    p.copy_(snap[n])
                    st = opt.state[p]
                    st["exp_avg"].copy_(snap_opt[n][0]); st["exp_avg_sq"].copy_(snap_opt[n][1]); st["step"] = snap_opt[n][2]
                m.threshold.copy_(snap_thr)
            m.num_batches_not_active = snap_cnt
            del snap, snap_opt
        out = m(xs, xt)
        res = m._last_res
        rp = res.sel.row_ptr.cpu().long()
        nnz = int(rp[-1])
        assert nnz == k * B and float(out["l0_norm"]) == k
        cols = res.sel.col_idx.cpu()[:nnz].long()
        rows = torch.repeat_interleave(torch.arange(B), rp[1:] - rp[:-1])
        key = rows * D + cols
        assert bool((key[1:] > key[:-1]).all())                      # sorted, unique flat indices
        vals = res.sel.vals.cpu()[:nnz].float()
        tau = torch.tensor([int(res.sel.state.cpu()[4])], dtype=torch.int32).view(torch.float32)
        assert float(vals.min()) == float(tau) and bool((vals > 0).all())
        # exact selection on a block of rows: everything above tau is in, nothing below tau is
        with torch.no_grad():
            acts = m.hot_path().encode(res.desc, xs, m.W_enc.t(), m.b_enc)
        blk = acts[:64].float().cpu()
        chosen = torch.zeros(64, D, dtype=torch.bool)
        in_blk = rows < 64
        chosen[rows[in_blk], cols[in_blk]] = True
        assert bool(chosen[blk > float(tau)].all()) and not bool(chosen[blk < float(tau)].any())
        del acts
        # one sampled reconstruction row against a host gather of the selected decoder rows
        r0 = 17
        s, e = int(rp[r0]), int(rp[r0 + 1])
        want = (vals[s:e, None] * m.W_dec.detach()[cols[s:e].to(DEV)].float().cpu()).sum(0) + m.b_dec.detach().float().cpu()
        _close(out["transcoder_out"][r0].float(), want, 2e-2, 2e-2, "x_hat row vs host gather")
        out["loss"].backward()
        fired = torch.zeros(D, dtype=torch.bool); fired[cols] = True
        assert bool((m.W_dec.grad[~fired.to(DEV)] == 0).all())       # silent features: zero gradient rows
        assert bool((m.W_enc.grad[:, ~fired.to(DEV)] == 0).all())
        opt.step(project=True)
        opt.zero_grad()
        if i == 1:
            for n, p in m.named_parameters():
                assert torch.equal(p.detach(), fused[n]), f"{n}: fused step != autograd step at {workload} size"
            del fused
        losses.append(float(out["loss"].detach().float()))
        assert float(out["fvu_min"]) <= float(out["fvu_mean"]) <= float(out["fvu_max"])
    norms = m.W_dec.detach().float().norm(dim=-1)
    assert float((norms - 1).abs().max()) < 2e-2
    assert losses[-1] < losses[0]
