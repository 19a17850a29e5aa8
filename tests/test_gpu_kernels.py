"""GPU parity: every libmtc kernel (through the C ABI) against the CPU oracle on the same inputs.
Bar: indices bit-exact given identical activations; floating point within the tolerances in the
asserts (fp32 rel 1e-4, bf16 rel 2e-2 / 1 ulp)."""
import pytest
import torch

from conftest import SMALL_CASES, load_golden
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _hot(cfg):
    from matryoshka_transcoder_amd.engine import HotPath
    a_s = cfg.get("source_act_size", cfg["act_size"])
    a_t = cfg.get("target_act_size", cfg["act_size"])
    return HotPath(a_s, a_t, cfg["prefix_sizes"], cfg["dtype"], cfg.get("input_unit_norm", False))


def _close(a, b, rtol, atol, msg=""):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (msg, a.shape, b.shape)
    err = (a - b).abs()
    ok = err <= atol + rtol * b.abs()
    assert bool(ok.all()), f"{msg}: {int((~ok).sum())} of {ok.numel()} off, max abs {err.max():.3e}"


def run_forward(hp, cfg, params, cnt, xs, xt, want_recons=True):
    """The single-GPU kernel chain for one training forward + backward (no aux)."""
    dev = DEV
    B = xs.shape[0]
    d = hp.desc(B)
    W_encT = params["W_enc"].t().contiguous().to(dev)
    b_enc, W_dec, b_dec = (params[k].to(dev) for k in ("b_enc", "W_dec", "b_dec"))
    thr = params["threshold"].clone().to(dev)
    x, t = xs.to(dev), xt.to(dev)
    if hp.unit_norm:
        xn, mean, std = hp.normalize(d, x)
    else:
        xn, mean, std = x, None, None
    acts = hp.encode(d, xn, W_encT, b_enc)
    sel = hp.select_topk(d, acts, cfg["top_k"] * B)
    hp.threshold_ema(d, sel, thr)
    cnt_dev = cnt.clone().to(dev)
    hp.build_csc(d, sel, cnt_dev)
    l1_term = cfg["l1_coeff"] / (B * hp.D)
    x_hat, g, row_stats, recons, d_vals = hp.decode(d, sel, W_dec, b_dec, t, mean, std, want_g=True, want_dvals=True,
                                                    l1_term=l1_term, want_recons=want_recons)
    scalars = hp.reduce_scalars(d, row_stats, sel, hp.D, cfg["l1_coeff"], cfg["aux_penalty"])
    grads = hp.backward(d, sel, g, d_vals, xn, None)
    torch.cuda.synchronize()
    return dict(xn=xn, mean=mean, std=std, acts=acts, sel=sel, thr=thr, cnt=cnt_dev, x_hat=x_hat, g=g,
                row_stats=row_stats, recons=recons, d_vals=d_vals, scalars=scalars, grads=grads)


def check_against_oracle(cfg, params, cnt, xs, xt, res, hp):
    dt = cfg["dtype"]
    f32 = dt == torch.float32
    B = xs.shape[0]
    D = hp.D
    acts_gpu = res["acts"].cpu()
    out, new, inter = R.forward(params, {"num_batches_not_active": cnt.clone()}, xs, xt, cfg, acts_override=acts_gpu)
    o_free, _, i_free = R.forward(params, {"num_batches_not_active": cnt.clone()}, xs, xt, cfg)
    # a2
    if hp.unit_norm:
        _close(res["xn"], inter["x_norm"], 0, 1e-6 if f32 else 0, "x_norm") if f32 else \
            _close(res["xn"], inter["x_norm"], 8e-3, 1e-6, "x_norm")
        _close(res["mean"], inter["mean"].flatten(), 1e-5 if f32 else 8e-3, 1e-6, "mean")
        _close(res["std"], inter["std"].flatten(), 1e-5 if f32 else 8e-3, 1e-6, "std")
    # a3 dense part: GEMM vs the oracle's own fp32 contraction (different summation order: the fp32 bound grows with
    # the square root of the contraction length; 1e-5 is the bar at K <= 768)
    ks = max(1.0, (xs.shape[1] / 768.0) ** 0.5)
    _close(acts_gpu, i_free["acts"], 1e-5 * ks if f32 else 8e-3, 1e-5 * ks if f32 else 1e-3, "acts")
    # a3 select: BIT-EXACT indices and values given identical activations
    sel = res["sel"]
    st = sel.state.cpu()
    nnz = int(st[7])
    rp = sel.row_ptr.cpu()
    assert int(rp[-1]) == nnz == inter["flat_idx"].numel()
    cols = sel.col_idx.cpu()[:nnz].long()
    rows = torch.repeat_interleave(torch.arange(B), (rp[1:] - rp[:-1]).long())
    flat = rows * D + cols
    assert torch.equal(flat, inter["flat_idx"]), "selected flat indices differ from the oracle's"
    assert torch.equal(sel.vals.cpu()[:nnz], inter["vals"])
    assert torch.equal(rp, inter["row_ptr"])
    tau_bits = int(st[4]) & 0xffffffff
    assert tau_bits == int(inter["tau"].float().view(torch.int32)) & 0xffffffff
    assert int(st[3]) == inter["n_gt"] and int(st[5]) == inter["n_tie"]
    # a5
    _close(res["thr"], new["threshold"], 1e-6 if f32 else 8e-3, 0, "threshold")
    # a7
    assert torch.equal(res["cnt"].cpu(), new["num_batches_not_active"])
    # CSC is a permutation of CSR, sorted by (feature, row)
    cp = sel.col_ptr.cpu().long()
    assert int(cp[-1]) == nnz
    crow, cpos = sel.csc_row.cpu()[:nnz].long(), sel.csc_pos.cpu()[:nnz].long()
    assert torch.equal(torch.sort(cpos).values, torch.arange(nnz))
    feat = torch.repeat_interleave(torch.arange(D), cp[1:] - cp[:-1])
    assert torch.equal(cols[cpos], feat) and torch.equal(rows[cpos], crow)
    key = feat * B + crow
    assert bool((key[1:] > key[:-1]).all()), "CSC must be ordered by (feature, batch row)"
    # a6 / a10
    for i in range(len(cfg["prefix_sizes"])):
        _close(res["recons"][i], inter["recons"][i], 1e-5 if f32 else 8e-3, 1e-5 if f32 else 2e-3, f"recon {i}")
    _close(res["x_hat"], inter["recons"][-1], 1e-5 if f32 else 8e-3, 1e-5 if f32 else 2e-3, "x_hat")
    G = len(cfg["prefix_sizes"])
    _close(res["row_stats"][:G].sum(1), inter["sse"], 1e-4 if f32 else 2e-2, 1e-6, "sse")
    s = res["scalars"].cpu()
    names = {"l2_loss": 0, "l1_loss": 1, "l0_norm": 2, "l1_norm": 3, "aux_loss": 4, "loss": 5, "fvu": 6, "fvu_min": 7,
             "fvu_max": 8, "fvu_mean": 9}
    for k, i in names.items():
        _close(s[i], out[k].float(), 1e-4 if f32 else 2e-2, 1e-6, k)
    _close(s[10], inter["var_target"], 1e-5, 1e-7, "var(x_t)")
    # a11
    grads = R.backward(params, xs, xt, cfg, inter)
    _close(res["g"], grads["g"], 1e-4 if f32 else 2e-2, 1e-7 if f32 else 1e-5, "g")
    if not f32 and inter["aux"] is None:
        # bf16: g follows autograd's rounding sequence (inv_n as a bf16 scalar, difference, product, de-normalisation each
        # rounded), which the oracle reproduces bit for bit against the reference's fixtures -- so wherever the HIP
        # reconstruction equals the oracle's (it differs only by the summation order of the gathered rows), g must be EQUAL
        same = res["x_hat"].cpu() == inter["recons"][-1]
        assert float(same.float().mean()) > 0.9
        assert torch.equal(res["g"].cpu()[same], grads["g"].to(dt)[same]), "g differs where x_hat is identical"
    _close(res["d_vals"][:nnz], grads["d_vals"], 1e-4 if f32 else 2e-2, 1e-7 if f32 else 1e-5, "d_vals")
    dW_dec, dW_encT, db_enc, db_dec = res["grads"]
    rt, at = (1e-4, 1e-7) if f32 else (2e-2, 1e-4)
    _close(dW_dec, grads["W_dec"], rt, at, "dW_dec")
    _close(dW_encT.t(), grads["W_enc"], rt, at, "dW_enc")
    _close(db_enc, grads["b_enc"], rt, at, "db_enc")
    _close(db_dec, grads["b_dec"], rt, at * 10, "db_dec")
    if f32:
        assert torch.equal(dW_dec.cpu() != 0, grads["W_dec"] != 0)
    return out, inter


@pytest.mark.parametrize("name", [c for c in SMALL_CASES if "aux" not in c])
def test_small_golden_shapes(name, gpu_lib):
    fx, cfg = load_golden(name)
    hp = _hot(cfg)
    params = {k: v.clone() for k, v in fx["init"].items()}
    s0 = fx["steps"][0]
    res = run_forward(hp, cfg, params, fx["cnt0"], s0["x_source"], s0["x_target"])
    check_against_oracle(cfg, params, fx["cnt0"], s0["x_source"], s0["x_target"], res, hp)
    # and against the reference's own numbers from the fixture
    s = res["scalars"].cpu()
    f32 = cfg["dtype"] == torch.float32
    for k, i in {"l2_loss": 0, "l0_norm": 2, "fvu": 6, "fvu_mean": 9}.items():
        _close(s[i], s0["out"][k].float(), 1e-4 if f32 else 2e-2, 1e-6, f"{k} vs reference fixture")


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("unit", [False, True])
def test_mid_shape_mfma_path(dt, unit, gpu_lib):
    """A=768 (multiple of 64 -> MFMA encoder for bf16), D=12288, GPT-2 prefixes, B=192 (ragged tiles)."""
    cfg = dict(seed=49, prefix_sizes=[768, 2304, 5376, 12288], act_size=768, top_k=64, top_k_aux=512,
               aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=0.0 if unit else 1e-3, dtype=dt, device="cpu",
               input_unit_norm=unit, lr=3e-4, beta1=0.9, beta2=0.99)
    B = 192
    params = R.init_params(cfg)
    params["b_enc"] = (0.01 * torch.randn(12288, generator=torch.Generator().manual_seed(3))).to(dt)
    params["b_dec"] = (0.01 * torch.randn(768, generator=torch.Generator().manual_seed(4))).to(dt)
    g = torch.Generator("cpu").manual_seed(1000)
    xs = torch.randn(B, 768, generator=g).to(dt)
    xt = torch.randn(B, 768, generator=g).to(dt)
    hp = _hot(cfg)
    cnt = torch.zeros(12288)
    res = run_forward(hp, cfg, params, cnt, xs, xt)
    check_against_oracle(cfg, params, cnt, xs, xt, res, hp)


def test_fewer_positives_than_k(gpu_lib):
    """Edge: fewer than k*B positive activations -> every positive is selected, zeros never are."""
    cfg = dict(seed=1, prefix_sizes=[16, 48, 112, 256], act_size=32, top_k=200, top_k_aux=8, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.float32, device="cpu", input_unit_norm=False)
    params = R.init_params(cfg)
    params["b_enc"] = torch.full((256,), -0.5)
    g = torch.Generator("cpu").manual_seed(5)
    xs, xt = torch.randn(16, 32, generator=g), torch.randn(16, 32, generator=g)
    hp = _hot(cfg)
    cnt = torch.zeros(256)
    res = run_forward(hp, cfg, params, cnt, xs, xt)
    out, inter = check_against_oracle(cfg, params, cnt, xs, xt, res, hp)
    assert inter["flat_idx"].numel() < 200 * 16
    assert int(res["sel"].state.cpu()[2]) == 1      # all_pos


def test_heavy_ties_bf16(gpu_lib):
    """Edge: thousands of activations equal to tau (bf16) -> ties taken in ascending flat index order."""
    cfg = dict(seed=2, prefix_sizes=[64, 128, 256, 512], act_size=64, top_k=8, top_k_aux=8, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.bfloat16, device="cpu", input_unit_norm=False)
    params = R.init_params(cfg)
    g = torch.Generator("cpu").manual_seed(6)
    # inputs on a coarse grid so that many pre-activations collide exactly
    xs = (torch.randint(-2, 3, (128, 64), generator=g).float() * 0.5).to(torch.bfloat16)
    params["W_enc"] = (torch.randint(-2, 3, (64, 512), generator=g).float() * 0.125).to(torch.bfloat16)
    xt = torch.randn(128, 64, generator=g).to(torch.bfloat16)
    hp = _hot(cfg)
    cnt = torch.zeros(512)
    res = run_forward(hp, cfg, params, cnt, xs, xt)
    out, inter = check_against_oracle(cfg, params, cnt, xs, xt, res, hp)
    acts = res["acts"].cpu().float().flatten()
    n_eq = int((acts == inter["tau"].float()).sum())
    assert n_eq > inter["n_tie"] > 0, (n_eq, inter["n_tie"])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_unitnorm_adam_against_oracle(dt, gpu_lib):
    from matryoshka_transcoder_amd.engine import HotPath, adam_hyper
    D, A = 512, 96
    hp = HotPath(A, A, [128, 512], dt, False)
    gen = torch.Generator().manual_seed(11)
    W = torch.randn(D, A, generator=gen).to(dt)
    m = (0.01 * torch.randn(D, A, generator=gen)).to(dt)
    v = (1e-4 * torch.rand(D, A, generator=gen)).to(dt)
    grad = (0.01 * torch.randn(D, A, generator=gen)).to(dt)
    grad[5] = 0
    Wn, gp = R.unit_norm_project(W, grad)
    step = 3
    p_ref, m_ref, v_ref = R.adam_step(Wn, gp, m, v, step, 3e-4, 0.9, 0.99)
    Wd, gd, md, vd = (t.clone().to(DEV) for t in (W, grad, m, v))
    hp.unitnorm_adam(Wd, gd, md, vd, adam_hyper(3e-4, 0.9, 0.99, 1e-8, step))
    # projection alone
    W2, g2 = W.clone().to(DEV), grad.clone().to(DEV)
    hp.unitnorm_project(W2, g2)
    torch.cuda.synchronize()
    rt, at = (1e-5, 1e-7) if dt == torch.float32 else (8e-3, 1e-4)
    _close(W2, Wn, rt, at, "normed")
    _close(g2, gp, rt * 4, at, "projected grad")
    _close(gd, gp, rt * 4, at, "projected grad (fused)")
    _close(md, m_ref, rt * 2, at, "exp_avg")
    _close(vd, v_ref, rt * 2, at * 1e-2, "exp_avg_sq")
    _close(Wd, p_ref, rt * 2, at, "W after step")
    # flat Adam
    n = 4096
    p = torch.randn(n, generator=gen).to(dt)
    gflat = (0.01 * torch.randn(n, generator=gen)).to(dt)
    m0 = torch.zeros(n).to(dt); v0 = torch.zeros(n).to(dt)
    pr, mr, vr = R.adam_step(p, gflat, m0, v0, 1, 3e-4, 0.9, 0.99)
    pd, gdv, mdv, vdv = (t.clone().to(DEV) for t in (p, gflat, m0, v0))
    hp.adam_flat(pd, gdv, mdv, vdv, adam_hyper(3e-4, 0.9, 0.99, 1e-8, 1))
    torch.cuda.synchronize()
    _close(pd, pr, rt, at, "flat p"); _close(mdv, mr, rt * 2, at, "flat m"); _close(vdv, vr, rt * 2, at * 1e-2, "flat v")


@pytest.mark.parametrize("aux_mode", ["gemm", "sparse"])
@pytest.mark.parametrize("name", ["fp32_aux", "fp32_aux_unit", "bf16_aux"])
def test_aux_path_against_oracle(name, aux_mode, gpu_lib):
    """a9: dead list, per-row top-k_aux, aux reconstruction / loss, and every gradient with aux active -- in the dense
    form over the dead set (GEMMs; the single-GPU default) and in the row-gather form (what the sharded model runs)."""
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    fx, cfg = load_golden(name)
    dt = cfg["dtype"]
    f32 = dt == torch.float32
    m = MatryoshkaTranscoder(dict(cfg, device=DEV, aux_mode=aux_mode))
    m.load_state_dict({k: v.to(DEV) for k, v in fx["init"].items()})
    m.num_batches_not_active = fx["cnt0"].clone().to(DEV)
    s0 = fx["steps"][0]
    xs, xt = s0["x_source"], s0["x_target"]
    out = m(xs.to(DEV), xt.to(DEV))
    res = m._last_res
    assert res.aux is not None
    params = {k: v.clone() for k, v in fx["init"].items()}
    acts_gpu = m.hot_path().encode(res.desc, res.x_norm, m.W_enc.t(), m.b_enc).cpu()
    o, new, inter = R.forward(params, {"num_batches_not_active": fx["cnt0"].clone()}, xs, xt, cfg, acts_override=acts_gpu)
    aux = inter["aux"]
    B, D = xs.shape[0], cfg["prefix_sizes"][-1]
    if aux_mode == "gemm":
        a = res.aux
        n = a["dead"].n()
        idx = a["dead"].dead_idx[:n].cpu().long()
        assert a["dense"] and not bool(a["acts_aux"][:, n:].any())
        assert torch.equal(a["dead"].row_map.cpu()[idx], torch.arange(n, dtype=torch.int32))       # feature -> slot map
        assert int((a["dead"].row_map >= 0).sum()) == n and not bool(a["dead"].dead_idx[n:].any())  # gatherable tail
        nzr, nzc = torch.nonzero(a["acts_aux"][:, :n].cpu(), as_tuple=True)
        got = set((nzr * D + idx[nzc]).tolist())
        assert len(got) == aux["rows"].numel()
    else:
        asel = res.aux["sel"]
        rp = asel.row_ptr.cpu().long()
        n = int(rp[-1])
        assert n == aux["rows"].numel()
        rows = torch.repeat_interleave(torch.arange(B), rp[1:] - rp[:-1])
        got = set((rows * D + asel.col_idx.cpu()[:n].long()).tolist())
    want = set((aux["rows"] * D + aux["cols"]).tolist())
    assert got == want, "aux selection differs from the oracle's"
    rt, at = (1e-4, 1e-7) if f32 else (2e-2, 1e-4)
    _close(out["aux_loss"], o["aux_loss"], rt, at, "aux_loss")
    _close(out["loss"], o["loss"], rt, at, "loss")
    assert out["loss"].dtype == torch.float32 and out["aux_loss"].dtype == torch.float32
    out["loss"].backward()
    grads = R.backward(params, xs, xt, cfg, inter)
    for k in ("W_enc", "W_dec", "b_enc", "b_dec"):
        _close(getattr(m, k).grad, grads[k], rt, at * (10 if k == "b_dec" else 1), f"grad {k}")
    # and against the reference's own autograd (fixture)
    if f32:
        for k in ("W_enc", "W_dec", "b_enc", "b_dec"):
            _close(getattr(m, k).grad, s0["grads"][k], 1e-4, 1e-7, f"grad {k} vs reference")


def test_dense_features_long_csc_segments(gpu_lib):
    """Edge: a few features fire on every row (segments > 64 entries): long-segment CSC ordering and the
    per-feature gradient rows with hundreds of contributions."""
    cfg = dict(seed=3, prefix_sizes=[64, 128, 256, 512], act_size=64, top_k=8, top_k_aux=8, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=torch.float32, device="cpu", input_unit_norm=False)
    params = R.init_params(cfg)
    params["b_enc"][[3, 200, 511]] = 5.0
    g = torch.Generator("cpu").manual_seed(8)
    xs, xt = torch.randn(320, 64, generator=g), torch.randn(320, 64, generator=g)
    hp = _hot(cfg)
    cnt = torch.zeros(512)
    res = run_forward(hp, cfg, params, cnt, xs, xt)
    check_against_oracle(cfg, params, cnt, xs, xt, res, hp)
    cp = res["sel"].col_ptr.cpu()
    assert int((cp[1:] - cp[:-1]).max()) == 320


def test_build_csc_in_two_parts_equals_one_call(gpu_lib):
    """mtc_build_csc_part 1 + 2 (the aux-active step puts the dead list between them) = mtc_build_csc: same index, same
    counters, and the counters are already final after part 1."""
    from matryoshka_transcoder_amd.engine import HotPath
    torch.manual_seed(3)
    B, D, k = 96, 2048, 12
    hp = HotPath(64, 64, [D], torch.float32, False)
    d = hp.desc(B)
    acts = torch.relu(torch.randn(B, D, device="cuda"))
    cnt0 = torch.randint(0, 5, (D,), device="cuda").float()
    sel_a = hp.select_topk(d, acts, k * B)
    sel_b = hp.select_topk(d, acts, k * B)
    cnt_a, cnt_b = cnt0.clone(), cnt0.clone()
    hp.build_csc(d, sel_a, cnt_a)
    hp.build_csc(d, sel_b, cnt_b, part=1)
    torch.cuda.synchronize()
    assert torch.equal(cnt_a, cnt_b) and torch.equal(sel_a.col_ptr, sel_b.col_ptr)
    hp.build_csc(d, sel_b, cnt_b, part=2)
    torch.cuda.synchronize()
    assert torch.equal(cnt_a, cnt_b)
    assert torch.equal(sel_a.csc_row, sel_b.csc_row) and torch.equal(sel_a.csc_pos, sel_b.csc_pos)


# ------------------------------------------------------------------ carried lower bound of tau (training loop select)
def _sel_tuple(sel, nnz):
    return (sel.state.clone(), sel.row_ptr.clone(), sel.col_idx[:nnz].clone(), sel.vals[:nnz].clone())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_select_carry_is_identical_to_full_select(dt, gpu_lib):
    """mtc_select_global_topk_carry must return exactly what mtc_select_global_topk returns: on the first
    call (no bound -> full passes), with a good bound (compact path), with a bound that is too high
    (too few candidates -> full passes) and with a row that overflows the compact matrix."""
    from matryoshka_transcoder_amd.engine import HotPath
    torch.manual_seed(5)
    B, D, k = 64, 4096, 24
    hp = HotPath(64, 64, [D], dt, False)
    d = hp.desc(B)

    def run_pair(acts):
        full = hp.select_topk(d, acts, k * B)
        car = hp.select_topk(d, acts, k * B, carry=True)
        torch.cuda.synchronize()
        nnz = int(full.row_ptr[-1])
        assert nnz == k * B
        for a, b, name in zip(_sel_tuple(car, nnz), _sel_tuple(full, nnz), ("state", "row_ptr", "col_idx", "vals")):
            if name == "state":
                a, b = a[:8], b[:8]
            assert torch.equal(a, b), name

    carry = hp.carry_state(torch.device("cuda:0"))
    acts = torch.relu(torch.randn(B, D, device="cuda")).to(dt)
    run_pair(acts)                                      # 1: no bound yet
    assert carry[4].item() == 0 and carry[5].item() == 1 and carry[0].item() != 0
    run_pair(acts)                                      # 2: bound from call 1 -> compact path
    assert carry[4].item() == 1
    acts2 = (acts.float() * 1.01).to(dt)                # 3: slightly different batch, still the compact path
    run_pair(acts2)
    assert carry[4].item() == 2
    if dt == torch.bfloat16:                            # heavy ties at tau inside the compact matrix
        acts3 = (acts2.float() * 16).round().div(16).to(dt)
        carry[0] = 0
        run_pair(acts3); run_pair(acts3)
        assert carry[4].item() == 3
        # tau rises ~75 bf16 ulps in one step: every entry >= tau is a candidate, but tau lies beyond the 31-ulp window the
        # fast path resolves -> full passes; the next call, with the refreshed bound, is served by the window again
        acts4 = (acts2.float() * 1.5).to(dt)
        run_pair(acts2)
        served = carry[4].item()
        run_pair(acts4)
        assert carry[4].item() == served and carry[2].item() >= k * B and carry[3].item() == 0
        run_pair(acts4)
        assert carry[4].item() == served + 1
    served = carry[4].item()
    run_pair((acts.float() * 0.25).to(dt))              # 4: bound far too high -> fewer than k candidates -> full
    assert carry[4].item() == served
    # 5: one row holds more candidates than the compact matrix has columns (1024) -> full passes
    acts5 = acts.clone()
    acts5[3] = (torch.rand(D, device="cuda") + 10).to(dt)
    hp.select_topk(d, acts5, k * B, carry=True)         # refresh the bound on this data (may or may not be compact)
    served = carry[4].item()
    run_pair(acts5)
    assert carry[4].item() == served
    # 6: fewer positives than k: bound disabled afterwards
    acts6 = torch.zeros(B, D, device="cuda", dtype=dt)
    acts6[:, :4] = 1.0
    full = hp.select_topk(d, acts6, k * B)
    car = hp.select_topk(d, acts6, k * B, carry=True)
    assert torch.equal(full.row_ptr, car.row_ptr) and int(car.row_ptr[-1]) == 4 * B
    assert carry[0].item() == 0


@pytest.mark.parametrize("B,D,k,iters", [(256, 4096, 16, 120), (4096, 36864, 96, 12)])
def test_select_carry_fallback_stress(B, D, k, iters, gpu_lib):
    """The fallback of the bf16 carried-bound select hands its per-row counts from every workgroup to the one whose
    ticket comes last (agent-scope release / acquire inside one launch).  Random rescaling of the batch between calls
    makes the bound fail in both directions again and again -- too few candidates, tau beyond the window, and in between
    calls the window serves -- at a small shape and at config #2's (256 workgroups spread over all XCDs); every call must
    equal the full select bit for bit."""
    from matryoshka_transcoder_amd.engine import HotPath
    g = torch.Generator("cpu").manual_seed(B + iters)
    hp = HotPath(64, 64, [D], torch.bfloat16, False)
    d = hp.desc(B)
    carry = hp.carry_state(torch.device("cuda:0"))
    carry.zero_()
    base = torch.relu(torch.randn(B, D, device="cuda"))
    scales = (torch.rand(iters, generator=g) * 1.5 + 0.5).tolist()
    for i in range(iters):
        if i % 3 == 2:
            scales[i] = scales[i - 1] * 1.01             # a small move: served by the window
    for i, sc in enumerate(scales):
        acts = (base * sc).to(torch.bfloat16)
        full = hp.select_topk(d, acts, k * B)
        car = hp.select_topk(d, acts, k * B, carry=True)
        torch.cuda.synchronize()
        nnz = int(full.row_ptr[-1])
        assert nnz == k * B
        for a, b, name in zip(_sel_tuple(car, nnz), _sel_tuple(full, nnz), ("state", "row_ptr", "col_idx", "vals")):
            assert torch.equal(a[:8] if name == "state" else a, b[:8] if name == "state" else b), (i, sc, name)
    served, fell = int(carry[4]), int(carry[5])
    assert served + fell == iters and served >= iters // 4 and fell >= iters // 4, (served, fell)


@pytest.mark.parametrize("B", [1000, 4096, 4104])
def test_select_carry_window_tail_row_counts(B, gpu_lib):
    """bf16 fast path at batch sizes on both sides of the register-resident tail's limit (4 rows per thread): B = 4104
    runs the looping form of the same launch.  Identical to the full select, served by the window from the second call."""
    from matryoshka_transcoder_amd.engine import HotPath
    torch.manual_seed(B)
    D, k = 512, 6
    hp = HotPath(64, 64, [D], torch.bfloat16, False)
    d = hp.desc(B)
    carry = hp.carry_state(torch.device("cuda:0"))
    carry.zero_()
    base = torch.relu(torch.randn(B, D, device="cuda"))
    for i, scale in enumerate((1.0, 1.02, 0.99, 1.0)):
        acts = (base * scale).to(torch.bfloat16)
        full = hp.select_topk(d, acts, k * B)
        car = hp.select_topk(d, acts, k * B, carry=True)
        torch.cuda.synchronize()
        nnz = int(full.row_ptr[-1])
        assert nnz == k * B
        for a, b, name in zip(_sel_tuple(car, nnz), _sel_tuple(full, nnz), ("state", "row_ptr", "col_idx", "vals")):
            assert torch.equal(a[:8] if name == "state" else a, b[:8] if name == "state" else b), (i, name)
        assert carry[4].item() == i                     # call 0 has no bound; every later one is served by the window


@pytest.mark.parametrize("dt,A", [(torch.bfloat16, 768), (torch.float32, 512), (torch.bfloat16, 2304),
                                  # NC = 3 / 7 / 9 / 16 chunks per lane (Gemma-2-9B: 3584, Gemma-2-27B: 4608)
                                  (torch.bfloat16, 1536), (torch.bfloat16, 3584), (torch.float32, 3584),
                                  (torch.bfloat16, 4608), (torch.float32, 4608), (torch.bfloat16, 8192)])
def test_fused_backward_adam_equals_backward_then_adam(dt, A, gpu_lib):
    """The fused per-row {gradient -> [projection] -> Adam} kernels (and d_vals produced by the decoder kernel)
    against the materialised path:
    mtc_backward_sparse, then mtc_unitnorm_adam_rows / mtc_adam_flat.  Same arithmetic -> identical bits."""
    from matryoshka_transcoder_amd.engine import HotPath, adam_hyper
    torch.manual_seed(11)
    D, B, k = 1208, 96, 12           # D/8 = 151 features per XCD: not a multiple of 4
    hp = HotPath(A, A, [200, 1208], dt, False)
    d = hp.desc(B)
    dev = "cuda"
    xs = torch.randn(B, A).to(dt).to(dev); xt = torch.randn(B, A).to(dt).to(dev)
    W_encT = (torch.randn(D, A) * 0.05).to(dt).to(dev); W_dec = (torch.randn(D, A) * 0.05).to(dt).to(dev)
    b_enc = (torch.randn(D) * 0.01).to(dt).to(dev); b_dec = torch.zeros(A, dtype=dt, device=dev)
    acts = hp.encode(d, xs, W_encT, b_enc)
    sel = hp.select_topk(d, acts, k * B)
    hp.build_csc(d, sel, None)
    l1_term = 3e-6
    x_hat, g, rs, _, d_vals = hp.decode(d, sel, W_dec, b_dec, xt, None, None, want_g=True, want_dvals=True, l1_term=l1_term)
    h = adam_hyper(3e-4, 0.9, 0.99, 1e-8, 2)

    def fresh():
        gen = torch.Generator("cpu").manual_seed(3)
        mk = lambda *s: (torch.randn(*s, generator=gen) * 1e-3).to(dt).to(dev)
        mv = lambda *s: (torch.rand(*s, generator=gen) * 1e-6).to(dt).to(dev)
        return dict(Wd=W_dec.clone(), We=W_encT.clone(), be=b_enc.clone(), md=mk(D, A), vd=mv(D, A), me=mk(D, A), ve=mv(D, A),
                    mb=mk(D), vb=mv(D))

    ref = fresh()
    dW_dec, dW_encT, db_enc, _ = hp.backward(d, sel, g, d_vals, xs, None)
    hp.unitnorm_adam(ref["Wd"], dW_dec, ref["md"], ref["vd"], h, write_grad=False)
    hp.adam_flat(ref["We"], dW_encT, ref["me"], ref["ve"], h)
    hp.adam_flat(ref["be"], db_enc, ref["mb"], ref["vb"], h)
    for deferred in (False, True):
        got = fresh()
        hp.fused_backward_adam(d, sel, g, None if deferred else d_vals, xs, None, got["Wd"], (got["md"], got["vd"]),
                               got["We"], (got["me"], got["ve"]), got["be"], (got["mb"], got["vb"]), h, h, h, l1_term=l1_term)
        torch.cuda.synchronize()
        for name in ref:
            assert torch.equal(got[name], ref[name]), (name, deferred, float((got[name].float() - ref[name].float()).abs().max()))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("unit", [False, True])
def test_ragged_shapes(dt, unit, gpu_lib):
    """Nothing is a multiple of a tile: B = 37 rows (odd: partial wave-per-row workgroups), A = 264 (multiple of 8
    only -> generic encoder, a half-filled last register chunk in the row gathers), D = 1208 with a tiny first
    prefix group, k = 5."""
    cfg = dict(seed=49, prefix_sizes=[8, 200, 1208], act_size=264, top_k=5, top_k_aux=16,
               aux_penalty=1 / 32, n_batches_to_dead=20, l1_coeff=1e-3, dtype=dt, device="cpu",
               input_unit_norm=unit, lr=3e-4, beta1=0.9, beta2=0.99)
    B = 37
    params = R.init_params(cfg)
    params["b_enc"] = (0.01 * torch.randn(1208, generator=torch.Generator().manual_seed(3))).to(dt)
    g = torch.Generator("cpu").manual_seed(77)
    xs = torch.randn(B, 264, generator=g).to(dt)
    xt = torch.randn(B, 264, generator=g).to(dt)
    hp = _hot(cfg)
    cnt = torch.zeros(1208)
    res = run_forward(hp, cfg, params, cnt, xs, xt)
    check_against_oracle(cfg, params, cnt, xs, xt, res, hp)
