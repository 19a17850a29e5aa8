"""a9 on libmtc's own MFMA tiles (mtc_gemm_nt and the helpers around it) -- the auxiliary branch without a BLAS
library -- and the fused step's dead-list hand-over from one forward to the next (launches sized from the previous
list's bound, the count itself never read back inside the step).

Reference: sae.py:318-347 (dense form: acts_aux @ W_dec[dead]) and its autograd.  The GEMM entry is a floating-point
kernel, so it is compared with a plain fp32 torch contraction of the same operands: fp32 outputs rel 1e-5 (summation
order only); bf16 outputs must be the fp32 sum rounded ONCE (half an ulp + the same summation-order noise).  Index / layout helpers are bit-exact."""
import ctypes as C

import pytest
import torch

from conftest import tol
from oracle import ref_cpu as R
from test_gpu_kernels import _close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hp(dt, A=64, D=256):
    from matryoshka_transcoder_amd.engine import HotPath
    return HotPath(A, A, [D], dt, False)


def _ulp_bf16(ref):
    _, e = torch.frexp(ref.float())
    return torch.ldexp(torch.ones_like(ref, dtype=torch.float32), (e - 8).clamp_min(-40))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K,lda_pad,gather", [
    (128, 128, 64, 0, False),          # one MFMA tile, one k-step
    (300, 200, 512, 0, False),         # ragged M and N tiles
    (100, 2304, 640, 64, True),        # gathered A rows with a row stride longer than K (the aux backward's W_dec[dead])
    (4096, 64, 2304, 0, False),        # tall: d_acts^T-like
    (77, 36, 50, 6, True),             # K not a multiple of the k-tile: FMA kernel
])
def test_gemm_nt_against_fp32_contraction(dt, M, N, K, lda_pad, gather, gpu_lib):
    g = torch.Generator("cpu").manual_seed(M + N + K)
    rows_total = M + 13 if gather else M
    lda = K + lda_pad
    A = torch.randn(rows_total, lda, generator=g).to(dt)
    Bt = torch.randn(N, K, generator=g).to(dt)
    a_rows = torch.randperm(rows_total, generator=g)[:M].to(torch.int32) if gather else None
    hp = _hp(dt)
    Ad, Bd = A.to(DEV), Bt.to(DEV)
    ar = a_rows.to(DEV) if gather else None
    src = A[a_rows.long()] if gather else A
    ref = src[:, :K].double() @ Bt.double().t()
    for out_f32 in ([False, True] if dt == torch.bfloat16 else [False]):
        out = torch.full((M, N), float("nan"), dtype=torch.float32 if out_f32 else dt, device=DEV)
        hp.gemm_nt(M, N, K, Ad, lda, Bd, K, out, N, a_rows=ar, out_f32=out_f32)
        torch.cuda.synchronize()
        got = out.cpu()
        assert not bool(torch.isnan(got.float()).any())
        if dt == torch.float32 or out_f32:
            err = (got.double() - ref).abs().max().item()
            assert err <= 1e-5 * max(ref.abs().max().item(), 1.0) * (K ** 0.5), err
        else:
            # ONE rounding of the fp32 sum: half a bf16 ulp of the result + the fp32 summation-order noise (absolute,
            # ~1e-5 sqrt(K) for N(0,1) operands: it dominates where the sum cancels to a value near zero)
            want = ref.float().to(dt)
            err = (got.double() - ref).abs()
            assert bool((err <= 0.5 * _ulp_bf16(want).double() + 1e-5 * K ** 0.5).all()), float(err.max())
            assert float((got == want).float().mean()) >= 0.99      # roundings flip only where the fp32 sums straddle a tie


@pytest.mark.parametrize("M,N,K,lda_pad,ldc_pad,gather", [
    (256, 256, 64, 0, 0, False),          # one 256 x 256 tile, one k-step
    (700, 520, 448, 0, 0, False),         # ragged tiles on both sides, an odd number of k-steps
    (300, 1096, 640, 64, 24, True),       # gathered A rows, row strides longer than the rows on A and C
    (2816, 2304, 1024, 0, 0, False),      # 99 tiles, 16 k-steps
    (5632, 2304, 4096, 0, 0, True),       # E_dec / E_enc at config #2 with 5 632 dead columns: the automatic route
])
def test_gemm_nt_persistent_tile_equals_small_tile(M, N, K, lda_pad, ldc_pad, gather, gpu_lib):
    """Large bf16 products run the persistent 256 x 256 kernel (plain epilogue).  Same contraction order per output
    (k ascending in 32-wide MFMA steps, fp32 accumulate, one rounding), so bf16 AND fp32 outputs must equal the
    128 x 128 tile's bit for bit -- which test_gemm_nt_against_fp32_contraction pins to the fp32 contraction."""
    g = torch.Generator("cpu").manual_seed(M + N + K)
    rows_total = M + 29 if gather else M
    lda, ldc = K + lda_pad, N + ldc_pad
    A = torch.randn(rows_total, lda, generator=g).to(torch.bfloat16).to(DEV)
    Bt = torch.randn(N, K, generator=g).to(torch.bfloat16).to(DEV)
    ar = torch.randperm(rows_total, generator=g)[:M].to(torch.int32).to(DEV) if gather else None
    hp = _hp(torch.bfloat16)
    lib = hp.lib
    for out_f32 in (False, True):
        odt = torch.float32 if out_f32 else torch.bfloat16
        outs = []
        for tile in (1, 2):
            prev = lib.mtc_encode_set_tile(tile)
            try:
                out = torch.full((M, ldc), -3.0, dtype=odt, device=DEV)
                hp.gemm_nt(M, N, K, A, lda, Bt, K, out, ldc, a_rows=ar, out_f32=out_f32)
                torch.cuda.synchronize()
            finally:
                lib.mtc_encode_set_tile(prev)
            outs.append(out)
        assert torch.equal(outs[0], outs[1]), (out_f32, float((outs[0].float() - outs[1].float()).abs().max()))
        assert bool((outs[1][:, N:] == -3.0).all())                      # the padding of C's rows is not written
        if M * N >= 5632 * 2304:                                          # the size the automatic choice sends to the big tile
            out = torch.empty(M, ldc, dtype=odt, device=DEV)
            hp.gemm_nt(M, N, K, A, lda, Bt, K, out, ldc, a_rows=ar, out_f32=out_f32)
            torch.cuda.synchronize()
            assert torch.equal(out[:, :N], outs[1][:, :N])


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_transpose_rows_mask_rowsum_scatter_add(dt, gpu_lib):
    g = torch.Generator("cpu").manual_seed(3)
    hp = _hp(dt)
    src = torch.randn(150, 200, generator=g).to(dt)
    rows = torch.randperm(150, generator=g)[:97].to(torch.int32)
    # gather + transpose + zero padding of the k-tile
    dst = hp.transpose_rows(src.to(DEV), 97, 200, R_pad=128, src_rows=rows.to(DEV))
    torch.cuda.synchronize()
    want = torch.zeros(200, 128, dtype=dt)
    want[:, :97] = src[rows.long()].t()
    assert torch.equal(dst.cpu(), want)
    assert torch.equal(hp.transpose_rows(src.to(DEV), 150, 200).cpu(), src.t().contiguous())
    # mask + row sums on the transposed layout
    for n, B in ((37, 300), (37, 1544)):              # element-wise rows; 8-element pieces (B % 8 == 0)
        a = torch.randn(n, B, generator=g).clamp_min(0).to(dt)
        d = torch.randn(n, B, generator=g).to(dt)
        a_d, dd, e_b = a.to(DEV), d.clone().to(DEV), torch.empty(n, dtype=dt, device=DEV)     # operands stay referenced
        _lib_check(hp.lib.mtc_aux_mask_rowsum(hp.code, n, B, _ptr(a_d), B, _ptr(dd), B, _ptr(e_b), None))
        torch.cuda.synchronize()
        masked = torch.where(a > 0, d, torch.zeros((), dtype=dt))
        assert torch.equal(dd.cpu(), masked)
        rt, at = (1e-5, 1e-5) if dt == torch.float32 else (1e-2, 1e-2)
        _close(e_b.float(), masked.float().sum(1), rt, at, "row sums")
    # grad[dead] += rows with the live count on the device: rows beyond it are left alone
    dst0 = torch.randn(64, 48, generator=g).to(dt)
    add = torch.randn(16, 48, generator=g).to(dt)
    idx = torch.randperm(64, generator=g)[:16].to(torch.int32)
    idx_d, add_d = idx.to(DEV), add.to(DEV)              # referenced until the launch has run (a temporary's block would be reused)
    for live in (16, 9, 0):
        dstd = dst0.clone().to(DEV)
        n_dev = torch.tensor([live], dtype=torch.int32, device=DEV)
        _lib_check(hp.lib.mtc_rows_scatter_add(hp.code, 16, 48, _ptr(idx_d), _ptr(add_d), 48, _ptr(dstd), 48, 64,
                                               _ptr(n_dev), None))
        torch.cuda.synchronize()
        want = dst0.clone()
        want[idx[:live].long()] = (want[idx[:live].long()].float() + add[:live].float()).to(dt)
        assert torch.equal(dstd.cpu(), want), live


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _lib_check(rc):
    from matryoshka_transcoder_amd._lib import check
    check(rc, "libmtc")


def test_dead_list_tail_bound_and_map(gpu_lib):
    hp = _hp(torch.float32, D=1000)
    g = torch.Generator("cpu").manual_seed(11)
    cnt = torch.randint(0, 25, (1000,), generator=g).float()
    dl = hp.aux_dead_list(hp.desc(8), cnt.to(DEV), 20.0)
    n = dl.n()
    dead = torch.nonzero(cnt >= 20).flatten()
    assert n == dead.numel() and dl.next_bound() == int((cnt >= 19).sum())
    assert torch.equal(dl.dead_idx[:n].cpu().long(), dead) and not bool(dl.dead_idx[n:].any())
    rm = torch.full((1000,), -1, dtype=torch.int32)
    rm[dead] = torch.arange(n, dtype=torch.int32)
    assert torch.equal(dl.row_map.cpu(), rm)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,D,n_dead,n_pad,k_aux,coarse", [
    (37, 3000, 700, 768, 64, False),        # the common case: threshold inside the positives
    (37, 3000, 700, 768, 64, True),         # values on a coarse grid: many ties at every threshold, quota < ties
    (9, 3000, 700, 1024, 600, False),       # fewer positives than k_aux: every positive one is taken
    (5, 512, 1, 128, 8, False),             # one dead feature
    (5, 512, 0, 128, 8, False),             # none: zero rows
    (6, 30000, 24000, 24064, 512, True),    # beyond the LDS budget: the entry falls back to the two launches
    (6, 1000, 333, 334, 50, True),          # odd width: no 8-byte rows -> fallback (bf16), fused (fp32)
])
def test_select_dense_one_pass_equals_select_then_fill(dt, B, D, n_dead, n_pad, k_aux, coarse, gpu_lib):
    """mtc_aux_select_dense against mtc_aux_select + mtc_aux_dense_fill (bit for bit: counts, threshold bits, tie quotas,
    acts_aux), and against the definition (sae.py:332-336: top-k_aux of the dead columns, ties to the lower feature)."""
    g = torch.Generator("cpu").manual_seed(B * 7 + n_dead)
    hp = _hp(dt, D=D)
    d = hp.desc(B)
    acts = torch.randn(B, D, generator=g)
    if coarse:
        acts = (acts * 4).round() / 4
    acts = acts.clamp_min(0).to(dt)
    dead = torch.sort(torch.randperm(D, generator=g)[:n_dead]).values
    dead_idx = torch.zeros(D, dtype=torch.int32)
    dead_idx[:n_dead] = dead.to(torch.int32)
    acts_d, idx_d = acts.to(DEV), dead_idx.to(DEV)
    n_d = torch.tensor([n_dead], dtype=torch.int32, device=DEV)
    i32 = dict(dtype=torch.int32, device=DEV)
    c1, t1, q1 = (torch.full((B,), -7, **i32) for _ in range(3))
    c2, t2, q2 = (torch.full((B,), -7, **i32) for _ in range(3))
    a1 = torch.full((B, n_pad), float("nan"), dtype=dt, device=DEV)
    a2 = torch.full((B, n_pad), float("nan"), dtype=dt, device=DEV)
    lib = hp.lib
    _lib_check(lib.mtc_aux_select(C.byref(d), _ptr(acts_d), _ptr(idx_d), _ptr(n_d), k_aux, _ptr(c1), _ptr(t1), _ptr(q1), None))
    _lib_check(lib.mtc_aux_dense_fill(C.byref(d), _ptr(acts_d), _ptr(idx_d), _ptr(n_d), _ptr(t1), _ptr(q1), n_pad, _ptr(a1), None))
    _lib_check(lib.mtc_aux_select_dense(C.byref(d), _ptr(acts_d), _ptr(idx_d), _ptr(n_d), k_aux, n_pad, _ptr(c2), _ptr(t2),
                                        _ptr(q2), _ptr(a2), None))
    torch.cuda.synchronize()
    assert torch.equal(c1, c2) and torch.equal(t1, t2) and torch.equal(q1, q2)
    assert torch.equal(a1.view(torch.int16 if dt == torch.bfloat16 else torch.int32),
                       a2.view(torch.int16 if dt == torch.bfloat16 else torch.int32))
    # the definition: stable sort of the dead columns by value (descending), the first k positives stay
    sub = acts[:, dead.long()].float()
    want = torch.zeros(B, n_pad)
    if n_dead:
        order = torch.sort(sub, dim=1, descending=True, stable=True).indices[:, :min(k_aux, n_dead)]
        keep = torch.zeros_like(sub, dtype=torch.bool).scatter_(1, order, True) & (sub > 0)
        want[:, :n_dead] = torch.where(keep, sub, torch.zeros(()))
    assert torch.equal(a2.float().cpu(), want)
    assert torch.equal(c2.cpu().long(), (want > 0).sum(1))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fused_steps_with_lagged_dead_bound(dt, gpu_lib):
    """Six fused steps while the dead set changes every step, against the oracle's train step.  Step 0 reads the dead
    count back once (first active step); afterwards each forward is sized from the bound the previous list left behind:
    the bound exceeds the count (near-dead features that fire again), the count falls to ZERO under a positive bound
    (feature X fires on step 3: the device switches the branch off, loss / aux_loss come back in the model dtype as
    sae.py:331,342-347 has it), and the branch comes back later."""
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    A, D, B, k = 64, 512, 128, 8
    f32 = dt == torch.float32
    cfg = dict(seed=49, prefix_sizes=[64, 192, 512], act_size=A, top_k=k, top_k_aux=16, aux_penalty=1 / 32,
               n_batches_to_dead=20, l1_coeff=0.0, dtype=dt, device=DEV, input_unit_norm=False, lr=3e-4, beta1=0.9,
               beta2=0.99)
    m = MatryoshkaTranscoder(cfg)
    ocfg = dict(cfg, device="cpu")
    params, state = R.init_params(ocfg), R.init_state(ocfg)
    # X = feature 7 listens to input column 0 only: the inputs decide on which steps it fires
    X, tiny = 7, torch.arange(300, 340)
    with torch.no_grad():
        params["W_enc"][:, X] = 0
        params["W_enc"][0, X] = 6.0
        params["W_enc"][:, tiny] *= 0.01                 # never win the global TopK: dead from step 4 on (counters 15 + 5)
        m.load_state_dict({k_: v.to(DEV) for k_, v in params.items()})
    cnt0 = torch.zeros(D)
    cnt0[X] = 19
    cnt0[tiny] = 15
    cnt0[400:420] = 18                                   # ordinary features near the limit: most fire again in time
    state["num_batches_not_active"] = cnt0.clone()
    m.num_batches_not_active = cnt0.clone().to(DEV)
    opt = FusedAdam(m, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    oopt = R.new_opt(params)
    hp = m.hot_path()
    fires = [False, False, False, True, True, False]     # X: dead on steps 0-2 (aux on), fires on 3 (aux off under a bound), ...
    seen = []
    for i, fire in enumerate(fires):
        g = torch.Generator("cpu").manual_seed(500 + i)
        xs = torch.randn(B, A, generator=g)
        xs[:, 0] = 3.0 if fire else -3.0
        xs, xt = xs.to(dt), torch.randn(B, A, generator=g).to(dt)
        xs_d, xt_d = xs.to(DEV), xt.to(DEV)
        with torch.no_grad():
            acts_gpu = hp.encode(hp.desc(B), xs_d, m.W_enc.t(), m.b_enc).cpu()
        bound_before = m._aux_bound
        out = opt.train_step(xs_d, xt_d)
        o, inter, _ = R.train_step(params, state, oopt, xs, xt, ocfg, acts_override=acts_gpu)
        n_dead = inter["n_dead"]
        seen.append((bound_before, n_dead))
        res = m._last_res
        if res.aux is not None:
            assert res.aux["dead"].n() == n_dead
            assert bound_before is None or bound_before >= n_dead
        else:
            assert n_dead == 0
        rt, at = tol(dt)
        _close(out["loss"].float(), o["loss"].float(), rt, at, f"step {i} loss")
        _close(out["aux_loss"].float(), o["aux_loss"].float(), rt, at * 0.01, f"step {i} aux_loss")
        assert out["loss"].dtype == o["loss"].dtype and out["aux_loss"].dtype == o["aux_loss"].dtype, (i, n_dead)
        # (the public num_batches_not_active accessor hands the tensor out for editing and so voids the carried bound)
        assert torch.equal(m._cnt.cpu(), state["num_batches_not_active"])
        for name in ("W_dec", "W_enc", "b_enc", "b_dec"):
            got, ref = getattr(m, name).detach().float().cpu(), params[name].float()
            if f32:
                _close(got, ref, 0, 1e-2 * cfg["lr"] + 1e-7, f"{name} after step {i}")
            else:
                assert float((got - ref).abs().max()) <= 2.0 * (i + 1) * cfg["lr"] + 2.0 ** -8 * float(ref.abs().max()), name
    # the scenario really went through the cases it is here for
    assert seen[0][0] is None and seen[0][1] > 0                       # first active step: own count
    assert any(b is not None and b > n > 0 for b, n in seen)           # bound above the count
    assert any(b is not None and b > 0 and n == 0 for b, n in seen)    # branch off under a positive bound
    assert seen[-1][1] > len(tiny) - 1                                 # and on again, with the tiny features dead
