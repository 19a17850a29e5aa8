"""Column-sliced encoder-row gradient + Adam kernel (fused_enc_sliced_kernel, optim.hip) against the row kernel it
replaces at large sizes: same inputs, same CSC walk, so W_enc / exp_avg / exp_avg_sq must come out BIT-IDENTICAL (every
element sees the same FMA order, one rounding of the gradient, the same op-by-op Adam); the bias element's gradient is
summed in a different fp32 order (sequential instead of a lane tree), so b_enc is compared to a tolerance.
Reference: autograd of sae.py:173-174 + training.py:178.  Also pins the CSC-ordered d a_bj hand-over between the two
launches of the fused step against the CSR-ordered one (d_vals made by the decode kernel)."""
import pytest
import torch

from test_gpu_kernels import _close
from test_gpu_widths import _steps_vs_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def sliced_knob(gpu_lib):
    prev = gpu_lib.mtc_fused_set_sliced(0)
    yield gpu_lib.mtc_fused_set_sliced
    gpu_lib.mtc_fused_set_sliced(prev)


def _problem(B, A, D, k, seed):
    from matryoshka_transcoder_amd.engine import HotPath
    g = torch.Generator("cpu").manual_seed(seed)
    dt = torch.bfloat16
    hp = HotPath(A, A, [D // 4, D], dt, False)
    d = hp.desc(B)
    xs = torch.randn(B, A, generator=g).to(dt).to(DEV)
    xt = torch.randn(B, A, generator=g).to(dt).to(DEV)
    W_encT = (torch.randn(D, A, generator=g) * A ** -0.5).to(dt).to(DEV)
    W_dec = torch.nn.functional.normalize(torch.randn(D, A, generator=g), dim=1).to(dt).to(DEV)
    b_enc = (0.1 * torch.randn(D, generator=g)).to(dt).to(DEV)
    # one feature fires on every row (a long CSC segment), a block of features never fires (empty segments)
    b_enc[5] = 6.0
    b_enc[D // 2:D // 2 + 40] = -50.0
    b_dec = torch.zeros(A, dtype=dt, device=DEV)
    acts = hp.encode(d, xs, W_encT, b_enc)
    sel = hp.select_topk(d, acts, k * B)
    hp.build_csc(d, sel, None)
    x_hat, gg, row_stats, _, d_vals = hp.decode(d, sel, W_dec, b_dec, xt, None, None, want_g=True, want_dvals=True)
    return hp, d, sel, gg, d_vals, xs, W_encT, W_dec, b_enc


def _run(hp, d, sel, gg, d_vals, xs, W_encT, W_dec, b_enc, step=3):
    from matryoshka_transcoder_amd.engine import adam_hyper
    g = torch.Generator("cpu").manual_seed(77)
    mk = lambda t, s: (s * torch.randn(t.shape, generator=g)).to(t.dtype).to(DEV)
    We, Wd, be = W_encT.clone(), W_dec.clone(), b_enc.clone()
    st = [(mk(We, 1e-3), mk(We, 1e-3).abs()), (mk(Wd, 1e-3), mk(Wd, 1e-3).abs()), (mk(be, 1e-3), mk(be, 1e-3).abs())]
    v_e0 = st[0][1].clone()
    h = adam_hyper(3e-4, 0.9, 0.99, 1e-8, step)
    hp.fused_backward_adam(d, sel, gg, d_vals, xs, None, Wd, st[1], We, st[0], be, st[2], h, h, h)
    torch.cuda.synchronize()
    return dict(v_e0=v_e0, W_encT=We, m_e=st[0][0], v_e=st[0][1], W_dec=Wd, m_d=st[1][0], v_d=st[1][1], b_enc=be, m_b=st[2][0], v_b=st[2][1])


@pytest.mark.parametrize("B,A,D,k,lpf", [
    (512, 768, 1000, 8, 32),         # 3 slices of 256 columns, ragged feature count (1000 = 125 blocks of 8)
    (5200, 384, 520, 4, 16),         # table slice too big for 256 columns at this B and A % 256 != 0: 128-column slices
    (300, 192, 264, 6, 8),           # 64-column slices, 8 features per wave
    (64, 2304, 72, 12, 32),          # the headline width: 9 slices
])
def test_sliced_encoder_rows_equal_row_kernel(B, A, D, k, lpf, sliced_knob):
    prob = _problem(B, A, D, k, seed=B + A)
    sliced_knob(1)
    row_csr = _run(*prob)                                        # d a_bj from the decode kernel (CSR order), row kernel
    prob_fused = prob[:4] + (None,) + prob[5:]
    row = _run(*prob_fused)                                      # d a_bj from the decoder launch (CSC order), row kernel
    sliced_knob(2)
    sl = _run(*prob_fused)                                       # same, column-sliced encoder launch
    for name in ("W_encT", "m_e", "v_e", "W_dec", "m_d", "v_d"):
        assert torch.equal(row[name], row_csr[name]), f"{name}: CSC-ordered d_vals differ from the CSR-ordered path"
        assert torch.equal(sl[name], row[name]), f"{name}: sliced kernel differs from the row kernel"
    for name in ("b_enc", "m_b", "v_b"):
        assert torch.equal(row[name], row_csr[name]), name
        _close(sl[name].float(), row[name].float(), 1e-2, 1e-6, name)
        assert float((sl[name] == row[name]).float().mean()) >= 0.99, name
    # the update really happened in every column (v <- 0.99 v + 0.01 g^2 moves nearly every bf16 entry)
    assert bool(((sl["v_e"] != sl["v_e0"]).float().mean(0) > 0.5).all())


def test_three_steps_vs_oracle_with_sliced_kernel(sliced_knob):
    """The whole fused step with the sliced encoder launch forced (3 slices), three steps against the oracle at the
    bf16 bar of SURVEY section 9 (>= 99.9 % of the weights within 1 ulp)."""
    sliced_knob(2)
    rep = _steps_vs_oracle(768, 2048, [128, 384, 896, 2048], B=256, k=16, dt=torch.bfloat16, unit=False, steps=3,
                           min_frac=0.999)
    print({f"{k[1]}@{k[0]}": round(v, 5) for k, v in rep.items()})
