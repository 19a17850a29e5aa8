"""The one entry point that allocates nothing (prompt item 7): mtc_colsum_static, called through the C ABI, against
mtc_colsum (caller workspace) -- bit-exact, on two calls in a row, the second with a smaller input; sizes beyond the
static scratch are refused with a clear error.  Collected last (file name) so a failure here hides no other test."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(lib, B, A, dt, seed, scale=None, accumulate=0):
    from matryoshka_transcoder_amd import _lib
    from matryoshka_transcoder_amd.engine import _p, _stream, dtype_code
    g = (torch.randn(B, A, generator=torch.Generator("cpu").manual_seed(seed)) * 1e-3).to(dt).to(DEV)
    d = _lib.make_desc(B, A, A, 1024, [256, 1024], dtype_code(dt), False)
    sc = None if scale is None else torch.tensor([scale], dtype=torch.float32, device=DEV)
    init = (torch.randn(A, generator=torch.Generator("cpu").manual_seed(seed + 1)) * 1e-3).to(dt).to(DEV)
    want, got = init.clone(), init.clone()
    ws_bytes = lib.mtc_colsum_workspace_bytes(C.byref(d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    _lib.check(lib.mtc_colsum(C.byref(d), _p(g), _p(sc), _p(ws), ws_bytes, _p(want), accumulate, _stream()), "mtc_colsum")
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    _lib.check(lib.mtc_colsum_static(C.byref(d), _p(g), _p(sc), _p(got), accumulate, _stream()), "mtc_colsum_static")
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    return got, want, g


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_colsum_static_matches_colsum_two_calls_in_a_row(dt, gpu_lib):
    lib = gpu_lib
    # call 1: a large input (all 64 slabs busy, the scratch is filled up to column 4608)
    got, want, g = _pair(lib, 4096, 4608, dt, seed=1)
    assert torch.equal(got, want)
    ref = g.float().sum(0)
    assert float((got.float() - ref).abs().max()) <= (1e-6 if dt == torch.float32 else 2e-2 * float(ref.abs().max()) + 1e-6)
    # call 2, right after, smaller in both dimensions (B = 37 rows: most slabs are empty and must contribute zeros,
    # not what call 1 left in the scratch), with a scale and accumulation into an existing gradient
    got, want, g = _pair(lib, 37, 264, dt, seed=2, scale=0.5, accumulate=1)
    assert torch.equal(got, want)
    # call 3: back to a different larger shape
    got, want, _ = _pair(lib, 300, 2304, dt, seed=3)
    assert torch.equal(got, want)


def test_colsum_static_rejects_sizes_beyond_its_scratch(gpu_lib):
    from matryoshka_transcoder_amd import _lib
    from matryoshka_transcoder_amd.engine import _p, _stream
    A = 8192 + 8
    g = torch.zeros(4, A, dtype=torch.float32, device=DEV)
    out = torch.zeros(A, dtype=torch.float32, device=DEV)
    d = _lib.make_desc(4, A, A, 1024, [1024], _lib.MTC_F32, False)
    rc = gpu_lib.mtc_colsum_static(C.byref(d), _p(g), None, _p(out), 0, _stream())
    assert rc != 0
    msg = gpu_lib.mtc_last_error().decode()
    assert "8192" in msg and "mtc_colsum_static" in msg
    # the largest supported width works
    got, want, _ = _pair(gpu_lib, 130, 8192, torch.float32, seed=4)
    assert torch.equal(got, want)
