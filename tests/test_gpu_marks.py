"""The encoder's candidate marks (mtc_encode_fwd_marked -> mtc_select_global_topk_carry_marked): the select that reads
only the flagged granules must return the selection of the select that streams all activations, bit for bit, and the
marks must flag exactly the granules that hold an entry >= the carried bound (sae.py:173-182)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(B, A, D, seed=0):
    g = torch.Generator("cpu").manual_seed(seed)
    dt = torch.bfloat16
    x = torch.randn(B, A, generator=g).to(dt).to(DEV)
    W = (torch.randn(D, A, generator=g) * (1.0 / A) ** 0.5).to(dt).to(DEV)
    b = (torch.randn(D, generator=g) * 0.01).to(dt).to(DEV)
    return x, W, b


def _sel_tuple(sel):
    rp = sel.row_ptr.cpu()
    nnz = int(rp[-1])
    return rp, sel.col_idx[:nnz].cpu(), sel.vals[:nnz].cpu().view(torch.int16), sel.state.cpu()[:8]


@pytest.mark.parametrize("B,A,D,k", [(512, 256, 4096, 24), (300, 192, 1672, 16), (1024, 128, 2056, 8)])
def test_marked_select_equals_streamed_select(B, A, D, k, gpu_lib):
    """Ragged shapes too (D not a multiple of 128 / 16, B not a multiple of 256): the 256 x 256 persistent kernel is
    forced; two HotPaths see the same activations, one with marks, one without."""
    from matryoshka_transcoder_amd.engine import HotPath
    lib = gpu_lib
    prev = lib.mtc_encode_set_tile(2)
    try:
        hp_m = HotPath(A, A, [D], torch.bfloat16, False)
        hp_s = HotPath(A, A, [D], torch.bfloat16, False)
        hp_s.use_marks = False
        d = hp_m.desc(B)
        assert lib.mtc_encode_marks_bytes(C.byref(d)) > 0
        used_marks = 0
        for step in range(5):
            x, W, b = _setup(B, A, D, seed=step // 2)           # pairs of steps share the data: the second has a valid bound
            if step == 3:
                W = (W.float() * 0.97).to(torch.bfloat16)      # tau moves a little: bound still valid
            acts_m = hp_m.encode(d, x, W, b, mark_for=hp_m.carry_state(x.device))
            acts_s = hp_s.encode(d, x, W, b, mark_for=hp_s.carry_state(x.device))
            assert torch.equal(acts_m, acts_s)
            marks = getattr(acts_m, "_mtc_marks", None)
            assert marks is not None and getattr(acts_s, "_mtc_marks", None) is None
            guess = int(hp_m.carry_state(x.device).cpu()[0])
            nblk = (D + 127) // 128
            body = (B * nblk + 3) // 4 * 4
            assert int(marks[body:body + 4].view(torch.int32).cpu()) == guess
            # marks = exactly the granules with an entry >= the bound (bit pattern compare on non-negative bf16 values)
            bits = acts_m.view(torch.int16).to(torch.int32) & 0xffff
            pad = nblk * 128 - D
            if pad:
                bits = torch.cat([bits, torch.zeros(B, pad, dtype=torch.int32, device=DEV)], 1)
            g16 = (guess >> 16) if guess else 0x10000
            want = (bits.reshape(B, nblk, 8, 16) >= g16).any(-1)                       # [B, nblk, 8]
            got = ((marks[:B * nblk].reshape(B, nblk, 1).to(torch.int32) >> torch.arange(8, device=DEV)) & 1).bool()
            assert torch.equal(got, want), f"step {step}: marks differ from the activations"
            before = hp_m.carry_state(x.device).cpu().clone()
            sm = hp_m.select_topk(d, acts_m, k * B, carry=True)
            ss = hp_s.select_topk(d, acts_s, k * B, carry=True)
            for a, b_ in zip(_sel_tuple(sm), _sel_tuple(ss)):
                assert torch.equal(a, b_), f"step {step}: marked select differs from the streamed select"
            after = hp_m.carry_state(x.device).cpu()
            used_marks += int(after[4] - before[4])              # calls served by the compact (carried-bound) path
        assert used_marks >= 2
    finally:
        lib.mtc_encode_set_tile(prev)


def test_marks_for_another_bound_are_ignored(gpu_lib):
    """Marks carry the bound they were made for: if the select runs with another bound (someone ran a select in
    between), it streams the activations instead -- same result as without marks."""
    from matryoshka_transcoder_amd.engine import HotPath
    lib = gpu_lib
    B, A, D, k = 512, 256, 4096, 24
    prev = lib.mtc_encode_set_tile(2)
    try:
        hp = HotPath(A, A, [D], torch.bfloat16, False)
        d = hp.desc(B)
        x, W, b = _setup(B, A, D)
        carry = hp.carry_state(x.device)
        hp.select_topk(d, hp.encode(d, x, W, b, mark_for=carry), k * B, carry=True)           # establishes a bound
        acts = hp.encode(d, x, W, b, mark_for=carry)
        ref = hp.select_topk(d, acts.clone(), k * B, carry=False)                             # full select, no carry
        c = carry.cpu()
        c[0] = int(c[0]) - (3 << 16)                                                          # a lower (still valid) bound
        carry.copy_(c.to(DEV))
        got = hp.select_topk(d, acts, k * B, carry=True)                                      # marks are stale now
        for a, b_ in zip(_sel_tuple(got)[:3], _sel_tuple(ref)[:3]):
            assert torch.equal(a, b_)
    finally:
        lib.mtc_encode_set_tile(prev)
