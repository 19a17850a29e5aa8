#!/usr/bin/env python3
"""A/B of the fused gradient + Adam launches at a BASELINE workload: row kernels vs the column-sliced forms, HIP events
around each launch pair, interleaved repetitions in one process (config #2 by default)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from matryoshka_transcoder_amd.engine import adam_hyper  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gemma2-2b")
    ap.add_argument("--reps", type=int, default=12)
    args = ap.parse_args()
    dt, dev = torch.bfloat16, "cuda:0"
    cfg, B = bench.make_cfg(args.workload, dt, dev)
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    tc = MatryoshkaTranscoder(cfg)
    hp = tc.hot_path()
    A, D, k = cfg["act_size"], hp.D, cfg["top_k"]
    xs, xt = bench.batch_pool(B, A, dt, dev, n=1)[0]
    d = hp.desc(B)
    W_encT, W_dec = tc.W_enc.detach().t(), tc.W_dec.detach()
    b_enc, b_dec = tc.b_enc.detach(), tc.b_dec.detach()
    acts = hp.encode(d, xs, W_encT, b_enc)
    sel = hp.select_topk(d, acts, k * B)
    hp.build_csc(d, sel, None)
    x_hat, g, row_stats, _, _ = hp.decode(d, sel, W_dec, b_dec, xt, None, None, want_g=True, want_dvals=False)
    m, v = torch.zeros_like(W_dec), torch.zeros_like(W_dec)
    m2, v2 = torch.zeros_like(W_encT), torch.zeros_like(W_encT)
    mb, vb = torch.zeros_like(b_enc), torch.zeros_like(b_enc)
    h = adam_hyper(3e-4, 0.9, 0.99, 1e-8, 1)
    import ctypes as C
    lib = hp.lib
    real_dec, real_enc = lib.mtc_fused_backward_adam_dec_extra, lib.mtc_fused_backward_adam_enc_extra
    evs = {}

    def timed(name, fn):
        def inner(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r = fn(*a); e1.record()
            evs.setdefault(name, []).append((e0, e1))
            return r
        return inner

    for mode, tag in ((1, "row"), (2, "sliced")) * args.reps:
        lib.mtc_fused_set_sliced(mode)
        lib.mtc_fused_backward_adam_dec_extra = timed(f"dec[{tag}]", real_dec)
        lib.mtc_fused_backward_adam_enc_extra = timed(f"enc[{tag}]", real_enc)
        hp.fused_backward_adam(d, sel, g, None, xs, None, W_dec, (m, v), W_encT, (m2, v2), b_enc, (mb, vb), h, h, h)
    # lower bound of a column-sliced DECODER-row launch (two passes over the slices, see mtc_diag_sliced_decoder_bound)
    import ctypes as C
    from matryoshka_transcoder_amd.engine import _p, _stream
    Wc, mc, vc = W_dec.clone(), m.clone(), v.clone()
    coef_csc = sel.vals[sel.csc_pos.long()].float().contiguous()
    scratch = torch.empty(D * (A // 256) * 2, dtype=torch.float32, device=dev)
    for ps, tag in ((0, "dec sliced bound: pass A (gather + partial sums)"), (1, "dec sliced bound: pass B (gather + update)")) * args.reps:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.mtc_diag_sliced_decoder_bound(C.byref(d), _p(sel.col_ptr), _p(sel.csc_row), _p(coef_csc), _p(g), _p(Wc), _p(mc),
                                               _p(vc), _p(scratch), ps, _stream())
        e1.record()
        assert rc == 0
        evs.setdefault(tag, []).append((e0, e1))
    torch.cuda.synchronize()
    for name, lst in sorted(evs.items()):
        t = sorted(a.elapsed_time(b) for a, b in lst[2:])
        print(f"{name:52s} median {t[len(t) // 2]:.4f} ms  min {t[0]:.4f} ms  ({len(t)} launches)", flush=True)


if __name__ == "__main__":
    main()
