#!/usr/bin/env python3
"""Kernel-level timing at BASELINE config #2 shapes (or --workload gpt2): each libmtc stage timed with HIP
events over several repetitions, interleaved, in one process.  Used for A/B work on single kernels."""
import argparse
import os
import re
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from matryoshka_transcoder_amd.engine import HotPath, adam_hyper  # noqa: E402


def timeit(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    evs = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in evs)
    return t[len(t) // 2], t[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gemma2-2b")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dev = "cuda:0"
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
    x_hat, g, row_stats, _, d_vals = hp.decode(d, sel, W_dec, b_dec, xt, None, None, want_g=True, want_dvals=True)
    hp1 = HotPath(A, A, [D], dt, False)
    d1 = hp1.desc(B)
    m, v = torch.zeros_like(W_dec), torch.zeros_like(W_dec)
    grads = hp.backward(d, sel, g, d_vals, xs, None)
    h = adam_hyper(3e-4, 0.9, 0.99, 1e-8, 1)
    m2, v2 = torch.zeros_like(W_encT), torch.zeros_like(W_encT)
    mb, vb = torch.zeros_like(b_enc), torch.zeros_like(b_enc)
    big = torch.empty(512 * 1024 * 1024, dtype=torch.uint8, device=dev)

    def flush():
        big.zero_()          # 512 MiB write: evicts L2 / Infinity Cache between repetitions

    # side path: top-sample tables fed with the batch's selection (B positions = B/128 sequences of 128 tokens)
    from matryoshka_transcoder_amd.activation_samples import ActivationSampleCollector
    col = ActivationSampleCollector(100, 20, 0.1)
    toks = torch.randint(0, 250000, (B // 128, 128), device=dev)
    dense = hp.densify(d, sel, D)

    tests = {
        "collect_selection": lambda: col.collect_selection(sel, dt, B, D, toks, None),
        "collect_dense": lambda: col.collect_batch_samples(dense, toks, None),
        "encode": lambda: hp.encode(d, xs, W_encT, b_enc),
        "select_topk": lambda: hp.select_topk(d, acts, k * B),
        "select_topk_carry": lambda: hp.select_topk(d, acts, k * B, carry=True),
        "build_csc": lambda: hp.build_csc(d, sel, None),
        "decode_G4_fwd_only": lambda: hp.decode(d, sel, W_dec, b_dec, xt, None, None, want_g=True, want_dvals=False),
        "decode_G1_fwd_only": lambda: hp1.decode(d1, sel, W_dec, b_dec, xt, None, None, want_g=True, want_dvals=False),
        "decode_G4_fused": lambda: hp.decode(d, sel, W_dec, b_dec, xt, None, None, want_g=True, want_dvals=True),
        "decode_bwd_vals": lambda: hp.decode_bwd_vals(d, sel, W_dec, g, 0.0),
        "backward": lambda: hp.backward(d, sel, g, d_vals, xs, None),
        "fused_backward_adam": lambda: hp.fused_backward_adam(
            d, sel, g, d_vals, xs, None, W_dec, (m, v), W_encT, (m2, v2), b_enc, (mb, vb), h, h, h),
        "fused_backward_adam_dvals": lambda: hp.fused_backward_adam(
            d, sel, g, None, xs, None, W_dec, (m, v), W_encT, (m2, v2), b_enc, (mb, vb), h, h, h),
        "unitnorm_adam": lambda: hp.unitnorm_adam(W_dec, grads[0], m, v, h, write_grad=False),
        "adam_flat": lambda: hp.adam_flat(W_dec, grads[0], m, v, h),
    }
    for name, fn in tests.items():
        if args.only and not re.search(args.only, name):
            continue
        med, mn = timeit(fn)
        # cold-cache variant: flush before each repetition (timed region excludes the flush)
        ts = []
        for _ in range(5):
            flush()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"{name:24s} warm median {med:7.3f} ms  min {mn:7.3f} ms   cold median {sorted(ts)[2]:7.3f} ms", flush=True)


if __name__ == "__main__":
    main()
