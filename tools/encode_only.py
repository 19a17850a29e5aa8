#!/usr/bin/env python3
"""Launches only the encoder (mtc_encode_fwd) at a BASELINE workload's shapes: a short program for rocprofv3 --pmc
passes and A/B timing of encoder variants.

    python tools/encode_only.py [--workload gemma2-2b] [--reps 40] [--zeros] [--tile 0|1|2] [--blas]

Prints one JSON line: HIP-event time per launch, TFLOP/s, fraction of the 2.5 PFLOP/s dense bf16 peak.  --blas also
times the same contraction through hipBLASLt (torch.matmul) as a yardstick for what the chip sustains on this shape
(a measurement aid only: the product path never calls it)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def timed(fn, reps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gemma2-2b", choices=list(bench.WORKLOADS))
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (clock held on trivial data)")
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--blas", action="store_true")
    ap.add_argument("--A", type=int, default=0, help="override the contraction length (slope / intercept of time over K)")
    args = ap.parse_args()
    from matryoshka_transcoder_amd.engine import HotPath
    from matryoshka_transcoder_amd import _lib
    torch.cuda.set_device(0)
    A, groups, k, B = bench.WORKLOADS[args.workload]
    A = args.A or A
    D = sum(groups)
    dt = torch.bfloat16
    g = torch.Generator("cpu").manual_seed(0)
    if args.zeros:
        x = torch.zeros(B, A, dtype=dt, device="cuda")
        W = torch.zeros(D, A, dtype=dt, device="cuda")
        b = torch.zeros(D, dtype=dt, device="cuda")
    else:
        x = torch.randn(B, A, generator=g).to(dt).cuda()
        W = (torch.randn(D, A, generator=g) * (6.0 / D) ** 0.5).to(dt).cuda()
        b = (torch.randn(D, generator=g) * 0.01).to(dt).cuda()
    hp = HotPath(A, A, bench.cumulative(groups), dt, False)
    d = hp.desc(B)
    if args.tile:
        _lib.load().mtc_encode_set_tile(args.tile)
    acts = torch.empty(B, D, dtype=dt, device="cuda")
    import ctypes as C
    from matryoshka_transcoder_amd.engine import _p, _stream
    lib = _lib.load()

    def run():
        lib.mtc_encode_fwd(C.byref(d), _p(x), _p(W), _p(b), _p(acts), _stream())

    ms = timed(run, args.reps)
    flop = 2.0 * B * A * D
    out = {"workload": args.workload, "B": B, "A": A, "D": D, "zeros": args.zeros, "tile": args.tile,
           "encode_ms": ms, "TFLOPs": flop / ms / 1e9, "frac_of_2500": flop / ms / 1e9 / 2500.0}
    if args.blas:
        Wt = W.t()                      # [A, D] view of the feature-major storage, as W_enc is held
        out["hipblaslt_xWt_ms"] = timed(lambda: torch.matmul(x, Wt), args.reps)
        Wc = Wt.contiguous()
        out["hipblaslt_xW_rowmajor_ms"] = timed(lambda: torch.matmul(x, Wc), args.reps)
        out["hipblaslt_best_TFLOPs"] = flop / min(out["hipblaslt_xWt_ms"], out["hipblaslt_xW_rowmajor_ms"]) / 1e9
    print(json.dumps(out))


if __name__ == "__main__":
    main()
