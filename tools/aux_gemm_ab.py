#!/usr/bin/env python3
"""A/B of the tiles behind the auxiliary branch's four products at their real shapes (config #2, n_pad dead columns):
mtc_gemm_nt's 128 x 128 tile, the persistent 256 x 256 encoder kernel run on the same contraction (timing of the tile
walk only: its epilogue adds a bias and a ReLU), and hipBLASLt (torch.matmul) as a yardstick of what the chip sustains
on the shape.  A measurement aid: the product path never calls a BLAS library.

    python tools/aux_gemm_ab.py [--n-pad 5632] [--reps 40]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.encode_only import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-pad", type=int, default=5632)
    ap.add_argument("--B", type=int, default=4096)
    ap.add_argument("--A", type=int, default=2304)
    ap.add_argument("--reps", type=int, default=40)
    args = ap.parse_args()
    from matryoshka_transcoder_amd.engine import HotPath, _p, _stream
    from matryoshka_transcoder_amd import _lib
    torch.cuda.set_device(0)
    lib = _lib.load()
    dt = torch.bfloat16
    B, A, n = args.B, args.A, args.n_pad
    shapes = {"x_hat_aux   [B, A]     K = n_pad": (B, A, n),
              "d_acts^T    [n_pad, B] K = A": (n, B, A),
              "E_dec/E_enc [n_pad, A] K = B": (n, A, B)}
    out = {"B": B, "A": A, "n_pad": n, "products": {}}
    g = torch.Generator("cpu").manual_seed(0)
    for name, (M, N, K) in shapes.items():
        a = torch.randn(M, K, generator=g).to(dt).cuda()
        bt = (torch.randn(N, K, generator=g) * K ** -0.5).to(dt).cuda()
        bias = torch.zeros(N, dtype=dt, device="cuda")
        c = torch.empty(M, N, dtype=dt, device="cuda")
        hp = HotPath(K, K, [N], dt, False)
        d = hp.desc(M)
        row = {"M": M, "N": N, "K": K, "tiles_128": -(-M // 128) * -(-N // 128), "tiles_256": -(-M // 256) * -(-N // 256)}
        row["gemm_nt_128_ms"] = timed(lambda: lib.mtc_gemm_nt(hp.code, M, N, K, _p(a), K, None, _p(bt), K, _p(c), N, 0, _stream()),
                                      args.reps)
        ref = c.float().clone()
        prev = lib.mtc_encode_set_tile(2)
        row["persistent_256_ms"] = timed(lambda: lib.mtc_encode_fwd(C.byref(d), _p(a), _p(bt), _p(bias), _p(c), _stream()), args.reps)
        lib.mtc_encode_set_tile(prev)
        # same contraction (the encoder rounds twice and clamps at 0: compare where the plain product is clearly positive)
        pos = ref > 0.05
        row["max_abs_diff_on_positive"] = float((c.float() - ref)[pos].abs().max())
        btT = bt.t()
        row["hipblaslt_ms"] = timed(lambda: torch.matmul(a, btT), args.reps)
        flop = 2.0 * M * N * K
        for k in ("gemm_nt_128_ms", "persistent_256_ms", "hipblaslt_ms"):
            row[k.replace("_ms", "_frac_of_2500")] = flop / row[k] / 1e9 / 2500.0
        out["products"][name] = row
        del a, bt, c, ref
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
