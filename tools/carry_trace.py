#!/usr/bin/env python3
"""Prints the select's carried-bound words after every training step (diagnosis of which steps take the full path):
    python tools/carry_trace.py [--aux-active] [--steps 40]
carry = [bound bits, flag, candidates, overflowing rows, steps served compact, steps served full, ...]."""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--aux-active", action="store_true")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--pool", type=int, default=8)
    args = ap.parse_args()
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    A, groups, k, B = bench.WORKLOADS["gemma2-2b"]
    dev = torch.device("cuda:0")
    cfg, _ = bench.make_cfg("gemma2-2b", torch.bfloat16, dev)
    tc = MatryoshkaTranscoder(cfg)
    opt = FusedAdam(tc, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    if args.aux_active:
        forced = torch.arange(5, tc.hot_path().D, 7, device=dev)
        with torch.no_grad():
            tc.W_enc[:, forced] *= 0.01
        cnt = tc.num_batches_not_active
        cnt[forced] = float(cfg["n_batches_to_dead"])
        tc.num_batches_not_active = cnt
    pool = bench.batch_pool(B, A, torch.bfloat16, dev, n=args.pool)
    hp = tc.hot_path()
    for i in range(args.steps):
        xs, xt = pool[i % len(pool)]
        out = opt.train_step(xs, xt)
        torch.cuda.synchronize()
        c = hp.carry_state(dev).cpu().tolist()
        thr = float(tc.threshold)
        print(i, "bound_bits %#x flag %d cand %d over %d compact %d full %d" % (c[0] & 0xffffffff, c[1], c[2], c[3], c[4], c[5]),
              "threshold %.5f loss %.4f" % (thr, float(out["loss"])), flush=True)

if __name__ == "__main__":
    main()
