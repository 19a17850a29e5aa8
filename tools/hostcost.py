#!/usr/bin/env python3
"""Host-side issue cost of one training step (no device sync inside the timed loop) next to its wall time: tells
whether a configuration is GPU-bound or launch-bound.  `--dist` initialises an RCCL process group first (world 1)."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dist", action="store_true")
    ap.add_argument("--profile", default="")
    args = ap.parse_args()
    if args.dist:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    variants = [("gpt2", False), ("gpt2", False), ("gpt2-noaux", False), ("gemma2-2b", False)]
    if args.dist:
        variants += [("gpt2-noaux", True), ("gpt2", True), ("gpt2", True), ("gemma2-2b", True)]
    for wl, sharded in variants:
        cfg, B = bench.make_cfg(wl.replace("-noaux", ""), torch.bfloat16, "cuda:0")
        if wl.endswith("-noaux"):
            cfg["n_batches_to_dead"] = 10 ** 9       # never dead: the aux branch (and its read-back) stays off
        if sharded:
            from matryoshka_transcoder_amd.sharded import ShardedFusedAdam, ShardedMatryoshkaTranscoder
            tc = ShardedMatryoshkaTranscoder(cfg)
            opt = ShardedFusedAdam(tc, lr=3e-4, betas=(0.9, 0.99))
        else:
            tc = MatryoshkaTranscoder(cfg)
            opt = FusedAdam(tc, lr=3e-4, betas=(0.9, 0.99))
        pool = bench.batch_pool(B, cfg["act_size"], torch.bfloat16, "cuda:0", n=8)
        for i in range(5):
            opt.train_step(*pool[i % 8])
        torch.cuda.synchronize()
        prof = cProfile.Profile() if args.profile == f"{wl}:{int(sharded)}" else None
        t0 = time.perf_counter()
        if prof:
            prof.enable()
        for i in range(50):
            opt.train_step(*pool[i % 8])
        if prof:
            prof.disable()
        t_host = (time.perf_counter() - t0) / 50
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / 50
        print(f"{wl:10s} sharded={sharded}: host issue {t_host * 1e3:.3f} ms/step, wall {t_all * 1e3:.3f} ms/step", flush=True)
        if prof:
            pstats.Stats(prof).sort_stats("tottime").print_stats(28)
        del tc, opt, pool
    if args.dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
