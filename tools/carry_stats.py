#!/usr/bin/env python3
"""How often the carried lower bound of the TopK threshold serves the select stage (compact path) over a bench-like run,
and how the threshold moves from step to step: per step {path, candidates >= bound, tau}."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 220
    wl = sys.argv[2] if len(sys.argv) > 2 else "gemma2-2b"
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    cfg, B = bench.make_cfg(wl, torch.bfloat16, "cuda:0")
    tc = MatryoshkaTranscoder(cfg)
    opt = FusedAdam(tc, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    pool = bench.batch_pool(B, cfg["act_size"], torch.bfloat16, "cuda:0")
    hp = tc.hot_path()
    prev = None
    full = []
    for i in range(steps):
        opt.train_step(*pool[i % len(pool)])
        c = hp.carry_state(torch.device("cuda:0")).cpu().tolist()
        st = tc._last_res.sel.state.cpu()
        tau = torch.tensor([int(st[4])], dtype=torch.int32).view(torch.float32).item()
        served_full = prev is not None and c[5] > prev[5]
        if served_full or i < 12 or i % 20 == 0:
            print(f"step {i:4d} {'FULL   ' if served_full else 'compact'} candidates {c[2]:8d} overflow rows {c[3]:4d} tau {tau:.5f} "
                  f"(k*B = {cfg['top_k'] * B})", flush=True)
        if served_full:
            full.append(i)
        prev = c
    print(f"{wl}: {len(full)} of {steps} steps on the full path: {full}")


if __name__ == "__main__":
    main()
