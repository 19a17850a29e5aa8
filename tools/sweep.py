#!/usr/bin/env python3
"""Single-GPU sweep over every BASELINE config and the variants SURVEY 8d asks for once: dtype, input_unit_norm,
the README lists taken literally, a step with the auxiliary loss active (forced-dead features), the sharded
classes at world 1, plus an HBM copy calibration.  One process; each variant: 5 warm-up + 20 timed
`FusedAdam.train_step` calls between two stream events.  Writes gpurun_out/sweep.json and prints a table.

    python tools/sweep.py [--only REGEX]
"""
import argparse
import gc
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import bench  # noqa: E402

VARIANTS = [
    # name, workload, dtype, unit_norm, forced-dead aux
    ("gpt2 bf16", "gpt2", "bf16", False, False),
    ("gpt2 fp32", "gpt2", "fp32", False, False),
    ("gpt2 literal-list bf16", "gpt2-literal", "bf16", False, False),
    ("gemma2-2b bf16 (headline)", "gemma2-2b", "bf16", False, False),
    ("gemma2-2b bf16 unit-norm", "gemma2-2b", "bf16", True, False),
    ("gemma2-2b fp32", "gemma2-2b", "fp32", False, False),
    ("gemma2-2b fp32 unit-norm", "gemma2-2b", "fp32", True, False),
    ("gemma2-2b literal-list bf16", "gemma2-2b-literal", "bf16", False, False),
    ("gemma2-2b bf16 aux active (1/7 of the features forced dead)", "gemma2-2b", "bf16", False, True),
    ("gemma2-9b bf16", "gemma2-9b", "bf16", False, False),
    ("gemma2-27b bf16", "gemma2-27b", "bf16", False, False),
    ("gemma2-27b 2x dictionary bf16", "gemma2-27b-2x", "bf16", False, False),
]


def run(workload, dt_name, unit, aux, steps=20, warm=5):
    from matryoshka_transcoder_amd.training import FusedAdam
    from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder
    dt = torch.bfloat16 if dt_name == "bf16" else torch.float32
    cfg, B = bench.make_cfg(workload, dt, "cuda:0", unit)
    A, D, k = cfg["act_size"], cfg["prefix_sizes"][-1], cfg["top_k"]
    t0 = time.time()
    tc = MatryoshkaTranscoder(cfg)
    if aux:
        dead = torch.arange(5, D, 7, device="cuda:0")
        with torch.no_grad():
            tc.W_enc[:, dead] *= 0.01                      # never wins the global TopK again
        cnt = tc.num_batches_not_active
        cnt[dead] = float(cfg["n_batches_to_dead"])
    opt = FusedAdam(tc, lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))
    pool = bench.batch_pool(B, A, dt, "cuda:0", n=4)
    init_s = time.time() - t0
    for i in range(warm):
        out = opt.train_step(*pool[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        out = opt.train_step(*pool[i % 4])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    rec = {"workload": workload, "dtype": dt_name, "input_unit_norm": unit, "aux_forced_dead": aux, "A": A, "D": D, "k": k,
           "B": B, "ms_per_step": ms, "tokens_per_s": B / ms * 1e3, "loss": float(out["loss"].detach().float()),
           "l0": float(out["l0_norm"]), "aux_loss": float(out["aux_loss"].float()), "fvu": float(out["fvu"]),
           "dead_features": int((tc.num_batches_not_active >= cfg["n_batches_to_dead"]).sum()),
           "encoder_TFLOPs_if_alone": None, "init_s": init_s, "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9}
    del tc, opt, pool, out
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    return rec


def copy_calibration():
    n = 2 << 30
    src = torch.empty(n, dtype=torch.uint8, device="cuda:0").random_(0, 255)
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    del src, dst
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    own = bench.hbm_calibration("cuda:0", gib=2.0)
    return {"bytes_each_way": n, "torch_copy_ms": ms, "torch_copy_read_plus_write_GBps": 2 * n / ms / 1e6,
            "mtc_copy16_ms": own["ms"], "mtc_copy16_read_plus_write_GBps": own["GBps_read_plus_write"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    out = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "variants": {}}
    out["hbm_copy_calibration"] = copy_calibration()
    print("copy calibration:", json.dumps(out["hbm_copy_calibration"]), flush=True)
    for name, wl, dt, unit, aux in VARIANTS:
        if args.only and not re.search(args.only, name):
            continue
        rec = run(wl, dt, unit, aux)
        out["variants"][name] = rec
        print(f"{name:62s} {rec['ms_per_step']:8.3f} ms/step {rec['tokens_per_s'] / 1e6:7.3f} M tok/s  loss {rec['loss']:.4f} "
              f"l0 {rec['l0']:.1f} dead {rec['dead_features']} mem {rec['peak_mem_GB']:.1f} GB", flush=True)
        with open(os.path.join(ROOT, "gpurun_out", "sweep.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
