#!/usr/bin/env python3
"""the REFERENCE's own training step timed on the build container's CPU cores, with the oracle's
dense port (oracle.DenseStep, what bench.py's `cpu_baseline` leg times on the GPU box's host) timed beside it.

Runs only where /root/reference exists (the reference does not travel to the GPU box).  The reference class is
imported unmodified through the same in-memory `transformer_lens` stub tools/make_golden.py uses; the step body is
the five lines of training.py:172-183 around it (forward, backward(retain_graph=True), decoder unit-norm projection,
torch.optim.Adam step, zero_grad).  Synthetic N(0, 1) inputs, seed-49 init, input_unit_norm False.

    python tools/time_reference.py [--configs 1,2] [--dtypes fp32,bf16] > profiles/r02_cpu_reference.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from make_golden import import_reference  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

CONFIGS = {"1": ("gpt2", 50, 2), "2": ("gemma2-2b", 3, 1)}        # workload, timed steps, warm-up steps


def time_steps(step, pool, warm, n):
    for i in range(warm):
        step(*pool[i % len(pool)])
    t0 = time.perf_counter()
    for i in range(n):
        step(*pool[(warm + i) % len(pool)])
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2")
    ap.add_argument("--dtypes", default="fp32,bf16")
    args = ap.parse_args()
    sae = import_reference()
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    rows = []
    for c in args.configs.split(","):
        workload, n, warm = CONFIGS[c]
        for dn in args.dtypes.split(","):
            dt = torch.float32 if dn == "fp32" else torch.bfloat16
            cfg, B = bench.make_cfg(workload, dt, "cpu")
            pool = bench.batch_pool(B, cfg["act_size"], dt, "cpu", n=2)
            m = sae.MatryoshkaTranscoder(cfg)
            opt = torch.optim.Adam(m.parameters(), lr=cfg["lr"], betas=(cfg["beta1"], cfg["beta2"]))

            def ref_step(xs, xt):
                out = m(xs, xt)
                out["loss"].backward(retain_graph=True)            # training.py:172
                m.make_decoder_weights_and_grad_unit_norm()        # :175
                opt.step(); opt.zero_grad()                        # :178-179
                return out["loss"].detach()

            t_ref = time_steps(ref_step, pool, warm, n)
            del m, opt
            port = R.DenseStep(cfg)
            t_port = time_steps(port.step, pool, warm, n)
            del port
            row = {"config": c, "workload": workload, "dtype": dn, "B": B, "D": cfg["prefix_sizes"][-1], "k": cfg["top_k"],
                   "steps_timed": n, "cores": cores, "torch": torch.__version__,
                   "reference_s_per_step": t_ref, "reference_tokens_per_s": B / t_ref,
                   "oracle_dense_port_s_per_step": t_port, "oracle_dense_port_tokens_per_s": B / t_port,
                   "port_over_reference_time": t_port / t_ref}
            rows.append(row)
            print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps({"what": "reference train step (imported from /root/reference) vs oracle.DenseStep, build container CPU",
                      "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
