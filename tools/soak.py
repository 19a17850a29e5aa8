#!/usr/bin/env python3
"""End-to-end soak of train_transcoder on synthetic activations at GPT-2 sizes: LR schedule, perf logging, checkpoints,
sample collection, dead features coming and going.  Prints the training log's first / last records and the files written."""
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from matryoshka_transcoder_amd.activation_store import TranscoderActivationsStore  # noqa: E402
from matryoshka_transcoder_amd.training import train_transcoder  # noqa: E402
from matryoshka_transcoder_amd.transcoder import MatryoshkaTranscoder  # noqa: E402


class Tok:
    def decode(self, ids):
        return " ".join(str(int(i)) for i in ids)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
    dt = torch.bfloat16
    cfg, B = bench.make_cfg("gpt2", dt, "cuda:0", unit_norm=True)
    tmp = tempfile.mkdtemp()
    cfg.update(num_tokens=B * steps, model_name="soak", layer=8, results_root=tmp, scheduler_type="warmup_decay",
               warmup_steps=50, perf_log_freq=100, checkpoint_freq=200, log_freq=50, save_activation_samples=True,
               sample_collection_freq=150, max_samples_per_feature=16, sample_context_size=8, model_batch_size=32,
               seq_len=64, num_batches_in_buffer=4, top_features_to_save=10, samples_per_feature_to_save=4)
    tc = MatryoshkaTranscoder(cfg)
    store = TranscoderActivationsStore(None, cfg)

    g = torch.Generator("cpu").manual_seed(1)

    def with_tokens():
        return (torch.randn(32 * 64, cfg["act_size"], generator=g).to(dt).cuda(),
                torch.randint(0, 50000, (32, 64), generator=g).cuda())
    store.get_batch_with_tokens = with_tokens
    store.tokenizer = Tok()
    t0 = time.time()
    exp = train_transcoder(tc, store, None, cfg)
    torch.cuda.synchronize()
    dt_s = time.time() - t0
    log = [json.loads(l) for l in open(exp / "train_log.jsonl")]
    steps_log = [r for r in log if "loss" in r]
    print("first:", steps_log[0])
    print("last :", steps_log[-1])
    print("perf :", [r for r in log if "performance/mse" in r][-1])
    print(f"{steps} steps in {dt_s:.1f} s (wall; the synthetic source draws its activations on the CPU, which is most of it)")
    for root, _, files in os.walk(exp):
        print(os.path.relpath(root, exp), sorted(files)[:6], len(files))
    assert all(torch.isfinite(p).all() for p in tc.parameters())
    assert steps_log[-1]["loss"] < steps_log[0]["loss"]


if __name__ == "__main__":
    main()
