#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python -u bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err || { tail -20 gpurun_out/bench_2rank.err; exit 1; }
cat gpurun_out/bench_2rank.json
timeout -k 10 400 python -u bench.py > gpurun_out/bench_r02c.json 2> gpurun_out/bench_r02c.err || { tail -20 gpurun_out/bench_r02c.err; exit 1; }
python3 -c "
import json; r=json.load(open('gpurun_out/bench_r02c.json')); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['cpu_baseline'])"
