#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_r02b.json 2> gpurun_out/bench_r02b.err || exit 1
python3 -c "
import json; r=json.load(open('gpurun_out/bench_r02b.json')); print(r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['launch_ms']); print({k: round(v['ms'],3) for k,v in r['stages'].items()})"
