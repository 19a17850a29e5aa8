#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests/test_gpu_sharded.py tests/test_gpu_kernels.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/sh_tests2.log 2>&1 || { tail -40 gpurun_out/r03/sh_tests2.log; exit 1; }
tail -2 gpurun_out/r03/sh_tests2.log
for extra in "" "--comm libmtc" "--world1-exchanges"; do
timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded $extra > gpurun_out/r03/bench_sh2.json 2> gpurun_out/r03/bench_sh2.err || { tail -20 gpurun_out/r03/bench_sh2.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench_sh2.json')); print('$extra', d['ms_per_step'], d['config']['final_loss'])"
done
