#!/bin/bash
# aux branch on the persistent tile: tests, product timings, encoder check, aux-active bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_aux_gemm.py tests/test_gpu_kernels.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r03/aux_tests.log 2>&1 || { tail -30 gpurun_out/r03/aux_tests.log; exit 1; }
tail -3 gpurun_out/r03/aux_tests.log
timeout -k 10 200 python -u tools/aux_gemm_ab.py > gpurun_out/r03/aux_gemm_ab2.json 2> gpurun_out/r03/aux_gemm_ab2.err || exit 1
grep -E '"(gemm_nt_128_ms|persistent_256_ms|hipblaslt_ms)"' gpurun_out/r03/aux_gemm_ab2.json
timeout -k 10 120 python -u tools/encode_only.py --reps 60 2>/dev/null || exit 1
timeout -k 10 300 python -u bench.py --aux-active --steps 30 --warmup 3 > gpurun_out/r03/bench_aux2.json 2> gpurun_out/r03/bench_aux2.err || { tail -20 gpurun_out/r03/bench_aux2.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench_aux2.json')); print(d['ms_per_step'], d['aux'])"
timeout -k 10 300 python -u bench.py --steps 100 --warmup 5 > gpurun_out/r03/bench4.json 2> gpurun_out/r03/bench4.err || exit 1
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench4.json')); print(d['ms_per_step'], d['roofline']['frac'])"
