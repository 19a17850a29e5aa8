#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 200 python -u tools/fused_ab.py > gpurun_out/r03/fused_ab.txt 2>&1 || { tail -20 gpurun_out/r03/fused_ab.txt; exit 1; }
cat gpurun_out/r03/fused_ab.txt
timeout -k 10 900 python -u -m pytest tests/test_gpu_sharded.py -x -q --timeout 300 --timeout-method thread -k "aux or gemm or sparse" > gpurun_out/r03/sharded_aux_tests.log 2>&1 || { tail -20 gpurun_out/r03/sharded_aux_tests.log; exit 1; }
tail -3 gpurun_out/r03/sharded_aux_tests.log
