#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 120 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider 2>&1 | tail -2
