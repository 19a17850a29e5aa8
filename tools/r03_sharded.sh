#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests/test_gpu_sharded.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/sharded_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r03/sharded_tests.log
exit $rc
