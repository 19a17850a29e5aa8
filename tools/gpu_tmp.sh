#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_widths.py -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider -k "256 or config2 or chain" > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -1 gpurun_out/t.log
for T in 3 2; do
  timeout -k 10 120 python -u tools/encode_only.py --tile $T --reps 40 || exit 1
done
timeout -k 10 120 python -u tools/encode_only.py --tile 2 --zeros || exit 1
