#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_marks.py tests/test_gpu_kernels.py -x -q --timeout 200 --timeout-method thread > gpurun_out/r03/marks_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r03/marks_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/r03/bench3.json 2> gpurun_out/r03/bench3.err || { tail gpurun_out/r03/bench3.err; exit 1; }
python3 -c "
import json
r=json.load(open('gpurun_out/r03/bench3.json'))
print(r['ms_per_step'], r['roofline']['frac'], {k:round(v['ms'],3) for k,v in r['stages'].items()}, r['hbm_calibration'])"
