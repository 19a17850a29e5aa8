#!/bin/bash
# quick: select tests + kernel trace of the select launches
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_marks.py -m gpu -x -q --timeout 300 --timeout-method thread -k "select or marked" > gpurun_out/r03/sel_tests2.log 2>&1 || { tail -40 gpurun_out/r03/sel_tests2.log; exit 1; }
tail -2 gpurun_out/r03/sel_tests2.log
rm -rf gpurun_out/r03/prof_sel
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_sel -o p -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r03/prof_sel.json 2> gpurun_out/r03/prof_sel.err || { tail -20 gpurun_out/r03/prof_sel.err; exit 1; }
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r03/prof_sel/p_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "select_" in n: d[n[:58]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for n, v in d.items():
    v = sorted(v); print(f"{n:60s} n={len(v):3d} median {v[len(v)//2]:7.1f} min {v[0]:7.1f}")
PY
