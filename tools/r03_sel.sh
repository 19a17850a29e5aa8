#!/bin/bash
# window fast path of the carried-bound select: parity tests, then the headline bench and a kernel trace
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 900 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_marks.py tests/test_gpu_fuzz.py tests/test_gpu_widths.py tests/test_gpu_module.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/sel_tests.log 2>&1 || { tail -40 gpurun_out/r03/sel_tests.log; exit 1; }
tail -3 gpurun_out/r03/sel_tests.log
timeout -k 10 300 python -u bench.py --steps 200 --warmup 10 > gpurun_out/r03/bench5.json 2> gpurun_out/r03/bench5.err || { tail -20 gpurun_out/r03/bench5.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench5.json')); print(d['ms_per_step'], d['roofline']['frac'], {k:round(v['ms'],4) for k,v in d['stages'].items()})"
rm -rf gpurun_out/r03/prof_sel
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_sel -o p -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/r03/prof_sel.json 2> gpurun_out/r03/prof_sel.err || { tail -20 gpurun_out/r03/prof_sel.err; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r03/prof_sel/p_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "encode_bf16_mfma256p_kernel<false, true" in r["Kernel_Name"]]
a,b=idx[-12],idx[-11]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
    print(f'{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} q{r["Queue_Id"]} {r["Kernel_Name"][:60]}')
PY
