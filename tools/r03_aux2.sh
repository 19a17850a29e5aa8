#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 600 python -u -m pytest tests/test_gpu_aux_gemm.py tests/test_gpu_kernels.py tests/test_gpu_module.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r03/aux_tests2.log 2>&1 || { tail -30 gpurun_out/r03/aux_tests2.log; exit 1; }
tail -2 gpurun_out/r03/aux_tests2.log
for i in 1 2; do
timeout -k 10 300 python -u bench.py --no-cpu-baseline --aux-active --steps 30 --warmup 3 > gpurun_out/r03/bench_aux3.json 2> gpurun_out/r03/bench_aux3.err || { tail -20 gpurun_out/r03/bench_aux3.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r03/bench_aux3.json')); print(d['ms_per_step'], d['select_paths'], d['aux']['stage_ms'])"
done
rm -rf gpurun_out/r03/prof_aux
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_aux -o p -- python3 bench.py --no-cpu-baseline --aux-active --steps 30 --warmup 5 > gpurun_out/r03/prof_aux.json 2> gpurun_out/r03/prof_aux.err || { tail -20 gpurun_out/r03/prof_aux.err; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r03/prof_aux/p_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "encode_bf16_mfma256p_kernel<false, true" in r["Kernel_Name"]]
a,b=idx[-5],idx[-4]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
    if s/1e3 > 850: print(f'{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} q{r["Queue_Id"]} {r["Kernel_Name"][:60]}')
PY
