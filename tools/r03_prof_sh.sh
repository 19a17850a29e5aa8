#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
rm -rf gpurun_out/r03/prof_sharded
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_sharded -o s -- python3 bench.py --no-cpu-baseline --sharded --steps 30 --warmup 5 > gpurun_out/r03/prof_sharded.json 2> gpurun_out/r03/prof_sharded.err || { tail -20 gpurun_out/r03/prof_sharded.err; exit 1; }
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r03/prof_sharded/s_kernel_trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "encode_bf16_mfma256p" in r["Kernel_Name"]]
a,b=idx[-14],idx[-13]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b+1]:
    s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0
    print(f'{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} q{r["Queue_Id"]} {r["Kernel_Name"][:64]}')
PY
