#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_sharded -o s -- python3 bench.py --no-cpu-baseline --sharded --steps 30 --warmup 5 > gpurun_out/r03/prof_sharded.json 2> gpurun_out/r03/prof_sharded.err || { tail -20 gpurun_out/r03/prof_sharded.err; exit 1; }
find gpurun_out/r03/prof_sharded -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03/sharded_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r03/sharded_kernel_stats.csv")))
for r in rows[:45]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.2f}')
PY
