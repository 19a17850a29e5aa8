#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sh -o sh -- python3 bench.py --sharded --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/prof_sh.json 2> gpurun_out/prof_sh.err || { tail -5 gpurun_out/prof_sh.err; exit 1; }
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_sh/sh_kernel_stats.csv')))
tot=0
for r in rows[:40]:
    n=int(r['Calls']); 
    print(f"{r['Name'][:90]:90s} calls/step {n/35:5.1f} avg {float(r['AverageNs'])/1e3:8.1f} us  per-step {float(r['TotalDurationNs'])/35/1e3:8.1f}")
    tot+=float(r['TotalDurationNs'])/35/1e3
print('sum per step us', tot, 'launches per step', sum(int(r['Calls']) for r in rows)/35)
PY
