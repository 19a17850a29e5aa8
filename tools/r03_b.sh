#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded > gpurun_out/r03/bench1_sharded.json 2> gpurun_out/r03/bench1_sharded.err || { tail -20 gpurun_out/r03/bench1_sharded.err; exit 1; }
timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded --comm libmtc > gpurun_out/r03/bench1_sharded_lib.json 2> gpurun_out/r03/bench1_sharded_lib.err || { tail -20 gpurun_out/r03/bench1_sharded_lib.err; exit 1; }
timeout -k 10 300 python -u tools/hostcost.py --dist --profile gemma2-2b:1 > gpurun_out/r03/hostcost1.txt 2>&1 || { tail -20 gpurun_out/r03/hostcost1.txt; exit 1; }
python3 - <<PY
import json
for f in ("bench1_sharded", "bench1_sharded_lib"):
    r = json.load(open(f"gpurun_out/r03/{f}.json"))
    print(f, r["ms_per_step"], r["value"])
PY
grep -E "sharded=|tottime|\.py|\{" gpurun_out/r03/hostcost1.txt | head -60
