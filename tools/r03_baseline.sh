#!/bin/bash
# round-3 baseline on the GPU box: bench (single-GPU / sharded world 1), host issue cost
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/r03/bench0.json 2> gpurun_out/r03/bench0.err || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded > gpurun_out/r03/bench0_sharded.json 2> gpurun_out/r03/bench0_sharded.err || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded --comm libmtc > gpurun_out/r03/bench0_sharded_lib.json 2> gpurun_out/r03/bench0_sharded_lib.err || exit 1
timeout -k 10 300 python -u tools/hostcost.py --dist > gpurun_out/r03/hostcost0.txt 2>&1 || exit 1
python3 - <<PY
import json
for f in ("bench0", "bench0_sharded", "bench0_sharded_lib"):
    r = json.load(open(f"gpurun_out/r03/{f}.json"))
    print(f, r["ms_per_step"], r["value"])
PY
cat gpurun_out/r03/hostcost0.txt
