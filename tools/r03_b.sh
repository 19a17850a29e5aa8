#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
for v in "" "--comm libmtc" "--world1-exchanges" "--exchange-blocks 4" "--exchange-blocks 4 --world1-exchanges"; do
  tag=$(echo "$v" | tr -d ' -')
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded $v > gpurun_out/r03/bench2_sharded_$tag.json 2> gpurun_out/r03/bench2_sharded_$tag.err || { tail -20 gpurun_out/r03/bench2_sharded_$tag.err; exit 1; }
  python3 -c "
import json,sys
r=json.load(open('gpurun_out/r03/bench2_sharded_$tag.json'))
print('$v', r['ms_per_step'], {k:round(v['ms'],3) for k,v in r['stages'].items()})"
done
timeout -k 10 300 python -u tools/hostcost.py --dist > gpurun_out/r03/hostcost2.txt 2>&1 || { tail -20 gpurun_out/r03/hostcost2.txt; exit 1; }
grep "sharded=" gpurun_out/r03/hostcost2.txt
