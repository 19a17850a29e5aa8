#!/bin/bash
# A/B of aux_select_dense variants (tools/build_variant.sh NAME aux -D...): kernel medians from an aux-active bench trace
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
for lib in matryoshka_transcoder_amd/libmtc.so matryoshka_transcoder_amd/libmtc_exp_*.so; do
  v=$(basename $lib .so)
  rm -rf gpurun_out/r03/prof_abl
  MTC_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/prof_abl -o p -- python3 bench.py --no-cpu-baseline --aux-active --steps 24 --warmup 4 --settle-steps 16 > gpurun_out/r03/prof_abl.json 2> gpurun_out/r03/prof_abl.err || { tail -20 gpurun_out/r03/prof_abl.err; exit 1; }
  python3 - "$v" <<PY
import csv, sys, json
rows = list(csv.DictReader(open("gpurun_out/r03/prof_abl/p_kernel_trace.csv")))
v = sorted((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "aux_select_dense" in r["Kernel_Name"])
d = json.load(open("gpurun_out/r03/prof_abl.json"))
print(sys.argv[1], "aux_select_dense median %.1f us (n=%d), step %.4f ms" % (v[len(v)//2], len(v), d["ms_per_step"]))
PY
done
