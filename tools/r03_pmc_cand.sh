#!/bin/bash
# wave-cycle breakdown of the select kernels (candidate pass, window tail) under the headline bench
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
rm -rf gpurun_out/r03/pmc_cand
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d gpurun_out/r03/pmc_cand -o c -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --settle-steps 8 > gpurun_out/r03/pmc_cand.log 2>&1 || { tail -5 gpurun_out/r03/pmc_cand.log; exit 1; }
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/r03/pmc_cand/c_counter_collection.csv")))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "select_" in n or "decode_fwd" in n or "fused_" in n or "encode_bf16" in n:
        acc[n[:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    med = {k: sorted(v)[len(v)//2] for k, v in c.items()}
    wc = med.get("SQ_WAVE_CYCLES", 1) or 1
    print(n)
    print("   ", {k: round(v) for k, v in med.items()})
    print("    shares of wave cycles: VALU active %.2f, any inst active %.2f, waiting %.2f; VALU insts per wave %.0f" % (
        med.get("SQ_ACTIVE_INST_VALU", 0) / wc, med.get("SQ_ACTIVE_INST_ANY", 0) / wc, med.get("SQ_WAIT_INST_ANY", 0) / wc,
        med.get("SQ_INSTS_VALU", 0) / max(med.get("SQ_WAVES", 1), 1)))
PY
