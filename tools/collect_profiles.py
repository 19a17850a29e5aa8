#!/usr/bin/env python3
"""Copies the summaries tools/measure.sh left under gpurun_out/ into profiles/ (kernel stats CSV, bench line,
per-kernel HBM-side traffic from the FETCH_SIZE / WRITE_SIZE passes with the gfx950 correction)."""
import collections
import csv
import json
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = {}
for kind, path in (("fetch", "gpurun_out/pmc_fetch/f_counter_collection.csv"), ("write", "gpurun_out/pmc_write/w_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[kind] = sum(v) / len(v)
        out[k]["n"] = len(v)
kernels = {}
for k, v in out.items():
    f, w = v.get("fetch", 0.0), v.get("write", 0.0)
    if f + w < 1000:
        continue
    kernels[k.split("(")[0].replace("void ", "")] = {
        "fetch_KiB_raw": round(f, 1), "write_KiB_raw": round(w, 1),
        "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024), "launches_sampled": v["n"]}
doc = {
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 bench.py "
               "--no-cpu-baseline --steps 4 --warmup 2",
    "correction": "MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 128-B requests at 64 B on gfx950 -> x2 for wide "
                  "coalesced reads; WRITE_SIZE exact for 16-B-per-lane stores; both in KiB.  Checked on the select passes: "
                  "they read acts once (301,989,888 B) and report ~147.6k KiB raw.  Infinity Cache hits are counted "
                  "(memory side of L2), so these are fabric bytes, an upper bound of HBM bytes.",
    "workload": "gemma2-2b bf16, B 4096", "kernels": kernels}
json.dump(doc, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
shutil.copy(f"gpurun_out/prof_{tag}/{tag}_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")
shutil.copy(f"gpurun_out/bench_{tag}.json", f"profiles/{tag}_bench_line.json")
shutil.copy("gpurun_out/sweep.json", f"profiles/{tag}_sweep.json")
# encoder evidence (tools/measure.sh): PMC pass + kernel duration of the SAME short program -> clock held, MFMA-pipe busy
import os
if os.path.exists("gpurun_out/pmc_enc/e_counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("gpurun_out/pmc_enc/e_counter_collection.csv")):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = {}
    for r in csv.DictReader(open("gpurun_out/prof_enc/e_kernel_stats.csv")):
        dur[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"])
    enc = {}
    for k, c in agg.items():
        if "encode" not in k:
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        ns = dur.get(k)
        e = {"counters_avg_per_launch": m, "launches": len(next(iter(c.values()))), "avg_ns_kernel_trace": ns}
        if ns and "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0                      # rocprofv3 sums the 8 XCDs
            e["cycles_per_xcd"] = cyc
            e["clock_GHz"] = cyc / ns
            e["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc)     # 1024 SIMDs
            e["mfma_bound_ms_at_held_clock"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / e["clock_GHz"] * 1e-6
        if "SQ_WAVE_CYCLES" in m:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                e[n + "_share_of_wave_cycles"] = m[n] / m["SQ_WAVE_CYCLES"]
        enc[k] = e
    extra = {}
    for name in ("enc_random", "enc_zeros", "enc_stamps"):
        try:
            extra[name] = json.load(open(f"gpurun_out/{name}.json"))
        except Exception:
            pass
    json.dump({"command": "rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                          "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/encode_only.py "
                          "--reps 10 (config #2 operands, random data, back-to-back launches) and rocprofv3 --kernel-trace --stats "
                          "of the same program for the duration",
               "kernels": enc, "yardsticks": extra}, open(f"profiles/{tag}_encode_pmc.json", "w"), indent=1)
for src, dst in ((f"gpurun_out/bench_aux_{tag}.json", f"profiles/{tag}_bench_aux_active.json"),
                 (f"gpurun_out/fused_ab_{tag}.txt", f"profiles/{tag}_fused_ab.txt"),
                 (f"gpurun_out/pmc_fused_{tag}.txt", f"profiles/{tag}_fused_pmc.txt")):
    if os.path.exists(src):
        shutil.copy(src, dst)
print("profiles updated for", tag)
