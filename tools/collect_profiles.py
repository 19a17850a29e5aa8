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
for src, dst in ((f"gpurun_out/bench_sharded_{tag}.json", f"profiles/{tag}_bench_sharded_world1.json"),
                 (f"gpurun_out/bench_sharded_lib_{tag}.json", f"profiles/{tag}_bench_sharded_world1_libmtc.json"),
                 (f"gpurun_out/bench_sharded_x_{tag}.json", f"profiles/{tag}_bench_sharded_world1_with_self_exchanges.json"),
                 (f"gpurun_out/hostcost_{tag}.txt", f"profiles/{tag}_hostcost.txt"),
                 (f"gpurun_out/bench_aux_{tag}.json", f"profiles/{tag}_bench_aux_active.json"),
                 (f"gpurun_out/fused_ab_{tag}.txt", f"profiles/{tag}_fused_ab.txt"),
                 (f"gpurun_out/pmc_fused_{tag}.txt", f"profiles/{tag}_fused_pmc.txt")):
    if os.path.exists(src):
        shutil.copy(src, dst)
# ---- one table per round: every kernel of the headline step with its algorithmic work at config #2, the
# rocprofv3 average duration, the fraction of the bound it is priced against, and the PMC fabric traffic next to the
# bytes it must move at least once
B, A, D, K_TOP, S_ = 4096, 2304, 36864, 96, 2
NNZ = K_TOP * B
HBM_PEAK, MFMA_PEAK = 8.0e12, 2.5e15
work = {   # kernel-name prefix -> (bound, algorithmic work, compulsory bytes or None, note)
    "mtc::encode_bf16_mfma256p_kernel": ("mfma", 2.0 * B * A * D, B * A * S_ + D * A * S_ + B * D * S_ + B * D // 128,
                                         "2 B A D FLOP; operands once + acts and candidate marks written"),
    "mtc::select_candidates_kernel": ("latency", 0, 0, "reads only the 32-byte granules of acts the encoder's epilogue flagged (~8 %; "
                                      "the first pass over the B D activations happens there); see pmc_fabric_bytes"),
    "mtc::decode_fwd_kernel": ("hbm", NNZ * (A * S_ + 8) + 2 * B * A * S_, D * A * S_ + 3 * B * A * S_ + NNZ * 6,
                               "gather of nnz decoder rows + target in / x_hat, g out"),
    "mtc::fused_grad_adam_kernel": ("hbm", 6 * D * A * S_ + NNZ * A * S_ + B * A * S_, 6 * D * A * S_ + B * A * S_ + NNZ * 12,
                                    "W_dec / moment rows read + written once, gather of nnz rows of g"),
    "mtc::fused_enc_sliced_kernel": ("hbm", 6 * D * A * S_ + NNZ * A * S_ + B * A * S_, 6 * D * A * S_ + B * A * S_ + NNZ * 8,
                                     "W_enc / moment rows read + written once, gather of nnz rows of x (L2-resident slices)"),
    "mtc::csc_": ("latency", NNZ * 16, NNZ * 16, "CSC build: count / scan / fill / sort, summed"),
    "mtc::colsum_": ("hbm", B * A * S_, B * A * S_, "db_dec column sums, both launches"),
    "mtc::select_": ("latency", 0, 0, "exact select behind the candidate pass, summed: window tail (tau, quota, offsets from per-row window counts; one workgroup), "
                     "two predicated fallback launches that leave at once, ordered fill"),
}
stats = {}
for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")):
    stats[r["Name"].split("(")[0].replace("void ", "")] = (float(r["AverageNs"]), int(r["Calls"]), float(r["TotalDurationNs"]))
steps = max((c for n, (a, c, t) in stats.items() if n.startswith("mtc::encode_bf16_mfma256p_kernel")), default=1)
rows = []
for prefix, (bound, alg, comp, note) in work.items():
    names = [n for n in stats if n.startswith(prefix) and not (prefix == "mtc::select_" and n.startswith("mtc::select_candidates"))]
    if not names:
        continue
    per_step_us = sum(stats[n][2] for n in names) / steps / 1e3
    traffic = sum(kernels.get(n, {}).get("hbm_bytes_per_launch", 0) * stats[n][1] / steps for n in names) or None
    row = {"kernel": prefix + ("*" if len(names) > 1 or names[0] != prefix else ""), "launches_per_step": round(sum(stats[n][1] for n in names) / steps, 2),
           "us_per_step": round(per_step_us, 1), "bound": bound, "note": note}
    if bound == "mfma":
        row.update(algorithmic_flop=alg, achieved_TFLOPs=round(alg / per_step_us / 1e6, 1), frac_of_peak=round(alg / per_step_us / 1e-6 / MFMA_PEAK, 3))
    elif bound == "hbm":
        row.update(algorithmic_bytes=int(alg), achieved_TBps=round(alg / per_step_us / 1e6, 2), frac_of_peak=round(alg / (per_step_us * 1e-6) / HBM_PEAK, 3),
                   compulsory_bytes=int(comp), frac_compulsory=round(comp / (per_step_us * 1e-6) / HBM_PEAK, 3))
    if traffic:
        row.update(pmc_fabric_bytes=int(traffic))
        if comp:
            row.update(traffic_over_compulsory=round(traffic / comp, 2))
    rows.append(row)
json.dump({"workload": "gemma2-2b bf16 (config #2): B 4096, A 2304, D 36864, k 96", "peaks": {"hbm_Bps": HBM_PEAK, "mfma_bf16_FLOPs": MFMA_PEAK},
           "sources": [f"profiles/{tag}_kernel_stats.csv (rocprofv3 --kernel-trace --stats)", f"profiles/{tag}_pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE)"],
           "definitions": "algorithmic = SURVEY 8d per-unit work (row gathers counted per selected entry); compulsory = bytes that must "
                          "cross HBM at least once; pmc_fabric_bytes = L2's memory-side traffic (Infinity Cache hits included)",
           "kernels": rows}, open(f"profiles/{tag}_roofline.json", "w"), indent=1)
print("profiles updated for", tag)
