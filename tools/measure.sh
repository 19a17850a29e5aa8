#!/bin/bash
# Runs on the GPU box (gpurun -- 'bash tools/measure.sh TAG'): GPU tests, smoke, bench with cpu_baseline, rocprofv3 kernel
# stats and the two PMC passes; everything lands under gpurun_out/.  tools/collect_profiles.py TAG then copies the
# summaries into profiles/.
set -o pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -20 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { tail -5 gpurun_out/smoke.log; exit 1; }
tail -1 gpurun_out/smoke.log
timeout -k 10 300 python -u bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python3 bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/bench_prof.json 2> gpurun_out/bench_prof.err || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > gpurun_out/pmc_write.log 2>&1 || exit 1
# encoder evidence: clock held (GRBM_GUI_ACTIVE / 8 / duration), MFMA-pipe busy share, wave wait breakdown, LDS conflicts
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_enc -o e -- python3 tools/encode_only.py --reps 10 > gpurun_out/pmc_enc.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_enc -o e -- python3 tools/encode_only.py --reps 10 > gpurun_out/prof_enc.log 2>&1 || exit 1
timeout -k 10 120 python -u tools/encode_only.py --blas > gpurun_out/enc_random.json 2> gpurun_out/enc_random.err || exit 1
timeout -k 10 120 python -u tools/encode_only.py --zeros > gpurun_out/enc_zeros.json 2> gpurun_out/enc_zeros.err || exit 1
# in-kernel clock / per-wave cycle stamps: needs the diagnostic build (tools/encode_stamps.sh; kept out of the push set by
# .gpurunignore unless a measurement pass wants it)
if [ -f matryoshka_transcoder_amd/libmtc_diag.so ]; then
  timeout -k 10 120 python -u tools/encode_stamps.py > gpurun_out/enc_stamps.json 2> gpurun_out/enc_stamps.err || exit 1
fi
timeout -k 10 900 python -u tools/sweep.py > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
# the step with the auxiliary loss active (its products on mtc_gemm_nt), and the fused gradient + Adam launches A/B
timeout -k 10 300 python -u bench.py --no-cpu-baseline --aux-active --steps 30 --warmup 3 > gpurun_out/bench_aux_$TAG.json 2> gpurun_out/bench_aux_$TAG.err || exit 1
timeout -k 10 200 python -u tools/fused_ab.py > gpurun_out/fused_ab_$TAG.txt 2>&1 || exit 1
bash tools/pmc_fused.sh > gpurun_out/pmc_fused_$TAG.txt 2>&1 || exit 1
# the feature-sharded classes over RCCL with a one-rank group (fixed overhead of the sharded step), host issue cost
timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded > gpurun_out/bench_sharded_$TAG.json 2> gpurun_out/bench_sharded_$TAG.err || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded --comm libmtc > gpurun_out/bench_sharded_lib_$TAG.json 2> gpurun_out/bench_sharded_lib_$TAG.err || exit 1
timeout -k 10 300 python -u bench.py --no-cpu-baseline --sharded --world1-exchanges > gpurun_out/bench_sharded_x_$TAG.json 2> gpurun_out/bench_sharded_x_$TAG.err || exit 1
timeout -k 10 300 python -u tools/hostcost.py --dist > gpurun_out/hostcost_$TAG.txt 2>&1 || exit 1
python3 - <<PY
import json
r = json.load(open("gpurun_out/bench_$TAG.json"))
print(r["value"], r["ms_per_step"], r["roofline"]["frac"], r["cpu_baseline"]["value"])
print({k: round(v["ms"], 3) for k, v in r["stages"].items()})
PY
