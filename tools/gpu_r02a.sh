#!/bin/bash
# Round 2, first GPU call: full GPU suite (with the new parity tests), bench line, encoder yardsticks and PMC evidence.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err || exit 1
timeout -k 10 120 python -u tools/encode_only.py --blas > gpurun_out/enc_random.json 2> gpurun_out/enc_random.err || exit 1
timeout -k 10 120 python -u tools/encode_only.py --zeros > gpurun_out/enc_zeros.json 2> gpurun_out/enc_zeros.err || exit 1
cat gpurun_out/enc_random.json gpurun_out/enc_zeros.json
(rocprofv3 -L > gpurun_out/counters.txt 2>&1 || true)
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_enc_a -o a -- python3 tools/encode_only.py --reps 10 > gpurun_out/pmc_enc_a.log 2>&1 || { tail -5 gpurun_out/pmc_enc_a.log; exit 1; }
echo pmc_a done
