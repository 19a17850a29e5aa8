#!/bin/bash
# PMC passes over tools/fused_ab.py (row vs column-sliced gradient + Adam launches); summaries by tools/pmc_summary.py
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pf_fetch -o f -- python3 tools/fused_ab.py --reps 3 > gpurun_out/pf_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pf_write -o w -- python3 tools/fused_ab.py --reps 3 > gpurun_out/pf_write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pf_sq -o s -- python3 tools/fused_ab.py --reps 3 > gpurun_out/pf_sq.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out/pf_fetch gpurun_out/pf_write gpurun_out/pf_sq
