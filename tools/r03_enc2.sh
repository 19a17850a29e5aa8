#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
for lib in matryoshka_transcoder_amd/libmtc_exp_*.so; do
  MTC_LIB=$PWD/$lib timeout -k 10 600 python -u -m pytest tests/test_gpu_widths.py tests/test_gpu_aux_gemm.py -m gpu -x -q --timeout 300 --timeout-method thread -k "256 or persistent or config2 or encoder" > gpurun_out/r03/enc_variant_tests.log 2>&1 || { tail -30 gpurun_out/r03/enc_variant_tests.log; exit 1; }
  tail -2 gpurun_out/r03/enc_variant_tests.log
done
bash tools/r03_enc.sh
