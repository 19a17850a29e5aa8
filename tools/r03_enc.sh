#!/bin/bash
# A/B of encoder build variants (tools/build_variant.sh NAME encode -D...): alternating processes, 3 rounds
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
: > gpurun_out/r03/enc_variants.txt
for round in 1 2 3; do
  for lib in matryoshka_transcoder_amd/libmtc.so matryoshka_transcoder_amd/libmtc_exp_*.so; do
    [ -f $lib ] || continue
    v=$(basename $lib .so)
    r=$(MTC_LIB=$PWD/$lib timeout -k 10 120 python -u tools/encode_only.py --reps 60 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['encode_ms'])") || exit 1
    echo "$round $v $r" | tee -a gpurun_out/r03/enc_variants.txt
  done
done
