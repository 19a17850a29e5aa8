#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
: > gpurun_out/r03/enc_variants.txt
for round in 1 2 3; do
  for v in base g2 g8 g16 wnt xnt wxnt; do
    if [ "$v" = base ]; then lib=matryoshka_transcoder_amd/libmtc.so; else lib=matryoshka_transcoder_amd/libmtc_exp_$v.so; fi
    [ -f $lib ] || continue
    r=$(MTC_LIB=$PWD/$lib timeout -k 10 120 python -u tools/encode_only.py --reps 60 2>/dev/null | python3 -c "import json,sys; print(json.load(sys.stdin)['encode_ms'])") || exit 1
    echo "$round $v $r" | tee -a gpurun_out/r03/enc_variants.txt
  done
done
