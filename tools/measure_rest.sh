set -o pipefail
TAG=r02b
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 120 python -u tools/encode_stamps.py > gpurun_out/enc_stamps.json 2> gpurun_out/enc_stamps.err || exit 1
timeout -k 10 900 python -u tools/sweep.py > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
timeout -k 10 300 python -u bench.py --no-cpu-baseline --aux-active --steps 30 --warmup 3 > gpurun_out/bench_aux_$TAG.json 2> gpurun_out/bench_aux_$TAG.err || exit 1
timeout -k 10 200 python -u tools/fused_ab.py > gpurun_out/fused_ab_$TAG.txt 2>&1 || exit 1
bash tools/pmc_fused.sh > gpurun_out/pmc_fused_$TAG.txt 2>&1 || exit 1
tail -15 gpurun_out/sweep.log
