#!/bin/bash
# rehearsal of the driver's multi-GPU launch line on the one-GPU box (ranks share the card, gloo transport)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r03
for n in 2 8; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $n --steps 4 --warmup 2 > gpurun_out/r03/rehearse_n$n.json 2> gpurun_out/r03/rehearse_n$n.err || { tail -30 gpurun_out/r03/rehearse_n$n.err; exit 1; }
  python3 -c "
import json
r=json.load(open('gpurun_out/r03/rehearse_n$n.json'))
print($n, r['ms_per_step'], r['config']['final_loss'], r['config']['parallelism'], r['config']['exchange_blocks'], list(r['exchanges'])[:3])"
done
timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 6 --warmup 0 > gpurun_out/r03/rehearse_n1.json 2>/dev/null && python3 -c "
import json
r=json.load(open('gpurun_out/r03/rehearse_n1.json'))
print(1, r['ms_per_step'], r['config']['final_loss'])"
