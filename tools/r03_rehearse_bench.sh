#!/bin/bash
# bench.py's N > 1 path end to end on a one-GPU box: 2 and 4 ranks on GPU 0, gloo + host-staged messages (L3K_BENCH_REHEARSAL=1);
# the partitioned apply is checked against one rank's apply on the whole mesh.  Numbers of these runs mean nothing.
set -o pipefail
export TMPDIR=/tmp L3K_BENCH_REHEARSAL=1
for n in 2 4; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29540+n)) \
      bench.py --gpus $n --ne ${NE:-24} --steps 3 --warmup 1 || exit 1
done
# round 4: the same commands AS TYPED, without a launcher -- bench.py / bench_config5.py start their ranks themselves (l3ster_amd/launch.py)
for n in 2 4; do
  timeout -k 10 400 python bench.py --gpus $n --ne ${NE:-24} --steps 3 --warmup 1 || exit 1
done
timeout -k 10 400 python tools/bench_config5.py --gpus 2 --ne 6 --order 4 || exit 1
