#!/bin/bash
# Control-flow rehearsal of `bench.py --gpus N` at the real problem size with N ranks on ONE GPU (loopback transport, see
# tests/loopback/): meshes, halo plans, default kernel choice and exchange schedule of every rank, the JSON line.  NOT a measurement.
#   bash tools/rehearse_bench_ranks.sh 4 [extra bench.py arguments]
set -e
N=${1:-2}; shift || true
export BP5_LIB=$PWD/deal-and-ceed-on-gpu_amd/libbp5_loopback.so
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N \
  --steps 10 --warmup 3 --sustained-iters 0 --rehearsal "$@"
