#!/bin/bash
# two gloo ranks of bench.py on ONE GPU (rehearsal of the data-parallel flow on a one-card box); args are passed to bench.py
# usage: tools/dp_rehearsal.sh <seconds> [bench args...]   -> gpurun_out/dp_rank{0,1}.log (python stacks on timeout)
secs=$1; shift
export SX_DIST_BACKEND=gloo MASTER_ADDR=127.0.0.1 MASTER_PORT=29573 WORLD_SIZE=2
mkdir -p gpurun_out
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 $secs python -X faulthandler bench.py --gpus 2 "$@" > gpurun_out/dp_rank$r.log 2>&1 &
done
wait
tail -c 400 gpurun_out/dp_rank0.log
