#!/bin/bash
# in-box A/B of the fp32 tier's multi-problem weight-gradient launch: -> gpurun_out/x3_ab.txt
set -e
out=gpurun_out/x3_ab.txt; : > $out
runF() { echo "== fp32 $*" | tee -a $out; env "$@" timeout -k 10 120 python bench.py --dtype fp32 --no-infer --no-cpu-baseline --no-roofline --no-fp32 --no-extra --steps 100 --warmup 15 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" | tee -a $out; }
for k in SX_X3_WG_MULTI=0 SX_X3_WG_MULTI=1 SX_X3_WG_BATCH_BLOCKS=64 SX_X3_WG_BATCH_BLOCKS=96 SX_X3_WG_BATCH_BLOCKS=192 SX_X3_WG_MULTI=0; do runF $k; done
