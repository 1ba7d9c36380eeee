#!/bin/bash
# in-box A/B of weight-gradient launch plans on cfg 5 and cfg A: tools/wg_ab.sh  -> gpurun_out/wg_ab.txt
set -e
out=gpurun_out/wg_ab.txt; : > $out
run5() { echo "== cfg5 $*" | tee -a $out; env "$@" timeout -k 10 120 python tools/cfg5_bench.py 32 12 bf16 2>&1 | tail -1 | tee -a $out; }
runA() { echo "== cfgA $*" | tee -a $out; env "$@" timeout -k 10 120 python bench.py --no-infer --no-cpu-baseline --no-roofline --no-fp32 --no-extra --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" | tee -a $out; }
for k in SX_WG_MIN_SPLITS=1 A=0 SX_WG_SORT_BLOCK=0 SX_WG_BLOCK_MFLOP=200 SX_WG_FINE_GF=30 A=0; do run5 $k; done
for k in SX_WG_MIN_SPLITS=1 A=0 SX_WG_SORT_BLOCK=0 A=0; do runA $k; done
