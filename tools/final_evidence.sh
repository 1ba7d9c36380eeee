#!/bin/bash
# on the GPU box: every profile / table the round's DESIGN.md cites -> gpurun_out/<tag>_*   (usage: tools/final_evidence.sh r03_final)
tag=${1:-r03_final}
root=$GRAFT_REPO_ROOT
B="--no-infer --no-cpu-baseline --no-roofline --no-fp32 --no-extra"
bash tools/prof_step.sh $tag || exit 1
PROF_ARGS="--dtype fp32" bash tools/prof_step.sh ${tag}_fp32 || exit 1
bash tools/prof_cmd.sh ${tag}_cfg5 25 tools/cfg5_bench.py 32 20 bf16 || exit 1
timeout -k 10 200 python tools/conv_layers.py > gpurun_out/${tag}_conv_layers.txt 2>&1 || exit 1
CFG=5 timeout -k 10 200 python tools/conv_layers.py > gpurun_out/${tag}_conv_layers_cfg5.txt 2>&1 || exit 1
timeout -k 10 200 python tools/wgrad_bench.py > gpurun_out/${tag}_wgrad_layers.txt 2>&1 || exit 1
CFG=5 timeout -k 10 200 python tools/wgrad_bench.py 32 > gpurun_out/${tag}_wgrad_layers_cfg5.txt 2>&1 || exit 1
bash tools/prof_pmc.sh ${tag}_pmc_f FETCH_SIZE bench.py $B --steps 5 --warmup 3 || exit 1
bash tools/prof_pmc.sh ${tag}_pmc_w WRITE_SIZE bench.py $B --steps 5 --warmup 3 || exit 1
cd $root
python3 tools/pmc_traffic.py $(find gpurun_out/${tag}_pmc_f -name "*_results.db" | head -1) $(find gpurun_out/${tag}_pmc_w -name "*_results.db" | head -1) \
  conv_ws_kernel 8 gpurun_out/${tag}_conv_ws_traffic.json || exit 1
rm -rf gpurun_out/${tag}_pmc_f gpurun_out/${tag}_pmc_w
timeout -k 10 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || exit 1
SX_DP_FORCE=1 timeout -k 10 200 python bench.py $B > gpurun_out/${tag}_dp_force.json 2> gpurun_out/${tag}_dp_force.err || exit 1
echo evidence done
