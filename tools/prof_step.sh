#!/bin/bash
# usage (on the GPU box): tools/prof_step.sh <tag>   -> gpurun_out/<tag>_kernels.txt, gpurun_out/<tag>_families.txt
# kernel-trace profile of `bench.py --no-infer --steps 20 --warmup 5` (25 training steps in the trace)
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/$tag
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/$tag -- python3 $root/bench.py --no-infer --no-cpu-baseline --no-roofline --no-fp32 --no-extra --steps 20 --warmup 5 $PROF_ARGS > $root/gpurun_out/$tag.log 2>&1 || exit 1
db=$(find $root/gpurun_out/$tag -name "*_results.db" | head -1)
python3 $root/tools/prof_summary.py $db 25 > $root/gpurun_out/${tag}_kernels.txt
python3 $root/tools/prof_cats.py $db 25 > $root/gpurun_out/${tag}_families.txt
n=$(python3 -c "import sqlite3,sys; print(sqlite3.connect(sys.argv[1]).execute('select count(*) from kernels').fetchone()[0])" $db)
per=$((n / 25))
python3 $root/tools/prof_seq.py $db $((n - per - 2)) $((per + 4)) > $root/gpurun_out/${tag}_sequence.txt
python3 $root/tools/elementwise_roofline.py $db > $root/gpurun_out/${tag}_elementwise.txt 2>&1
rm -rf $root/gpurun_out/$tag
