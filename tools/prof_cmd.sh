#!/bin/bash
# usage (on the GPU box): tools/prof_cmd.sh <tag> <steps-in-trace> <script.py> [args...]
#   -> gpurun_out/<tag>_kernels.txt, <tag>_families.txt, <tag>_sequence.txt
# kernel-trace profile of any python script of this tree whose trace holds <steps-in-trace> equal steps
tag=$1; steps=$2; shift 2
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $root/gpurun_out/$tag
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/$tag -- python3 $root/"$1" "${@:2}" > $root/gpurun_out/$tag.log 2>&1 || exit 1
db=$(find $root/gpurun_out/$tag -name "*_results.db" | head -1)
python3 $root/tools/prof_summary.py $db $steps > $root/gpurun_out/${tag}_kernels.txt
python3 $root/tools/prof_cats.py $db $steps > $root/gpurun_out/${tag}_families.txt
n=$(python3 -c "import sqlite3,sys; print(sqlite3.connect(sys.argv[1]).execute('select count(*) from kernels').fetchone()[0])" $db)
per=$((n / steps))
python3 $root/tools/prof_seq.py $db $((n - per - 2)) $((per + 4)) > $root/gpurun_out/${tag}_sequence.txt
rm -rf $root/gpurun_out/$tag
