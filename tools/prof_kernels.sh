#!/bin/bash
# usage: tools/prof_kernels.sh <outdir-name> <python script> [args...]   (run on the GPU box; env passes through)
# kernel-trace profile of one python tool; prints nothing, leaves gpurun_out/<name>/.../*_results.db
name=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
