#!/bin/bash
# usage: tools/prof_pmc.sh <outdir-name> "<counters>" <python script> [args...]   (GPU box; one --pmc pass)
name=$1; shift
ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctr -d $GRAFT_REPO_ROOT/gpurun_out/$name -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/$name.log 2>&1
