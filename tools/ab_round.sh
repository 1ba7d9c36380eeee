#!/bin/bash
# on the GPU box: the round-2 final tree (_ab/r02, built by `git archive 5d1e169 | tar -x -C _ab/r02; make -C _ab/r02/unet_amd/csrc`)
# against this tree, same box, alternating: headline step, fp32 tier   -> gpurun_out/ab_round.txt
out=$GRAFT_REPO_ROOT/gpurun_out/ab_round.txt; : > $out
one() { ( cd $1 && timeout -k 10 200 python bench.py --no-infer --no-cpu-baseline --no-roofline $2 --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$3', d['ms_per_step'], (d.get('fp32_tier') or {}).get('ms_per_step'))" ) | tee -a $out; }
for i in 1 2 3; do
  one $GRAFT_REPO_ROOT/_ab/r02 "" r02 || exit 1
  one $GRAFT_REPO_ROOT "--no-extra" r03 || exit 1
done
