#!/bin/bash
# How much of the forward's wall time hangs on each kernel family: the forward is timed with that family's launches NOT issued
# (HH_DEBUG_SKIP, results wrong by construction) -- an upper bound on what a faster kernel of that family can buy end to end.
# bash tools/probes/skip_sensitivity.sh
cd $GRAFT_REPO_ROOT
for cat in none s2big s2 trans0 upadd c1x1 c256 c128 junc bb32 bb64 stem deconv head none; do
  if [ $cat = none ]; then unset HH_DEBUG_SKIP; else export HH_DEBUG_SKIP=$cat; fi
  ms=$(timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 40 --warmup 10 --dense-people 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['ms_per_step'])")
  echo "skip $cat: forward_ms ms_per_step = $ms"
done
