#!/bin/bash
set -eo pipefail
for v in base s2occ3 s2occ4; do
  if [ $v = base ]; then unset HH_LIB; else export HH_LIB=$GRAFT_REPO_ROOT/scratch/libhh_$v.so; fi
  timeout -k 10 200 python tools/layer_profile.py > gpurun_out/lp_$v.log 2>&1
  echo "== $v"; grep -E "conv total|cfg\(3, 2" gpurun_out/lp_$v.log | cut -c1-150
done
