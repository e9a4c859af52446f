#!/bin/bash
# A/B of two builds on one box: bash tools/probes/ab_bench.sh  (prev = tools/probes/libhh_prev.so, new = in-tree)
set -eo pipefail
for i in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then export HH_LIB=$GRAFT_REPO_ROOT/tools/probes/libhh_prev.so; else unset HH_LIB; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['config'].get('forward_ms'), d['config'].get('decode_ms'))"
  done
done
