#!/bin/bash
# bench with several library builds on one box: bash tools/probes/ab_libs.sh base deep resepi   (base = in-tree)
set -eo pipefail
for i in 1 2 3; do
  for v in "$@"; do
    ( if [ $v != base ]; then export HH_LIB=$GRAFT_REPO_ROOT/tools/probes/libhh_$v.so; fi
      timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile --steps 30 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['config'].get('forward_ms'), d['config'].get('decode_ms'))" )
  done
done
