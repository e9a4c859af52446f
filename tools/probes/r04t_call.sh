#!/bin/bash
# A/B of the fat kernels' CU budgets with the tall tile layout (alternating, two rounds)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for e in "" "HH_FAT_CUS=144,112" "HH_FAT_CUS=160,96" "HH_FAT_CUS=112,144" "HH_FAT_CUS=149,128"; do
    r=$(env $e timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 60 --warmup 10 --dense-people 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['config']['decode_ms'], d['value'])")
    echo "${e:-default}: $r"
  done
done
