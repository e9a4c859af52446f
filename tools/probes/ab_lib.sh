#!/bin/bash
# alternating A/B of several builds of the library on one box: bash tools/probes/ab_lib.sh <rounds> <lib|-> ...   ("-" = the in-tree build)
cd $GRAFT_REPO_ROOT
n=$1; shift
args="--no-cpu-baseline --no-profile --steps 60 --warmup 10 --dense-people 0"
for i in $(seq $n); do
  for l in "$@"; do
    if [ "$l" = "-" ]; then e=""; else e="HH_LIB=$l"; fi
    bb=$(env $e timeout -k 10 100 python tools/bb_compare.py 2>/dev/null | tail -2 | sed -e 's/.*producer\/consumer \([0-9.]*\) us.*/\1/' | tr '\n' ' ')
    r=$(env $e timeout -k 10 120 python bench.py $args 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['config']['decode_ms'], d['value'])")
    printf "%-32s bbpc us (128^2 256^2): %s  bench: %s\n" "$l" "$bb" "$r"
  done
done
