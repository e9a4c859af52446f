#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== old lib"; HH_LIB=$GRAFT_REPO_ROOT/scratch/libhhrnet_before_s2.so timeout -k 10 200 python tools/probes/forward_hash.py 2>&1 | grep "^W"
echo "== new lib"; timeout -k 10 200 python tools/probes/forward_hash.py 2>&1 | grep "^W"
for i in 1 2 3; do
  for l in "" scratch/libhhrnet_before_s2.so; do
    r=$(HH_LIB=${l:+$GRAFT_REPO_ROOT/$l} timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 60 --warmup 10 --dense-people 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['config']['decode_ms'], d['value'])")
    echo "${l:-new}: $r"
  done
done
