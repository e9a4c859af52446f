#!/bin/bash
# alternating A/B of several environment settings on one box: bash tools/probes/ab_env.sh <rounds> "VAR=a" "VAR=b VAR2=c" ...   ("-" = no setting)
cd $GRAFT_REPO_ROOT
n=$1; shift
args="--no-cpu-baseline --no-profile --steps 60 --warmup 10 --dense-people 0"
for i in $(seq $n); do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    r=$(env $e timeout -k 10 120 python bench.py $args 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['config']['decode_ms'], d['value'])")
    printf "%-28s %s\n" "$v" "$r"
  done
done
