#!/bin/bash
# alternating A/B of one engine switch on one box: bash tools/probes/ab_switch.sh "HH_NO_CONV_DB=1" [rounds]
cd $GRAFT_REPO_ROOT
sw="$1"; n=${2:-3}
run() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 60 --warmup 10 --dense-people 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['value'])"; }
for i in $(seq $n); do
  echo "default: $(run)"
  echo "$sw: $(env $sw python bench.py --no-cpu-baseline --no-profile --steps 60 --warmup 10 --dense-people 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['value'])")"
done
