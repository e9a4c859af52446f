#!/bin/bash
# A/B on one box: fused 64-channel BasicBlock on / off (HH_NO_BB64=1), alternating runs
for i in 1 2; do
  for v in 0 1; do
    if [ $v = 1 ]; then export HH_NO_BB64=1; else unset HH_NO_BB64; fi
    python bench.py --no-cpu-baseline --no-profile --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('no_bb64=$v', d['value'], d['config']['forward_ms'], d['config']['decode_ms'])"
  done
done
