#!/bin/bash
# A/B on one box: fused 128-channel BasicBlock on / off (HH_NO_BB128=1), alternating runs; prints value / forward_ms / decode_ms
for i in 1 2; do
  for v in 0 1 2; do
    if [ $v = 1 ]; then export HH_BB128=none; elif [ $v = 2 ]; then export HH_BB128=all; else unset HH_BB128; fi
    python bench.py --no-cpu-baseline --no-profile --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bb128 mode $v (0 stage-2 only, 1 none, 2 all)', d['value'], d['config']['forward_ms'], d['config']['decode_ms'])"
  done
done
