#!/bin/bash
# A/B on one box: thin (half-CU, weights in registers) vs fat fused 32-channel block; multi-lane and single-lane
python -m pytest tests -m gpu -x -q -k "forward_with_taps or forward_outputs" 2>&1 | tail -1
HH_BB32=thin python -m pytest tests -m gpu -x -q -k "forward_with_taps or forward_outputs or full_size" 2>&1 | tail -1
for i in 1 2; do
  for v in fat thin; do
    for lane in "" "--single-lane"; do
      HH_BB32=$v python bench.py --no-cpu-baseline --no-profile --steps 60 $lane 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bb32=$v $lane', d['value'], d['config']['forward_ms'], d['config']['decode_ms'])"
    done
  done
done
