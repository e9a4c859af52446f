#!/bin/bash
# A/B on one box: producer/consumer vs tile form of the fused 32-channel block; parity tests under the switch, then bench (multi-lane and single-lane)
HH_BB32=pc python -m pytest tests -m gpu -x -q -k "forward or taps or fused or chained or full_size or infer" 2>&1 | tail -1
for i in 1 2 3; do
  for v in tile pc; do
    for lane in "" "--single-lane"; do
      HH_BB32=$v python bench.py --no-cpu-baseline --no-profile --steps 60 $lane 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bb32=$v $lane', d['value'], d['config']['forward_ms'], d['config']['decode_ms'])"
    done
  done
done
