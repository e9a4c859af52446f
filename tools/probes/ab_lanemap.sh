#!/bin/bash
# A/B on one box: which resolution branches share a stream (HH_LANE_MAP digit l = stream of lane l; default 0123)
for i in 1 2; do
  for v in 0123 0012 0011 0122 0112 0101; do
    HH_LANE_MAP=$v timeout -k 5 120 python bench.py --no-cpu-baseline --no-profile --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('map=$v', d['value'], d['config']['forward_ms'], d['config']['decode_ms'])"
  done
done
