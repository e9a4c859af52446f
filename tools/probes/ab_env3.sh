#!/bin/bash
# A/B of engine switches on one box: bash tools/probes/ab_env3.sh "VAR=val" "VAR2=val" ...   ("-" = defaults)
for i in 1 2; do
  for v in "$@"; do
    ( if [ "$v" != "-" ]; then export $v; fi
      timeout -k 5 120 python bench.py --no-cpu-baseline --no-profile --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', d['value'], d['config']['forward_ms'], d['config']['decode_ms'])" )
  done
done
