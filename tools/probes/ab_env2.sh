#!/bin/bash
# bench under several environment settings on one box: bash tools/probes/ab_env2.sh "A=1 B=2" "A=3" ...  ("-" = none)
set -eo pipefail
for i in 1 2; do
  for cfg in "$@"; do
    ( if [ "$cfg" != "-" ]; then export $cfg; fi
      timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile --steps 30 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$cfg]', d['value'], d['ms_per_step'], d['config'].get('forward_ms'), d['config'].get('decode_ms'))" )
  done
done
