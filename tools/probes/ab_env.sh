#!/bin/bash
# A/B of an environment switch on one box: bash tools/probes/ab_env.sh VAR=value   (bench without cpu baseline / probe step)
set -eo pipefail
for i in 1 2; do
  for v in off on; do
    if [ $v = on ]; then export "$1"; else unset "${1%%=*}"; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile --steps 30 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['config'].get('forward_ms'), d['config'].get('decode_ms'))"
  done
done
