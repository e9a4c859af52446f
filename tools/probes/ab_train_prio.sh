#!/bin/bash
set -eo pipefail
for i in 1 2; do
  for v in 0 -1; do
    echo "priority $v: $(HH_STREAM_PRIORITY=$v timeout -k 10 300 python bench.py --train --steps 5 --warmup 2 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
  done
done
