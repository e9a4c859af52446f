#!/bin/bash
# power / clock samples while the default bench loop runs (is the forward held by the board's power limit?): bash tools/probes/power_probe.sh
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^=\|^$" | head -20
timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 3000 --warmup 10 --dense-people 0 > gpurun_out/power_bench.log 2>&1 &
pid=$!
sleep 8
for i in $(seq 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | tr '\n' ' ' | sed 's/GPU\[0\]//g; s/  */ /g'; echo
  sleep 0.7
done
wait $pid
tail -1 gpurun_out/power_bench.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['config']['forward_ms'], d['config']['decode_ms'], d['value'])"
