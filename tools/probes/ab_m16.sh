#!/bin/bash
# conv3x3_m16 experiment: parity of the switch, then alternating A/B against the default (DB) and the round-2 (KC = 32) instantiations,
# then one profiled bench line per variant (per-kernel averages in config.kernels)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule_and_fusion" > gpurun_out/m16_test.log 2>&1 || { tail -30 gpurun_out/m16_test.log; exit 1; }
tail -2 gpurun_out/m16_test.log
bash tools/probes/ab_env.sh 3 - HH_CONV_M16=1 HH_NO_CONV_DB=1 | tee gpurun_out/m16_ab.log
for v in - HH_CONV_M16=1 HH_NO_CONV_DB=1; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env HH_BENCH_ALL_KERNELS=1 $e timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --dense-people 0 2>/dev/null | tail -1 > gpurun_out/m16_bench_${v//[^A-Z0-9]/}.json
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/m16_bench_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['config'].get('forward_ms'))
    r=d['roofline']; print('   ', r['kernel'][:60], r['launches'], r['avg_launch_us'], r['frac'])
    for k in r['runners_up']: print('   ', k['kernel'][:60], k['launches'], k['avg_launch_us'], k['frac'])
PY
