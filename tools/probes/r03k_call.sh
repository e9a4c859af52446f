#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== bn probe, previous build"; HH_LIB=scratch/libhhrnet_prev.so timeout -k 10 200 python tools/probes/bn_probe.py 2>&1 | tee gpurun_out/bn_probe_prev.log
echo "== bn probe, this build"; timeout -k 10 200 python tools/probes/bn_probe.py 2>&1 | tee gpurun_out/bn_probe_new.log
echo "== m16 (maps wider than 16 only)"
for v in - HH_CONV_M16=1; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env HH_BENCH_ALL_KERNELS=1 $e timeout -k 10 200 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --dense-people 0 2>/dev/null | tail -1 > gpurun_out/m16b_bench_${v//[^A-Z0-9]/}.json
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/m16b_bench_*.json')):
    d=json.load(open(f)); print(f, d['value'], d['config'].get('forward_ms'))
    r=d['roofline']
    for k in [r]+r['runners_up']:
        if 'KS=3,S=1' in k['kernel'] or 'm16' in k['kernel']: print('   ', k['kernel'][:60], k['launches'], k['avg_launch_us'], k['frac'])
PY
