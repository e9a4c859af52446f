#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
K="decode or parse or end_to_end or chained or validation or infer_images or evaluate or fp8 or native or full_size"
timeout -k 10 800 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "$K" > gpurun_out/d1_test.log 2>&1
rc=$?; tail -8 gpurun_out/d1_test.log
[ $rc -ne 0 ] && exit $rc
bash tools/probes/power_probe.sh 2>&1 | tee gpurun_out/d1_power.log
bash tools/probes/ab_env.sh 3 - "HH_CONV_DB_MIN=64" 2>&1 | tee gpurun_out/d1_ab.log
fp8() { timeout -k 10 300 python bench.py --config fp8_w48_b64_640 --no-cpu-baseline --no-profile --steps 30 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['value'])"; }
for i in 1 2; do
  echo "fp8 base lib: $(HH_LIB=$PWD/scratch/libhhrnet_base.so fp8)"
  echo "fp8 this lib: $(fp8)"
done 2>&1 | tee gpurun_out/d1_fp8.log
for i in 1 2 3; do echo "decode: $(timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)   dense $(HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)"; done | tee gpurun_out/d1_decode.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/d1_bench.log 2>&1; tail -1 gpurun_out/d1_bench.log | cut -c1-300
