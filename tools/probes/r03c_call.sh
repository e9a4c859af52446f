#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
K="forward or switches or lane or head_folded or classification or native or chained or full_size or decode or parse or end_to_end or validation or infer_images or evaluate"
timeout -k 10 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "$K" > gpurun_out/c1_test.log 2>&1
rc=$?; tail -8 gpurun_out/c1_test.log
[ $rc -ne 0 ] && exit $rc
HH_S2_KC32=32 timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "forward_outputs or forward_with_taps or full_size or classification" > gpurun_out/c1_test_s2.log 2>&1
rc=$?; tail -8 gpurun_out/c1_test_s2.log
[ $rc -ne 0 ] && exit $rc
bash tools/probes/ab_env.sh 3 - "HH_BB64=sb" "HH_S2_KC32=256" "HH_S2_KC32=64" "HH_S2_KC32=32" 2>&1 | tee gpurun_out/c1_ab.log
for i in 1 2 3; do
  echo "no coarse: $(HH_DECODE_NO_COARSE=1 timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)   dense $(HH_DECODE_NO_COARSE=1 HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>/dev/null| tail -1)"
  echo "default  : $(timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)   dense $(HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)"
done 2>&1 | tee gpurun_out/c1_decode_ab.log
bash tools/probes/decode_kstats.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c1_kstats.log
timeout -k 10 200 python bench.py --no-cpu-baseline > gpurun_out/c1_bench.log 2>&1; tail -1 gpurun_out/c1_bench.log | cut -c1-400
