#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
K="decode or parse or end_to_end or chained or validation or infer_images or evaluate or native or full_size or assignment"
timeout -k 10 800 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "$K" > gpurun_out/i1_test.log 2>&1
rc=$?; tail -6 gpurun_out/i1_test.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  echo "per tile : $(HH_NMS_NO_GROUP=1 timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)   dense $(HH_NMS_NO_GROUP=1 HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)"
  echo "2x2 group: $(timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)   dense $(HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>/dev/null | tail -1)"
done | tee gpurun_out/i1_decode.log
bash tools/probes/decode_kstats.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/i1_kstats.log
bash tools/probes/decode_kstats.sh dense 2>&1 | grep -v amdgpu.ids | tee gpurun_out/i1_kstats_dense.log
