#!/bin/bash
# one GPU call: decode tests + decode A/B (coarse maxima on / off) + per-kernel decode times + a kernel trace of the forward
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "decode or parse or end_to_end or chained or validation or infer_images or evaluate or full_size or native" > gpurun_out/b1_test.log 2>&1
rc=$?; tail -8 gpurun_out/b1_test.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  echo "no coarse: $(HH_DECODE_NO_COARSE=1 timeout -k 10 100 python tools/decode_time.py | tail -1)   dense $(HH_DECODE_NO_COARSE=1 HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py | tail -1)"
  echo "default  : $(timeout -k 10 100 python tools/decode_time.py | tail -1)   dense $(HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py | tail -1)"
done 2>&1 | tee gpurun_out/b1_decode_ab.log
bash tools/probes/decode_kstats.sh 2>&1 | tee gpurun_out/b1_kstats.log
bash tools/probes/decode_kstats.sh dense 2>&1 | tee gpurun_out/b1_kstats_dense.log
bash tools/probes/timeline.sh 2>&1 | tee gpurun_out/b1_timeline.log
python3 tools/probes/module_gantt.py $(ls -S gpurun_out/tl/*/*kernel_trace.csv | head -1) > gpurun_out/b1_gantt.txt 2>&1
tail -3 gpurun_out/b1_gantt.txt
