#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "decode or parse or end_to_end or chained or validation or infer_images or evaluate or native or assignment" > $out/r04n_test.log 2>&1
rc=$?
tail -4 $out/r04n_test.log
grep -q "Memory access fault" $out/r04n_test.log && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python tools/decode_time.py 2>&1 | tail -1
HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>&1 | tail -1
bash tools/probes/decode_kstats.sh && bash tools/probes/decode_kstats.sh dense
timeout -k 10 300 python bench.py --steps 60 --warmup 10 > $out/r04n_bench.log 2>&1; tail -1 $out/r04n_bench.log
