#!/bin/bash
# round 4, call a: decode tests on the new front end, then decode timings (bench maps and dense maps) and per-kernel stats
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decode or parse or end_to_end or chained or validation or infer_images or evaluate or native" > $out/r04a_test.log 2>&1
rc=$?
tail -5 $out/r04a_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python tools/decode_time.py > $out/r04a_decode.log 2>&1 && HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py >> $out/r04a_decode.log 2>&1
cat $out/r04a_decode.log
bash tools/probes/decode_kstats.sh && bash tools/probes/decode_kstats.sh dense
