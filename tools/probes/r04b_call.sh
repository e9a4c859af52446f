#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "decode or parse or end_to_end or chained or validation or infer_images or evaluate or native" > $out/r04j_test.log 2>&1
rc=$?
tail -5 $out/r04j_test.log
[ $rc -ne 0 ] && exit $rc
bash tools/probes/peaks_probe.sh "-DPEAKS_WPS=4" "-DPEAKS_WPS=3" > $out/r04j_probe.log 2>&1
cat $out/r04j_probe.log
timeout -k 10 100 python tools/decode_time.py > $out/r04j_decode.log 2>&1 && HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py >> $out/r04j_decode.log 2>&1
cat $out/r04j_decode.log
bash tools/probes/decode_kstats.sh
