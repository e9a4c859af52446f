cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "decode or parse or end_to_end or chained or validation or infer_images or evaluate or native or assignment" > gpurun_out/dec_tests.log 2>&1
rc=$?
tail -3 gpurun_out/dec_tests.log
grep -q "Memory access fault" gpurun_out/dec_tests.log && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python tools/decode_time.py 2>&1 | tail -1 && HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>&1 | tail -1 && bash tools/probes/decode_kstats.sh && bash tools/probes/decode_kstats.sh dense
