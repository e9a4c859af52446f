#!/bin/bash
# the head-in-the-epilogue fusion on a GPU box: its test + the switch / golden tests, then an alternating A/B against HH_NO_FINAL_FUSE=1
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "final_layer or switches or golden or forward or head" > gpurun_out/fin_tests.log 2>&1
rc=$?
tail -5 gpurun_out/fin_tests.log
grep -q "Memory access fault" gpurun_out/fin_tests.log && exit 9
[ $rc -ne 0 ] && exit $rc
bash tools/probes/ab_env.sh 3 "-" "HH_NO_FINAL_FUSE=1"
