#!/bin/bash
# stage-0 conv2 (3x3, 64 -> 64 at 128^2) on the double-buffered form: golden tests, per-kernel times, A/B
cd $GRAFT_REPO_ROOT
HH_CONV_DB_MIN_CIN=64 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "golden or taps" > gpurun_out/db64_tests.log 2>&1
rc=$?
tail -3 gpurun_out/db64_tests.log
[ $rc -ne 0 ] && exit $rc
HH_CONV_DB_MIN_CIN=64 bash tools/probes/kstats.sh "conv_mfma_kernel<3, 1" && bash tools/probes/kstats.sh "conv_mfma_kernel<3, 1" && bash tools/probes/ab_env.sh 3 "-" "HH_CONV_DB_MIN_CIN=64"
