#!/bin/bash
# soak: the full GPU suite three times back to back, then the API and latency figures
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/j1_test_$i.log 2>&1
  rc=$?; tail -2 gpurun_out/j1_test_$i.log
  [ $rc -ne 0 ] && exit $rc
done
timeout -k 10 200 python tools/api_throughput.py 512 > gpurun_out/j1_api.log 2>&1; tail -4 gpurun_out/j1_api.log
timeout -k 10 200 python tools/latency.py > gpurun_out/j1_latency.log 2>&1; tail -6 gpurun_out/j1_latency.log
