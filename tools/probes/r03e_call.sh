#!/bin/bash
# full GPU suite + the bench lines of the round's configurations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/e1_test.log 2>&1
rc=$?; tail -6 gpurun_out/e1_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/e1_bench.log 2>&1; tail -1 gpurun_out/e1_bench.log | cut -c1-600
timeout -k 10 300 python bench.py --train --steps 20 --warmup 5 > gpurun_out/e1_train.log 2>&1; tail -1 gpurun_out/e1_train.log | cut -c1-400
timeout -k 10 300 python bench.py --config fp8_w48_b64_640 --no-cpu-baseline --steps 30 > gpurun_out/e1_fp8.log 2>&1; tail -1 gpurun_out/e1_fp8.log | cut -c1-300
timeout -k 10 300 python bench.py --config bf16_w48_b64_640 --no-cpu-baseline --steps 30 > gpurun_out/e1_w48.log 2>&1; tail -1 gpurun_out/e1_w48.log | cut -c1-300
timeout -k 10 300 python bench.py --chained --no-cpu-baseline --no-profile > gpurun_out/e1_chained.log 2>&1; tail -1 gpurun_out/e1_chained.log | cut -c1-300
