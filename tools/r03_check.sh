#!/bin/bash
# One GPU-box call of round 3: tests, fp8 accuracy table, bench lines (usage: bash tools/r03_check.sh <tag> [pytest -k expr])
tag=${1:-r03}
kexpr=${2:-}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "$kexpr" > gpurun_out/${tag}_test.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/${tag}_test.log 2>&1
fi
rc=$?
tail -25 gpurun_out/${tag}_test.log
if [ $rc -ge 124 ]; then echo "tests timed out / killed: stopping"; exit $rc; fi
timeout -k 10 200 python tools/fp8_check.py 32 > gpurun_out/${tag}_fp8_w32.log 2>&1 && grep -E "==|OUT|stem#0|stages.3 " gpurun_out/${tag}_fp8_w32.log
timeout -k 10 200 python tools/fp8_check.py 48 > gpurun_out/${tag}_fp8_w48.log 2>&1 && grep -E "==|OUT" gpurun_out/${tag}_fp8_w48.log
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/${tag}_bench.log; exit 1; }
tail -1 gpurun_out/${tag}_bench.log | cut -c1-1500
timeout -k 10 300 python bench.py --config fp8_w48_b64_640 --no-cpu-baseline --steps 30 > gpurun_out/${tag}_bench_fp8.log 2>&1 || { echo "fp8 bench failed"; tail -5 gpurun_out/${tag}_bench_fp8.log; exit 1; }
tail -1 gpurun_out/${tag}_bench_fp8.log | cut -c1-1200
exit $rc
