#!/bin/bash
# One GPU-box call: tests, bench line, serial rocprofv3 pass (usage: bash tools/gpu_round.sh <tag> [pytest -k expr])
tag=${1:-r02}
kexpr=${2:-}
export TMPDIR=/tmp
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$kexpr" > gpurun_out/${tag}_test.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_test.log 2>&1
fi
rc=$?
tail -5 gpurun_out/${tag}_test.log
if [ $rc -ge 124 ]; then echo "tests timed out / killed: stopping"; exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/${tag}_bench.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/${tag}_bench.log; exit 1; }
tail -1 gpurun_out/${tag}_bench.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_serial -- python3 bench.py --no-cpu-baseline --single-lane > gpurun_out/${tag}_serial.log 2>&1 || { echo "serial rocprof failed"; tail -5 gpurun_out/${tag}_serial.log; exit 1; }
tail -1 gpurun_out/${tag}_serial.log
f=$(ls gpurun_out/${tag}_serial/*/*kernel_stats.csv | head -1)
head -12 $f
