#!/bin/bash
# N full `pytest -m gpu` runs back to back on one box, stopping at the first failure with its text: bash tools/probes/soak_full_suite.sh [N=5]
cd $GRAFT_REPO_ROOT
n=${1:-5}
mkdir -p gpurun_out
for i in $(seq $n); do
  timeout -k 10 400 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/soak_suite_$i.log 2>&1
  rc=$?
  echo "run $i: rc $rc: $(tail -1 gpurun_out/soak_suite_$i.log)"
  if [ $rc -ne 0 ]; then grep -E "^E |Error|FAILED" gpurun_out/soak_suite_$i.log | head -20; exit $rc; fi
done
timeout -k 10 300 python tools/probes/soak_step.py 2>&1 | tail -3
