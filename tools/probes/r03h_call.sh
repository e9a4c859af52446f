#!/bin/bash
# end-of-session check: full GPU suite, the bench lines, then the profile recipe
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/h1_test.log 2>&1
rc=$?; tail -6 gpurun_out/h1_test.log
[ $rc -ne 0 ] && exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/h1_smoke.log 2>&1; tail -2 gpurun_out/h1_smoke.log
timeout -k 10 300 python bench.py > gpurun_out/h1_bench.log 2>&1; tail -1 gpurun_out/h1_bench.log | cut -c1-700
timeout -k 10 300 python bench.py --chained --no-cpu-baseline --no-profile > gpurun_out/h1_chained.log 2>&1; tail -1 gpurun_out/h1_chained.log | cut -c1-200
bash tools/profile_round.sh r03 > gpurun_out/r03_profile_round.log 2>&1; tail -3 gpurun_out/r03_profile_round.log
