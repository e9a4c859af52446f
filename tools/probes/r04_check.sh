#!/bin/bash
# round-4 check on a GPU box: full GPU suite, then decode timings / per-kernel stats and one bench line.  bash tools/probes/r04_check.sh
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 1100 python -m pytest tests/ -x -q -s -m gpu > $out/r04_check_tests.log 2>&1
rc=$?
tail -4 $out/r04_check_tests.log
grep -q "Memory access fault" $out/r04_check_tests.log && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python tools/decode_time.py 2>&1 | tail -1
HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>&1 | tail -1
bash tools/probes/decode_kstats.sh && bash tools/probes/decode_kstats.sh dense
timeout -k 10 400 python bench.py > $out/r04_check_bench.log 2>&1; tail -1 $out/r04_check_bench.log
