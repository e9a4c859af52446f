#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r04q_full.log 2>&1
rc=$?
tail -6 gpurun_out/r04q_full.log
grep -q "Memory access fault" gpurun_out/r04q_full.log && exit 9
exit $rc
