#!/bin/bash
# kernel trace of the default (multi-lane) bench and the timeline statistics of one forward: bash tools/probes/timeline.sh
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $out/tl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/tl -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile --steps 40 --warmup 5 > $out/tl.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/probes/timeline.py $(ls -S gpurun_out/tl/*/*kernel_trace.csv | head -1)
