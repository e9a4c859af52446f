#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $out/dprof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dprof -- python3 $GRAFT_REPO_ROOT/tools/decode_time.py > $out/dprof.log 2>&1
cd $GRAFT_REPO_ROOT
grep decode $out/dprof.log
python3 -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/dprof/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:12]: print(r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
