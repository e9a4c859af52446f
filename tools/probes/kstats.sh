#!/bin/bash
# kernel-trace stats of a short single-lane bench: bash tools/probes/kstats.sh [pattern]
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $out/kstats
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kstats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-profile --single-lane --steps 10 --warmup 3 > $out/kstats.log 2>&1
cd $GRAFT_REPO_ROOT
python3 -c "
import csv,glob,sys
f=sorted(glob.glob('gpurun_out/kstats/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:40]:
    if '$1' in r['Name']: print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1))
"
