#!/bin/bash
# kernel-trace stats of the batched API path (tools/api_throughput.py): bash tools/probes/api_kstats.sh
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $out/api_kstats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/api_kstats -- python3 $GRAFT_REPO_ROOT/tools/api_throughput.py 512 > $out/api_kstats.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 $out/api_kstats.log
python3 -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/api_kstats/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), round(float(r['TotalDurationNs'])/1e6,1), r['Percentage'])
"
