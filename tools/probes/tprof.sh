#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $out/tprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/tprof -- python3 $GRAFT_REPO_ROOT/tools/train_bench.py 32 5 > $out/tprof.log 2>&1
cd $GRAFT_REPO_ROOT
grep "train step" $out/tprof.log
python3 -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/tprof/**/*kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f))); tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel ms per step', tot/7e6)
for r in rows[:14]: print(r['Name'][:60], r['Calls'], round(float(r['TotalDurationNs'])/tot*100,1), round(float(r['AverageNs'])/1e3,1))
print('--- by calls per step')
for r in sorted(rows, key=lambda r: -int(r['Calls']))[:16]: print(r['Name'][:70], int(r['Calls'])//7, round(float(r['AverageNs'])/1e3,1))
print('launches per step', sum(int(r['Calls']) for r in rows)//7)
"
