#!/bin/bash
# per-kernel times of the training step: bash tools/probes/train_kstats.sh
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $out/train_kstats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/train_kstats -- python3 $GRAFT_REPO_ROOT/bench.py --train --steps 6 --warmup 2 > $out/train_kstats.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $out/train_kstats.log | cut -c1-200
python3 -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/train_kstats/**/*kernel_stats.csv',recursive=True))[-1]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', round(tot/1e6/8,2), '(8 steps traced incl. warmup)')
for r in rows[:40]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), str(round(float(r['AverageNs'])/1e3,1)).rjust(8), str(round(100*float(r['TotalDurationNs'])/tot,1)).rjust(6))
"
