#!/bin/bash
# per-kernel times of hh_decode alone (bench maps, 10 people/image; arg "dense" = 27 people/image): bash tools/probes/decode_kstats.sh [dense]
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
tag=decode_kstats${1:+_$1}
cd /tmp
rm -rf $out/$tag
HH_DECODE_PEOPLE=${1:+27} timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- python3 $GRAFT_REPO_ROOT/tools/decode_time.py > $out/$tag.log 2>&1
cd $GRAFT_REPO_ROOT
tail -1 $out/$tag.log
python3 -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/$tag/**/*kernel_stats.csv',recursive=True))[-1]
tot=0
for r in list(csv.DictReader(open(f)))[:16]:
    print(r['Name'][:44].ljust(44), r['Calls'].rjust(5), str(round(float(r['AverageNs'])/1e3,1)).rjust(8))
"
