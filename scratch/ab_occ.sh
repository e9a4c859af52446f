#!/bin/bash
set -eo pipefail
export TMPDIR=/tmp
for v in 6 8; do
  out=$GRAFT_REPO_ROOT/gpurun_out
  cd /tmp; rm -rf $out/dprof$v
  HH_LIB=$GRAFT_REPO_ROOT/scratch/libhh_occ$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dprof$v -- python3 $GRAFT_REPO_ROOT/tools/decode_time.py > $out/dprof$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/dprof$v/**/*kernel_stats.csv',recursive=True))[-1]
for r in list(csv.DictReader(open(f)))[:9]:
    if 'refine_argmax' in r['Name']: print('occ $v', r['Name'][:30], round(float(r['AverageNs'])/1e3,1))
"
done
