#!/bin/bash
# A/B of the decode kernels: in-tree build vs tools/probes/libhh_$1.so  (bash tools/probes/ab_dec.sh u16)
set -eo pipefail
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
for v in base $1; do
  if [ $v = base ]; then unset HH_LIB; else export HH_LIB=$GRAFT_REPO_ROOT/tools/probes/libhh_$v.so; fi
  cd /tmp; rm -rf $out/dprof_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/dprof_$v -- python3 $GRAFT_REPO_ROOT/tools/decode_time.py > $out/dprof_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  python3 -c "
import csv,glob
f=sorted(glob.glob('gpurun_out/dprof_$v/**/*kernel_stats.csv',recursive=True))[-1]
print('$v', ' '.join(r['Name'][:14]+'='+str(round(float(r['AverageNs'])/1e3,1)) for r in list(csv.DictReader(open(f)))[:7]))
"
done
