#!/bin/bash
# A/B of the double-buffered 128- / 256-channel 3x3 convs on one box: old build (scratch/libhhrnet_base.so, optional), this build with
# HH_NO_CONV_DB=1 (only the branch-free staging), this build as is.  usage: bash tools/probes/ab_convdb.sh [rounds]
cd $GRAFT_REPO_ROOT
n=${1:-3}
line() { tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['value'])"; }
args="--no-cpu-baseline --no-profile --steps 60 --warmup 10 --dense-people 0"
for i in $(seq $n); do
  [ -f scratch/libhhrnet_base.so ] && echo "base lib : $(HH_LIB=$PWD/scratch/libhhrnet_base.so timeout -k 10 120 python bench.py $args 2>/dev/null | line)"
  echo "no conv db: $(HH_NO_CONV_DB=1 timeout -k 10 120 python bench.py $args 2>/dev/null | line)"
  echo "default   : $(timeout -k 10 120 python bench.py $args 2>/dev/null | line)"
done
