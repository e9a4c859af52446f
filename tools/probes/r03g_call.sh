#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "training or train or wgrad or batchnorm or fusion_sum or native" > gpurun_out/g1_test.log 2>&1
rc=$?; tail -6 gpurun_out/g1_test.log
[ $rc -ne 0 ] && exit $rc
tr() { timeout -k 10 300 python bench.py --train --steps 20 --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for i in 1 2; do
  echo "autograd adds: $(HH_TRAIN_NO_RESBOX=1 tr)"
  echo "skip boxes   : $(tr)"
done | tee gpurun_out/g1_train.log
