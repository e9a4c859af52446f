#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "forward or fused or schedule or many_live or stale or multi_lane" > $out/r04o_test.log 2>&1
rc=$?
tail -4 $out/r04o_test.log
grep -q "Memory access fault" $out/r04o_test.log && exit 9
[ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  for e in "" "HH_NO_BB_TALL=1"; do
    r=$(env $e timeout -k 10 120 python bench.py --no-cpu-baseline --no-profile --steps 60 --warmup 10 --dense-people 0 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['forward_ms'], d['config']['decode_ms'], d['value'])")
    echo "${e:-default}: $r"
  done
done
timeout -k 10 100 python tools/bb_compare.py 2>/dev/null | tail -3
