#!/bin/bash
# junction kernels after a change: golden / switch tests, bits, per-kernel times (single-lane bench under rocprofv3)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "golden or switches or taps or classif" > gpurun_out/junc_tests.log 2>&1
rc=$?
tail -3 gpurun_out/junc_tests.log
grep -q "Memory access fault" gpurun_out/junc_tests.log && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python tools/probes/forward_hash.py && bash tools/probes/kstats.sh junction
