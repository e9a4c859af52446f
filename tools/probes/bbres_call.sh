#!/bin/bash
# residual-by-MFMA + late-finish consumer (in-tree) against the build before it (csrc/variants/libhh_head.so): tests first, then alternating A/B
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "final_layer or switches or golden or forward or head or stale_lds or workspace" > gpurun_out/bbres_tests.log 2>&1
rc=$?
tail -5 gpurun_out/bbres_tests.log
grep -q "Memory access fault" gpurun_out/bbres_tests.log && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 100 python tools/probes/forward_hash.py && HH_LIB=$GRAFT_REPO_ROOT/pytorch-human-pose_amd/csrc/variants/libhh_head.so timeout -k 10 100 python tools/probes/forward_hash.py && bash tools/probes/ab_lib.sh 3 - $GRAFT_REPO_ROOT/pytorch-human-pose_amd/csrc/variants/libhh_head.so
