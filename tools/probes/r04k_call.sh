#!/bin/bash
# tests on the new refine + decode timings for two builds of refine_bb_kernel (3 workgroups per CU in-tree, 4 with spills in scratch/lib4)
cd $GRAFT_REPO_ROOT
out=gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "decode or parse or end_to_end or chained or validation or infer_images or evaluate or native or assignment" > $out/r04m_test.log 2>&1
rc=$?
tail -4 $out/r04m_test.log
grep -q "Memory access fault" $out/r04m_test.log && exit 9
[ $rc -ne 0 ] && exit $rc
for lib in "" scratch/lib256/libhhrnet.so; do
  echo "== lib: ${lib:-in-tree}"
  HH_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout -k 10 100 python tools/decode_time.py 2>&1 | tail -1
  HH_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} HH_DECODE_PEOPLE=27 timeout -k 10 100 python tools/decode_time.py 2>&1 | tail -1
done
bash tools/probes/decode_kstats.sh && bash tools/probes/decode_kstats.sh dense
timeout -k 10 150 python tools/probes/gemm_control.py 4 > $out/r04m_gemm_control.log 2>&1; cat $out/r04m_gemm_control.log | grep -v amdgpu.ids
