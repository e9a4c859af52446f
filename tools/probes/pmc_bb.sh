#!/bin/bash
# SQ counter passes over the two fused 32-channel block kernels (tools/bb_compare.py): bash tools/probes/pmc_bb.sh
set -eo pipefail
export TMPDIR=/tmp
out=gpurun_out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/bb_sq1 -- python3 tools/bb_compare.py > $out/bb_sq1.log 2>&1
echo "pass 1 done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $out/bb_sq2 -- python3 tools/bb_compare.py > $out/bb_sq2.log 2>&1
echo "pass 2 done"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $out/bb_sq3 -- python3 tools/bb_compare.py > $out/bb_sq3.log 2>&1
echo "pass 3 done"
