#!/bin/bash
# One SQ counter pass over the serial bench (run on the GPU box): bash tools/pmc_pass.sh r01
set -eo pipefail
tag=${1:-r01}
export TMPDIR=/tmp
out=gpurun_out
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_sq -- python3 bench.py --no-cpu-baseline --single-lane --sequential --steps 3 --warmup 1 > $out/${tag}_sq.log 2>&1
echo "sq pass done"
