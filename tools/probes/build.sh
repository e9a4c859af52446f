#!/bin/bash
# Builds the stand-alone kernel probes (run them on the GPU box: gpurun -- ./tools/probes/conv_probe 64 64 1).
#   conv_probe      phase stamps inside conv_mfma_kernel, start skew, busy / gap per back-to-back launch
#   nms_probe       phase stamps inside nms_tile_topk_kernel
#   dispatch_probe  how fast the chip starts the workgroups of an empty kernel
# The probes #include the product kernel sources with -DHH_CONV_DEBUG / -DHH_NMS_DEBUG; the product build has no stamps.
set -euo pipefail
cd "$(dirname "$0")"
csrc=../../pytorch-human-pose_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I$csrc conv_probe.hip -o conv_probe
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I$csrc nms_probe.hip -o nms_probe
hipcc --offload-arch=gfx950 -O3 dispatch_probe.hip -o dispatch_probe
echo built
