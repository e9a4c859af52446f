#!/bin/bash
# builds tools/probes/peaks_probe.hip in several configurations on the GPU box and runs each: bash tools/probes/peaks_probe.sh "<flags>" ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DHH_PEAKS_DEBUG $cfg -Ipytorch-human-pose_amd/csrc tools/probes/peaks_probe.hip -o /tmp/peaks_probe 2>/dev/null || { echo "build failed: $cfg"; continue; }
  echo "== $cfg"
  timeout -k 5 60 /tmp/peaks_probe 10
  timeout -k 5 60 /tmp/peaks_probe 27 | head -1
done
