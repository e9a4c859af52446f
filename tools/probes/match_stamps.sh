#!/bin/bash
# build here:  (cd pytorch-human-pose_amd/csrc && mkdir -p variants && hipcc -DHH_MATCH_STAMP -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c decode_kernels.hip -o /tmp/dk.o &&
#               hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libhh_mstamp.so $(ls *.o | grep -v decode_kernels.o) /tmp/dk.o)
# run on the GPU box:  bash tools/probes/match_stamps.sh
cd $GRAFT_REPO_ROOT
HH_LIB=$GRAFT_REPO_ROOT/pytorch-human-pose_amd/csrc/variants/libhh_mstamp.so timeout -k 10 120 python tools/probes/match_stamps.py
