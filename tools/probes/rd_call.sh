#!/bin/bash
# read-ahead depth of the fused block's fragment reads (RD producer, RDC consumer): kernel bench + forward per build, alternating
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/pytorch-human-pose_amd/csrc/variants
bash tools/probes/ab_lib.sh 2 - $V/libhh_rd2_6.so $V/libhh_rd3_4.so $V/libhh_rd3_6.so $V/libhh_rd2_8.so
