#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== stamps: pc (first) vs sw (second); the printed stamps are the LAST kernel's (sw)"
HH_LIB=$GRAFT_REPO_ROOT/scratch/libstamp/libhhrnet.so HH_BB_CMP=sw timeout -k 10 120 python tools/bb_compare.py 2>&1 | tail -24
