#!/bin/bash
# A/B of two builds, forward only (graph single-lane and eager multi-lane), 4 alternations on one box:
#   prev = tools/probes/libhh_prev.so, new = the in-tree library
set -eo pipefail
for i in 1 2 3 4; do
  for v in prev new; do
    if [ $v = prev ]; then export HH_LIB=$GRAFT_REPO_ROOT/tools/probes/libhh_prev.so; else unset HH_LIB; fi
    echo "$v $(timeout -k 10 200 python tools/time_forward.py 32 '((True, 0), (False, 1))' 2>&1 | grep 'B=32' | awk '{print $4}' | tr '\n' ' ')"
  done
done
