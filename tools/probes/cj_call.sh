#!/bin/bash
# redundant-wait elision in enqueue() (default) against HH_KEEP_WAITS=1: bits, then alternating A/B
cd $GRAFT_REPO_ROOT
HH_KEEP_WAITS=1 timeout -k 10 100 python tools/probes/forward_hash.py && timeout -k 10 100 python tools/probes/forward_hash.py && bash tools/probes/ab_env.sh 4 "-" "HH_KEEP_WAITS=1"
