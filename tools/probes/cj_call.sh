#!/bin/bash
# lane synchronisation variants of enqueue(): bits, then alternating A/B (wait elision, event fences)
cd $GRAFT_REPO_ROOT
HH_KEEP_WAITS=1 timeout -k 10 100 python tools/probes/forward_hash.py && HH_EVENT_SYSTEM_FENCE=1 timeout -k 10 100 python tools/probes/forward_hash.py && timeout -k 10 100 python tools/probes/forward_hash.py && bash tools/probes/ab_env.sh 4 "-" "HH_KEEP_WAITS=1" "HH_EVENT_SYSTEM_FENCE=1"
