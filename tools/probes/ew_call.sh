#!/bin/bash
# cross-stream edge costs, then the early-wait plan against the default (alternating), then its bits
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/probes/event_cost.py && bash tools/probes/ab_env.sh 3 "-" "HH_EARLY_WAIT=1" && HH_EARLY_WAIT=1 timeout -k 10 100 python tools/probes/forward_hash.py && timeout -k 10 100 python tools/probes/forward_hash.py
