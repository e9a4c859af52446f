#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/api_throughput.py 512 2>&1 | grep -v amdgpu.ids | tail -6
API_PHASES=1 timeout -k 10 200 python tools/api_throughput.py 512 2>&1 | grep -v amdgpu.ids | tail -12
