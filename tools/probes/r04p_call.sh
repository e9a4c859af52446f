#!/bin/bash
# round-4 profile recipe: kernel-trace / stats passes, FETCH / WRITE passes, SQ pass, summary
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r04 && bash tools/pmc_pass.sh r04 && python3 tools/summarise_profile.py r04 | tail -5
cp gpurun_out/r04m_gemm_control.log profiles/r04_gemm_control.txt 2>/dev/null
mkdir -p gpurun_out/r04_profiles && cp profiles/r04_* profiles/traffic_r04.json gpurun_out/r04_profiles/ 2>/dev/null
ls gpurun_out/r04_profiles
