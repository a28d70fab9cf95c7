#!/bin/bash
# On the GPU box: rebuild the in-tree library with extra compiler flags (the box's copy is
# scratch).  Default -DRFM_ABLATE: RFM_ABLATE_MASK then switches parts of the forward kernel off
# for timing experiments.   usage: profiles/ablate_build.sh [flags...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
C=$R/relevance_factorizationmachine_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared ${@:--DRFM_ABLATE} \
  -o $R/relevance_factorizationmachine_amd/librfm_hip.so \
  $C/rfm_capi.hip $C/rfm_fm.hip $C/rfm_fm_plan.hip $C/rfm_mf.hip $C/rfm_eval.hip $C/rfm_csr.hip \
  $C/rfm_host.cpp $C/rfm_comm.cpp -lpthread -ldl
