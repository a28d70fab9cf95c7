#!/bin/bash
# On the GPU box: rebuild the in-tree library with extra compiler flags (the box's copy is
# scratch).  Default -DRFM_ABLATE: RFM_ABLATE_MASK then switches parts of the forward kernel off
# for timing experiments.   usage: profiles/ablate_build.sh [flags...]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
RFM_HIPCC_FLAGS="${*:--DRFM_ABLATE}" python -c "
from relevance_factorizationmachine_amd import _lib
print(_lib.build(force=True))"
