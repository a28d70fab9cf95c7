#!/bin/bash
# GPU box: phase ablation of the step at the published point (C2-shaped log, k = 400, B = 2 000).
TAG=${1:-ablk400}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
profiles/ablate_build.sh -DRFM_ABLATE > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }
export ABL_SHAPE=kuairec_small ABL_K=400 ABL_BATCH=2000
python profiles/ablate.py base= no_marks=RFM_ABLATE_MASK=1 no_q=RFM_ABLATE_MASK=2 one_v=RFM_ABLATE_MASK=4 bare=RFM_ABLATE_MASK=7 \
  only_fwd=RFM_ABLATE_MASK=128 only_cons=RFM_ABLATE_MASK=64 2>&1 | cut -c1-190 | tee "$OUT/out.txt"
