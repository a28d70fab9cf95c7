#!/bin/bash
# GPU box: phase ablation of the B = 2 000 step (config 3, k = 32) with the -DRFM_ABLATE build:
# RFM_ABLATE_MASK bits: 1 slot marks, 2 Q store, 4 V gathers to one row, 8 hot LDS adds,
# 16 slab store, 32 hot pass, 64 no forward launch, 128 no gradient launch.
TAG=${1:-abl2000}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
profiles/ablate_build.sh -DRFM_ABLATE > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }
export ABL_BATCH=2000
python profiles/ablate.py base= prep=RFM_PREP=1 no_q=RFM_ABLATE_MASK=2 one_v=RFM_ABLATE_MASK=4 no_atom=RFM_ABLATE_MASK=8 \
  no_slab=RFM_ABLATE_MASK=16 no_hotpass=RFM_ABLATE_MASK=32 no_hot_slab=RFM_ABLATE_MASK=48 bare=RFM_ABLATE_MASK=62 \
  only_fwd=RFM_ABLATE_MASK=128 only_cons=RFM_ABLATE_MASK=64 neither=RFM_ABLATE_MASK=192 hot_off=ABL_HOT=-1 "$@" 2>&1 | cut -c1-230 | tee "$OUT/out.txt"
