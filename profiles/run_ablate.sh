#!/bin/bash
# GPU box: parity tests, then timing sweeps of the FM step with the ablation
# builds (librfm_hip_ablate*.so, compiled with -DRFM_ABLATE).  Output: gpurun_out/<tag>/out.txt
TAG=${1:-abl}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
P=relevance_factorizationmachine_amd
{
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
RFM_LIB_PATH=$R/$P/librfm_hip_ablate.so python profiles/ablate.py base= pc1=RFM_FWD_PER_CU=1 pc3=RFM_FWD_PER_CU=3 pc4=RFM_FWD_PER_CU=4 b256=RFM_FWD_BLOCK=256 \
  hot_off=ABL_HOT=-1 no_marks=RFM_ABLATE_MASK=1 no_atom=RFM_ABLATE_MASK=8 no_hotpass=RFM_ABLATE_MASK=32 no_all=RFM_ABLATE_MASK=63 \
  s2000=ABL_BATCH=2000 s16k=ABL_BATCH=16384 s262k=ABL_BATCH=262144
[ -f $P/librfm_hip_ablate_w1.so ] && RFM_LIB_PATH=$R/$P/librfm_hip_ablate_w1.so python profiles/ablate.py alt_base= alt_pc1=RFM_FWD_PER_CU=1
} > "$OUT/out.txt" 2>&1
cat "$OUT/out.txt"
