#!/bin/bash
# GPU box: is the plain forward (validation loss / predict) bound by its gathers of V rows?  -DRFM_ABLATE
# build, profiles/val_forward_bench.py as it is and with every gather pointed at row 0 of V
# (RFM_ABLATE_MASK=4: the same instructions, all hits on one line).   usage: ... <tag>
TAG=${1:-valgather}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
bash profiles/ablate_build.sh -DRFM_ABLATE > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }
for M in 0 4; do
  echo "RFM_ABLATE_MASK=$M"
  RFM_ABLATE_MASK=$M timeout -k 10 300 python profiles/val_forward_bench.py 2>/dev/null | tail -8
done | tee "$OUT/summary.txt"
