#!/bin/bash
# GPU box: the forward's clock-stamp timeline (profiles/forward_stamps.sh) with parts of the hot pass
# switched off (-DRFM_ABLATE masks: 8 = no LDS adds, 512 = no adds of the five always-present columns
# after a workgroup's first trip, 32 = no hot pass, 1 = no marks).   usage: ... <tag>
TAG=${1:-fwdstampsabl}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
bash profiles/ablate_build.sh -DRFM_FWD_STAMPS -DRFM_ABLATE > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }
for M in 0 8 512 32 1; do
  echo "RFM_ABLATE_MASK=$M"
  RFM_ABLATE_MASK=$M RFM_FWD_STAMPS=1 python bench.py --no-pmc --no-cpu-baseline --no-extra > "$OUT/bench_$M.json" 2> "$OUT/bench_$M.err"
  grep "forward stamps" "$OUT/bench_$M.err" | head -3 | cut -c1-420
done | tee "$OUT/summary.txt"
