#!/bin/bash
# GPU box: rows per lane group x entries unrolled in the PLAIN many-rows forward (rebuilds).
TAG=${1:-plainfwd}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
for V in "2 3" "3 3" "4 2" "4 3" "4 4" "6 2" "8 1"; do
  set -- $V
  profiles/ablate_build.sh -DRFM_FWD_ROWS_PLAIN=$1 -DRFM_FWD_PLAIN_UNROLL=$2 > "$OUT/build.log" 2>&1 || { echo "rows=$1 unroll=$2: build failed"; tail -3 "$OUT/build.log"; continue; }
  echo "rows=$1 unroll=$2: $(python profiles/val_forward_bench.py 2>/dev/null | tail -1)"
done | tee "$OUT/out.txt"
