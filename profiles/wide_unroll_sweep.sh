#!/bin/bash
# GPU box: entries' gathers in flight in the one-row forward at k = 300 / 400 (rebuilds).
TAG=${1:-wideunroll}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
for U in 2 3 4 6 8; do
  profiles/ablate_build.sh -DRFM_FWD_SMALL_WIDE_UNROLL=$U > "$OUT/build.log" 2>&1 || { echo "unroll=$U: build failed"; continue; }
  python bench.py --published-only all 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)['published_config']
for k,v in d.items(): print('unroll=$U', k, round(1e3*v['step']['ms_per_step'],1), {a.split('_')[1]:round(1e3*b,1) for a,b in v['step']['kernels_avg_ms'].items()}, round(1e3*v['fit_wall']['ms_per_iteration_second_fit_same_log'],1))"
done | tee "$OUT/out.txt"
