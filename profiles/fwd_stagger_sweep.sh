#!/bin/bash
# GPU box: the upper half of the forward's wavefronts starting late (-DRFM_FWD_STAGGER=<units of 64
# clocks>), headline step for 0 / 32 / 64 / 100 / 127 / 200 / 254 units, three bench runs each.
#   usage: profiles/fwd_stagger_sweep.sh <tag>
TAG=${1:-stagger}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
for D in 0 32 64 100 127 200 254; do
  bash profiles/ablate_build.sh -DRFM_FWD_STAGGER=$D > "$OUT/build_$D.log" 2>&1 || { tail -5 "$OUT/build_$D.log"; exit 1; }
  for rep in 1 2 3; do
    python bench.py --no-pmc --no-cpu-baseline --no-extra > "$OUT/b_$D.json" 2>/dev/null
    python - "$OUT/b_$D.json" $D <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("stagger", sys.argv[2], "ms/step", round(d["ms_per_step"], 5), {k: round(v, 5) for k, v in d["roofline"]["all_kernels_avg_ms"].items()})
PY
  done
done | tee "$OUT/summary.txt"
