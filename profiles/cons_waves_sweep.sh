#!/bin/bash
# GPU box: the chunk-per-workgroup gradient launch compiled for 0 (no cap) / 4 / 5 / 6 waves per
# SIMD (-DRFM_CONS_CH_WAVES), step and fit() at the published point each.
#   usage: profiles/cons_waves_sweep.sh <tag>
TAG=${1:-conswaves}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
for W in 0 4 5 6; do
  bash profiles/ablate_build.sh -DRFM_CONS_CH_WAVES=$W > "$OUT/build_$W.log" 2>&1 || { tail -5 "$OUT/build_$W.log"; exit 1; }
  timeout -k 10 300 python bench.py --published-only kuairec_fm_ips > "$OUT/pub_$W.json" 2> "$OUT/pub_$W.err" || { tail -5 "$OUT/pub_$W.err"; exit 1; }
  python - "$OUT/pub_$W.json" $W <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["published_config"]["kuairec_fm_ips"]
print("waves", sys.argv[2], "step ms", round(d["step"]["ms_per_step"], 5), {k: round(v, 5) for k, v in d["step"]["kernels_avg_ms"].items()},
      "fit", round(d["fit_wall"]["ms_per_iteration"], 4), round(d["fit_wall"]["ms_per_iteration_second_fit_same_log"], 4))
PY
done | tee "$OUT/summary.txt"
