#!/bin/bash
# GPU box: the chunk-per-workgroup gradient launch with 4 (default) / 3 / 2 entries' Q and V rows in
# flight (-DRFM_CONS_CH_BATCH: 134 / 120 / 106 VGPRs, i.e. 3 / 4 / 4 waves per SIMD resident), step and
# fit() at the published point each.   usage: profiles/cons_batch_sweep.sh <tag>
TAG=${1:-consbatch}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
for B in 4 3 2; do
  bash profiles/ablate_build.sh -DRFM_CONS_CH_BATCH=$B > "$OUT/build_$B.log" 2>&1 || { tail -5 "$OUT/build_$B.log"; exit 1; }
  for NAME in kuairec_fm_ips coat_fm_ips; do
  timeout -k 10 300 python bench.py --published-only $NAME > "$OUT/pub_${B}_$NAME.json" 2> "$OUT/pub_$B.err" || { tail -5 "$OUT/pub_$B.err"; exit 1; }
  python - "$OUT/pub_${B}_$NAME.json" $B $NAME <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["published_config"][sys.argv[3]]
print("batch", sys.argv[2], sys.argv[3], "step ms", round(d["step"]["ms_per_step"], 5), {k: round(v, 5) for k, v in d["step"]["kernels_avg_ms"].items()},
      "fit", round(d["fit_wall"]["ms_per_iteration"], 4), round(d["fit_wall"]["ms_per_iteration_second_fit_same_log"], 4))
PY
  done
done | tee "$OUT/summary.txt"
