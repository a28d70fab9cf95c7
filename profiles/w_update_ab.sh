#!/bin/bash
# GPU box: w[col] of a finished column updated by a no-return atomic add (default) or by a
# read-modify-write (-DRFM_W_RMW=1), headline step / B = 2 000 / published point, three runs each.
#   usage: profiles/w_update_ab.sh <tag>
TAG=${1:-wupdate}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
for V in 1 0; do
  bash profiles/ablate_build.sh -DRFM_W_RMW=$V > "$OUT/build_$V.log" 2>&1 || { tail -5 "$OUT/build_$V.log"; exit 1; }
  for rep in 1 2 3; do
    python bench.py --no-pmc --no-cpu-baseline --no-extra > "$OUT/b_$V.json" 2>/dev/null
    python bench.py --batch-size 2000 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline --no-extra > "$OUT/b2k_$V.json" 2>/dev/null
    python bench.py --published-only kuairec_fm_ips > "$OUT/p_$V.json" 2>/dev/null
    python - "$OUT" $V <<'PY'
import json, sys
o, v = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{o}/b_{v}.json").read().strip().splitlines()[-1])
e = json.loads(open(f"{o}/b2k_{v}.json").read().strip().splitlines()[-1])
p = json.load(open(f"{o}/p_{v}.json"))["published_config"]["kuairec_fm_ips"]
print("rmw" if v == "1" else "atomic", "B=65536", round(d["ms_per_step"], 5), round(d["roofline"]["all_kernels_avg_ms"]["fm_consume_kernel"], 5),
      "| B=2000", round(e["ms_per_step"], 5), "| k=400", round(p["step"]["ms_per_step"], 5), round(p["step"]["kernels_avg_ms"]["fm_consume_kernel"], 5))
PY
  done
done | tee "$OUT/summary.txt"
