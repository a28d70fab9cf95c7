#!/bin/bash
# Tuning experiments at the published operating point (GPU box): the same bench leg under
# environment knobs of the plan builder / launch geometry.  usage: profiles/published_variants.sh <tag>
set -o pipefail
TAG=${1:-pubvar}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --published-only kuairec_fm_ips > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "$name failed"; tail -5 "$OUT/$name.err"; return 1; }
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["published_config"]["kuairec_fm_ips"]
s = d["step"]
print(f"{sys.argv[2]:28s} step {s['ms_per_step']*1e3:7.1f} us  kernels(us) " + " ".join(f"{k.split('_')[1]}={v*1e3:.1f}" for k, v in s["kernels_avg_ms"].items()) + f"  hot={s['hot_columns']} fit={d['fit_wall']['ms_per_iteration']*1e3:.1f}/{d['fit_wall']['ms_per_iteration_second_fit_same_log']*1e3:.1f} us/it")
PY
}
run base RFM_DUMMY=1 && run task_words_2 RFM_TASK_WORDS=2 && run task_words_8 RFM_TASK_WORDS=8 && run task_words_16 RFM_TASK_WORDS=16
# rebuilt variants (compile-time knobs), last: the box's copy of the library is scratch
for B in 4 6; do
  profiles/ablate_build.sh -DRFM_CONS_CH_BATCH=$B > "$OUT/build_$B.log" 2>&1 && run ch_batch_$B RFM_DUMMY=1
done
