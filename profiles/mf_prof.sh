#!/bin/bash
# GPU box: extra.mf of bench.py (BASELINE config 5) plain and under the rocprofv3 kernel trace.
#   usage: profiles/mf_prof.sh <tag>
set -o pipefail
TAG=${1:-mf}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout -k 10 900 python bench.py --mf-only all > "$OUT/mf.json" 2> "$OUT/mf.err" || { echo "mf bench failed"; tail -20 "$OUT/mf.err"; exit 1; }
cut -c1-4000 "$OUT/mf.json"
cd /tmp && export TMPDIR=/tmp
for NAME in c2_kuairec_small_k16 c5_1m_x_100k_k128; do
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$NAME" -- \
    python3 "$R/bench.py" --mf-only $NAME > "$OUT/${NAME}_under_rocprof.json" 2> "$OUT/prof_$NAME.err" || { echo "rocprof failed"; tail -20 "$OUT/prof_$NAME.err"; exit 1; }
  F=$(find "$OUT/prof_$NAME" -name "*kernel_stats*.csv" | head -1)
  [ -n "$F" ] && cp "$F" "$OUT/mf_kernel_stats_$NAME.csv" && head -8 "$OUT/mf_kernel_stats_$NAME.csv" | cut -c1-160
  rm -rf "$OUT/prof_$NAME"
done
