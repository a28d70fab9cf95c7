#!/bin/bash
# Runs on the GPU box (through gpurun): the reference's PUBLISHED operating point (k=400/B=2000
# KuaiRec-shaped, k=300/B=500 Coat-shaped) -- bench.py --published-only under the rocprofv3 kernel
# trace and the FETCH_SIZE / WRITE_SIZE counter passes (separate, --kernel-trace only).
#   usage: profiles/published_prof.sh <tag>
set -o pipefail
TAG=${1:-pub}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout -k 10 600 python bench.py --published-only all > "$OUT/published.json" 2> "$OUT/published.err" || { echo "published bench failed"; tail -20 "$OUT/published.err"; exit 1; }
cat "$OUT/published.json" | cut -c1-3000
cd /tmp && export TMPDIR=/tmp
for NAME in kuairec_fm_ips coat_fm_ips; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$NAME" -- \
    python3 "$R/bench.py" --published-only $NAME > "$OUT/${NAME}_under_rocprof.json" 2> "$OUT/prof_$NAME.err" || { echo "rocprof failed"; tail -20 "$OUT/prof_$NAME.err"; exit 1; }
  F=$(find "$OUT/prof_$NAME" -name "*kernel_stats*.csv" | head -1)
  [ -n "$F" ] && cp "$F" "$OUT/kernel_stats_$NAME.csv" && head -8 "$OUT/kernel_stats_$NAME.csv" | cut -c1-160
  rm -rf "$OUT/prof_$NAME"
  for CTR in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 600 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d "$OUT/pmc_${NAME}_$CTR" -- \
      python3 "$R/bench.py" --published-only $NAME > /dev/null 2> "$OUT/pmc_${NAME}_$CTR.err" || { echo "pmc pass failed"; tail -5 "$OUT/pmc_${NAME}_$CTR.err"; exit 1; }
  done
done
cd "$R"
for NAME in kuairec_fm_ips coat_fm_ips; do
  python profiles/pmc_by_instantiation.py "$OUT/pmc_${NAME}_FETCH_SIZE" "$OUT/pmc_${NAME}_WRITE_SIZE" > "$OUT/pmc_$NAME.json" 2>&1 || true
  rm -rf "$OUT/pmc_${NAME}_FETCH_SIZE" "$OUT/pmc_${NAME}_WRITE_SIZE"
done
