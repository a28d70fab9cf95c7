#!/bin/bash
# Runs on the GPU box (through gpurun): parity tests, the bench line, and the
# rocprofv3 kernel-trace summary of the same bench command.  Outputs land in
# gpurun_out/<tag>/ ; the summaries worth keeping are copied to profiles/ by hand.
#   usage: profiles/collect.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit=$?" | tee -a "$OUT/pytest_gpu.log"
tail -5 "$OUT/pytest_gpu.log"
timeout -k 10 600 python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -20 "$OUT/bench.err"; }
cat "$OUT/bench.json"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- \
  python3 "$R/bench.py" "$@" --no-cpu-baseline --no-extra > "$OUT/prof_bench.json" 2> "$OUT/prof.err" || { echo "rocprof failed"; tail -20 "$OUT/prof.err"; }
find "$OUT/prof" -name "*kernel_stats*.csv" | head -3
F=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv" && head -12 "$OUT/kernel_stats.csv"
# the per-dispatch trace is large: keep only the stats
find "$OUT/prof" -name "*kernel_trace*.csv" -size +2M -delete
# HBM traffic: FETCH_SIZE and WRITE_SIZE in their own passes (TCC slots do not fit both)
cd /tmp
for C in fetch:FETCH_SIZE write:WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --pmc ${C#*:} --kernel-trace --output-format csv -d "$OUT/pmc_${C%%:*}" -- \
    python3 "$R/bench.py" "$@" --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/pmc_${C%%:*}.err" || { echo "pmc ${C#*:} failed"; tail -5 "$OUT/pmc_${C%%:*}.err"; }
done
cd "$R"
python profiles/pmc_summary.py "$OUT" 65536 "$OUT/traffic.json" > "$OUT/pmc_summary.json" 2>&1; cat "$OUT/pmc_summary.json" | head -40
find "$OUT" -name "*kernel_trace*.csv" -size +2M -delete
find "$OUT" -name "*counter_collection*.csv" -size +8M -delete
