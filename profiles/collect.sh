#!/bin/bash
# Runs on the GPU box (through gpurun): parity tests, the bench line (which collects its own
# PMC traffic passes: gpurun_out/bench_pmc/pmc_summary.json), and the rocprofv3 kernel-trace
# summary of the same bench command.  Outputs land in gpurun_out/<tag>/ ; what is worth
# keeping is copied to profiles/<tag>/ by hand (with the commit it was taken at).
#   usage: profiles/collect.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout -k 10 900 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit=$?" | tee -a "$OUT/pytest_gpu.log"
tail -3 "$OUT/pytest_gpu.log"
timeout -k 10 900 python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -20 "$OUT/bench.err"; }
cp gpurun_out/bench_pmc/pmc_summary.json "$OUT/pmc_summary.json" 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- \
  python3 "$R/bench.py" "$@" --no-cpu-baseline --no-extra --no-pmc > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof.err" || { echo "rocprof failed"; tail -20 "$OUT/prof.err"; }
F=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv" && head -8 "$OUT/kernel_stats.csv" | cut -c1-150
rm -rf "$OUT/prof"
# the reference's own batch size under the kernel trace as well
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof2k" -- \
  python3 "$R/bench.py" --batch-size 2000 --steps 200 --warmup 20 "$@" --no-cpu-baseline --no-extra --no-pmc > "$OUT/bench_b2000_under_rocprof.json" 2>> "$OUT/prof.err"
F=$(find "$OUT/prof2k" -name "*kernel_stats*.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/kernel_stats_b2000.csv" && head -6 "$OUT/kernel_stats_b2000.csv" | cut -c1-150
rm -rf "$OUT/prof2k"
