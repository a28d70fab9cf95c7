#!/bin/bash
# Runs on the GPU box (through gpurun): the sliced loss forward (rfm_fm_sliced.hpp) at the
# reference's published operating point -- its parity tests, bench.py --published-only with and
# without it, and the rocprofv3 kernel trace of the run with it.
#   usage: profiles/sliced_loss.sh <tag> [skip-tests]
set -o pipefail
TAG=${1:-sliced}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
if [ -z "$2" ]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_published.py -q -x -k "sliced or published or merged or chunk_count" > "$OUT/pytest.log" 2>&1
  echo "pytest exit=$?"; tail -5 "$OUT/pytest.log"
  grep -q "failed\|error" "$OUT/pytest.log" && { grep -n "Error\|assert" "$OUT/pytest.log" | head -20; exit 1; }
fi
for MODE in 0 1; do
  RFM_SLICED_LOSS=$MODE timeout -k 10 300 python bench.py --published-only kuairec_fm_ips > "$OUT/published_sliced$MODE.json" 2> "$OUT/published_sliced$MODE.err" || { echo "bench failed"; tail -20 "$OUT/published_sliced$MODE.err"; exit 1; }
  python - "$OUT/published_sliced$MODE.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["published_config"]
for name, e in d.items():
    print(sys.argv[1].rsplit("/", 1)[1], name, {k: e[k] for k in e if k.startswith(("fit", "step", "ms"))})
PY
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- \
  python3 "$R/bench.py" --published-only kuairec_fm_ips > "$OUT/under_rocprof.json" 2> "$OUT/prof.err" || { echo "rocprof failed"; tail -20 "$OUT/prof.err"; exit 1; }
F=$(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1)
[ -n "$F" ] && cp "$F" "$OUT/kernel_stats.csv" && head -9 "$OUT/kernel_stats.csv" | cut -c1-170
rm -rf "$OUT/prof"
