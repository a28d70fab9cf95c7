#!/bin/bash
# Runs on the GPU box: the sliced loss forward rebuilt with clock stamps (-DRFM_SLICED_STAMPS),
# one published-point fit, the phase readings on stderr.   usage: profiles/sliced_stamps.sh <tag>
set -o pipefail
TAG=${1:-stamps}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
bash profiles/ablate_build.sh -DRFM_SLICED_STAMPS > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 1; }
RFM_SLICED_STAMPS=1 timeout -k 10 300 python bench.py --published-only kuairec_fm_ips > "$OUT/published.json" 2> "$OUT/published.err"
grep "sliced stamps" "$OUT/published.err" | tail -4
