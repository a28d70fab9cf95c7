#!/bin/bash
# GPU box: where a task wavefront of the gradient launch spends its clocks -- the library rebuilt with
# clock stamps (-DRFM_CONS_STAMPS), the headline step (B = 65 536), B = 2 000 and the published point
# (k = 400, B = 2 000); the readings of the 60th step on stderr.   usage: profiles/consume_stamps.sh <tag>
TAG=${1:-consstamps}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
bash profiles/ablate_build.sh -DRFM_CONS_STAMPS > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }
{
echo "config 3, B = 65 536:"; RFM_CONS_STAMPS=1 python bench.py --no-pmc --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "consume stamps" | head -2
echo "config 3, B = 2 000:"; RFM_CONS_STAMPS=1 python bench.py --batch-size 2000 --steps 200 --warmup 20 --no-pmc --no-cpu-baseline --no-extra 2>&1 >/dev/null | grep "consume stamps" | head -2
echo "published point (k = 400, B = 2 000):"; RFM_CONS_STAMPS=1 python bench.py --published-only kuairec_fm_ips 2>&1 >/dev/null | grep "consume stamps" | head -2
} | tee "$OUT/summary.txt"
