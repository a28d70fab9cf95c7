#!/bin/bash
# GPU box: where a wavefront of the headline forward (config 3, B = 65 536: 256 workgroups of 16
# wavefronts, two trips of 128 rows each) spends its clocks -- the library rebuilt with clock stamps
# (-DRFM_FWD_STAMPS), one bench run, the readings of the 60th step on stderr.   usage: ... <tag>
TAG=${1:-fwdstamps}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
bash profiles/ablate_build.sh -DRFM_FWD_STAMPS > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }
RFM_FWD_STAMPS=1 python bench.py --no-pmc --no-cpu-baseline --no-extra > "$OUT/bench.json" 2> "$OUT/bench.err"
grep "forward stamps" "$OUT/bench.err" | head -3 | tee "$OUT/summary.txt"
