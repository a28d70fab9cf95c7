#!/bin/bash
# GPU box: gathers in flight per lane in the many-rows forward (config 3, B = 65 536): rows per
# lane group x entries unrolled.  Rebuilds; timing only.
TAG=${1:-fwdunroll}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
run() { local name=$1 flags=$2; shift 2
  profiles/ablate_build.sh $flags > "$OUT/build_$name.log" 2>&1 || { echo "$name: build failed"; return 0; }
  env "$@" python profiles/ablate.py "$name=" 2>&1 | cut -c1-170; }
run base "-DRFM_DUMMY" X=1
run rows1_unroll5 "-DRFM_FWD_ROWS=1 -DRFM_FWD_BIG_UNROLL=5" X=1
run rows1_unroll8 "-DRFM_FWD_ROWS=1 -DRFM_FWD_BIG_UNROLL=8" X=1
run rows1_unroll15 "-DRFM_FWD_ROWS=1 -DRFM_FWD_BIG_UNROLL=15" X=1
run rows2_unroll2 "-DRFM_FWD_BIG_UNROLL=2" X=1
run rows2_unroll5 "-DRFM_FWD_BIG_UNROLL=5" X=1
