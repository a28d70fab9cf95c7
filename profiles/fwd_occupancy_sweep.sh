#!/bin/bash
# GPU box: occupancy variants of the many-rows forward at the headline config (config 3, B = 65 536):
# rows in flight per lane group, waves per SIMD the kernel is compiled for, workgroup size and
# workgroups per CU.  Each variant is a rebuild (the box's copy of the library is scratch).
TAG=${1:-fwdocc}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
run() {  # name, "compile flags", env...
  local name=$1 flags=$2; shift 2
  profiles/ablate_build.sh $flags > "$OUT/build_$name.log" 2>&1 || { echo "$name: build failed"; tail -3 "$OUT/build_$name.log"; return 0; }
  env "$@" python profiles/ablate.py "$name=" 2>&1 | cut -c1-200
}
run base "-DRFM_DUMMY" RFM_X=1
run rows1_w8_pc2 "-DRFM_FWD_ROWS=1 -DRFM_FWD_BIG_WAVES=8" RFM_FWD_PER_CU=2
run rows1_w4_pc1 "-DRFM_FWD_ROWS=1" RFM_FWD_PER_CU=1
run b512_rows2_pc2 "-DRFM_FWD_BIG_BLOCK=512 -DRFM_FWD_BIG_WAVES=4" RFM_FWD_PER_CU=2
run b512_rows1_w8_pc3 "-DRFM_FWD_BIG_BLOCK=512 -DRFM_FWD_ROWS=1 -DRFM_FWD_BIG_WAVES=8" RFM_FWD_PER_CU=3
run b512_rows1_w6_pc3 "-DRFM_FWD_BIG_BLOCK=512 -DRFM_FWD_ROWS=1 -DRFM_FWD_BIG_WAVES=6" RFM_FWD_PER_CU=3
run rows3_w4_pc1 "-DRFM_FWD_ROWS=3 -DRFM_FWD_BIG_UNROLL=2" RFM_FWD_PER_CU=1
