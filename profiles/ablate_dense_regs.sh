#!/bin/bash
# GPU box: upper bound of what register accumulation of the always-present columns across a
# workgroup's trips could save in the forward: -DRFM_ABLATE build, mask 512 drops
# exactly the LDS adds such a scheme would drop (results are wrong by construction; timing only).
TAG=${1:-dense}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
profiles/ablate_build.sh -DRFM_ABLATE > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }
for i in 1 2 3; do
python profiles/ablate.py base= drop_dense_after_first_trip=RFM_ABLATE_MASK=512 no_atom=RFM_ABLATE_MASK=8 2>&1 | cut -c1-160
done | tee "$OUT/out.txt"
