#!/bin/bash
# GPU box: how much of the MF level chain's time is the workgroup barrier -- the sequential kernel
# rebuilt WITHOUT the barrier between two one-example levels (-DRFM_MF_PROBE_NOBARRIER: a timing probe,
# its results are not valid), bench.py --mf-only before and after.   usage: profiles/mf_barrier_probe.sh <tag>
TAG=${1:-mfprobe}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
for V in base probe; do
  [ $V = probe ] && { bash profiles/ablate_build.sh -DRFM_MF_PROBE_NOBARRIER > "$OUT/build.log" 2>&1 || { tail -5 "$OUT/build.log"; exit 1; }; }
  timeout -k 10 600 python bench.py --mf-only all > "$OUT/mf_$V.json" 2> "$OUT/mf_$V.err" || { tail -5 "$OUT/mf_$V.err"; exit 1; }
  python - "$OUT/mf_$V.json" $V <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["mf"]
for name, v in d.items():
    for b, e in v.items():
        if isinstance(e, dict) and "exact" in e:
            print(sys.argv[2], name, b, "levels", e.get("levels_per_batch"), "exact ms", round(e["exact"]["ms_per_batch"], 4), "us/level", e["exact"].get("us_per_level"))
PY
done | tee "$OUT/summary.txt"
