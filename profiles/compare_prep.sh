#!/bin/bash
# GPU box: prepared steps on (RFM_PREP=1) / off at B = 2 000 (config 3) and at the published
# operating points, the headline step, and the parity tests.   usage: profiles/compare_prep.sh <tag>
TAG=${1:-cmp}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
if "published_config" in d:
    for k, v in d["published_config"].items():
        print(f"{sys.argv[2]:10s} {k:16s} step {1e3*v['step']['ms_per_step']:6.1f} us  " +
              " ".join(f"{n.split('_')[1]}={1e3*t:.1f}" for n, t in v["step"]["kernels_avg_ms"].items()) +
              f"  fit {1e3*v['fit_wall']['ms_per_iteration']:.1f} / {1e3*v['fit_wall']['ms_per_iteration_second_fit_same_log']:.1f} us/it")
else:
    print(f"{sys.argv[2]:10s} B={d['config']['batch_size_per_gpu']:6d} step {1e3*d['ms_per_step']:6.2f} us  " +
          " ".join(f"{n.split('_')[1]}={1e3*t:.1f}" for n, t in d["roofline"]["all_kernels_avg_ms"].items()))
PY
}
for v in prep prep2 noprep; do
  E=RFM_PREP=1; [ $v = prep2 ] && E=RFM_PREP=2; [ $v = noprep ] && E=RFM_PREP=0
  env $E python bench.py --batch-size 2000 --steps 200 --warmup 20 --no-cpu-baseline --no-extra --no-pmc > "$OUT/b2000_$v.json" 2>> "$OUT/err.txt" && show "$OUT/b2000_$v.json" $v
  env $E python bench.py --published-only all > "$OUT/pub_$v.json" 2>> "$OUT/err.txt" && show "$OUT/pub_$v.json" $v
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc > "$OUT/b65536.json" 2>> "$OUT/err.txt" && show "$OUT/b65536.json" headline
timeout -k 10 700 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit=$?"; tail -2 "$OUT/pytest_gpu.log"
