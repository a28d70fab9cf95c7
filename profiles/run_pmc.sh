#!/bin/bash
# GPU box: PMC counters of the FM step's kernels at B = 65536 (config 3), one rocprofv3 pass
# per counter set (--pmc with --kernel-trace only), averaged per kernel.  Output:
# gpurun_out/<tag>/sq_tcp_counters.txt
TAG=${1:-pmc}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_ATOMIC_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS_ATOMIC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_BUSY_avr"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$N" -- python3 "$R/profiles/ablate.py" --child > "$OUT/pmc_$N.log" 2>&1 || tail -3 "$OUT/pmc_$N.log"
done
cd "$R"
python - "$OUT" <<'PY' | tee "$OUT/sq_tcp_counters.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("rfm::", "")
        if not name.startswith("fm_"):
            continue
        a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in sorted(acc.items()):
    print(k, {c: round(v[0] / v[1], 1) for c, v in sorted(d.items())})
PY
rm -rf "$OUT"/pmc_*/
