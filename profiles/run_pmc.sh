#!/bin/bash
# GPU box: standalone kernel timings (other launches of the step skipped) and PMC
# counters of the step's kernels.  Output: gpurun_out/<tag>/
TAG=${1:-pmc}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$R"
P=relevance_factorizationmachine_amd
export RFM_LIB_PATH=$R/$P/librfm_hip_ablate.so
python profiles/ablate.py only_cons=RFM_ABLATE_MASK=320 only_cons_nomarks=RFM_ABLATE_MASK=321 only_fwd=RFM_ABLATE_MASK=384 only_fin=RFM_ABLATE_MASK=192 \
   cons_fin=RFM_ABLATE_MASK=64 fwd_cons=RFM_ABLATE_MASK=256 base= > "$OUT/standalone.txt" 2>&1
cat "$OUT/standalone.txt"
cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$N" -- python3 "$R/profiles/ablate.py" --child > "$OUT/pmc_$N.log" 2>&1 || tail -3 "$OUT/pmc_$N.log"
done
cd "$R"
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("rfm::", "")
        a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k, {c: round(v[0] / v[1], 1) for c, v in sorted(d.items())})
PY
find "$OUT" -name "*.csv" -size +1M -delete
