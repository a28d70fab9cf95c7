#!/bin/bash
# GPU box: SQ / TCP / TCC counters of the kernels of the reference's PUBLISHED operating point
# (bench.py --published-only kuairec_fm_ips: k = 400, B = 2 000), one rocprofv3 pass per counter
# set (--pmc with --kernel-trace only), averaged per kernel INSTANTIATION.
TAG=${1:-pmcpub}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_ATOMIC_sum" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_BUSY_avr"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/pmc_$N" -- python3 "$R/bench.py" --published-only kuairec_fm_ips > "$OUT/pmc_$N.log" 2>&1 || tail -3 "$OUT/pmc_$N.log"
done
cd "$R"
python - "$OUT" <<'PY' | tee "$OUT/sq_tcp_counters_published.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rfm::", "")
        if not name.startswith("fm_"):
            continue
        a = acc[name][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, d in sorted(acc.items()):
    c = {n: v[0] / v[1] for n, v in d.items()}
    extra = {}
    if c.get("SQ_WAVE_CYCLES"):
        extra["wait_share"] = round(c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"], 3)
    if c.get("TCC_REQ_sum"):
        extra["l2_hit_rate"] = round(c.get("TCC_HIT_sum", 0) / c["TCC_REQ_sum"], 3)
    if c.get("TCP_TCC_READ_REQ_sum"):
        extra["mean_l2_read_latency_cycles"] = round(c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / c["TCP_TCC_READ_REQ_sum"], 1)
    print(k, {n: round(v, 1) for n, v in sorted(c.items())}, extra)
PY
rm -rf "$OUT"/pmc_*/
