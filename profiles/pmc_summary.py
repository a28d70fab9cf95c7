#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected in separate
runs) into per-kernel average HBM bytes per launch.

    python profiles/pmc_summary.py <dir with pmc_fetch/ and pmc_write/> <batch> [out.json]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads, so the read side is reported both raw and x2 (the
x2 figure is the one `traffic` uses; our loads are 16 B/lane where it matters)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(dirpath, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("rfm::", "")
                acc[name][0] += float(row["Counter_Value"])
                acc[name][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


def main():
    root, batch = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else None
    fetch = per_kernel(os.path.join(root, "pmc_fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(root, "pmc_write"), "WRITE_SIZE")
    summary = {}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, 0.0) * 1024, write.get(name, 0.0) * 1024
        summary[name] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w,
                         "hbm_bytes": 2 * f + w}
    print(json.dumps(summary, indent=1))
    if out:
        traffic = {}
        if os.path.exists(out):
            traffic = json.load(open(out))
        for name, v in summary.items():
            traffic.setdefault(name, {})[str(batch)] = v["hbm_bytes"]
        json.dump(traffic, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
