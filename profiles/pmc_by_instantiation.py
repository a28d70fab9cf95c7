#!/usr/bin/env python3
"""Per-kernel-INSTANTIATION averages of rocprofv3 --pmc passes (template arguments kept: the
training forward, the loss forwards and predict are instantiations of one kernel template).

    python profiles/pmc_by_instantiation.py <dir of FETCH_SIZE pass> <dir of WRITE_SIZE pass>

FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced
reads, so hbm_bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
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
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("rfm::", "")
                acc[name][0] += float(row["Counter_Value"])
                acc[name][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(name, (0.0, 0))
        w, _ = write.get(name, (0.0, 0))
        out[name] = {"launches": nf, "fetch_bytes_x2": 2 * f * 1024, "write_bytes": w * 1024,
                     "hbm_bytes": 2 * f * 1024 + w * 1024}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
