#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command) into
profiles/<round>_traffic_per_launch.json: HBM bytes per launch per kernel.

    python tools/traffic_json.py <fetch_dir> <write_dir> <out.json>

Units / corrections as MI355X_MICROARCH.md prescribes: both counters are in KB; on gfx950 FETCH_SIZE counts 64 B per
128-B request on wide streaming reads, so fetch_bytes_corrected = 2 x raw."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "")).strip()
                acc[name].append(float(r["Counter_Value"]) * 1024.0)
    return acc


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps 1 --warmup 1 "
                    "--no-cpu-baseline --no-events`; KB -> bytes; mean per launch; gfx950 FETCH_SIZE counts 64 B per 128-B request "
                    "on wide streaming reads, so fetch_bytes_corrected = 2 x raw (MI355X_MICROARCH.md, HBM)"}
    for k in sorted(set(fetch) | set(write)):
        fr = sum(fetch[k]) / max(len(fetch[k]), 1)
        wr = sum(write[k]) / max(len(write[k]), 1)
        out[k] = {"launches": max(len(fetch[k]), len(write[k])), "fetch_bytes_raw": round(fr), "fetch_bytes_corrected": round(2 * fr),
                  "write_bytes": round(wr)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print("wrote", sys.argv[3], len(out) - 1, "kernels")


if __name__ == "__main__":
    main()
