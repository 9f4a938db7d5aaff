#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (average per dispatch).

  python tools/pmc_summary.py [--workload c2] [--command "..."] gpurun_out/pmc_fetch gpurun_out/pmc_write \
      > profiles/rNN_pmc_traffic.json

The summary is stamped (`_meta`) with a fingerprint of the kernel sources it was measured on, so
bench.py quotes `roofline.traffic` only while the kernels are unchanged.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section): `read_bytes_corrected` applies that factor; gather-dominated
kernels are uncalibrated, so both raw and corrected values are kept.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(\w+_kernel)\b", name)
    if m and "anonymous" in name:
        return m.group(1)
    return None


def main(argv):
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--command", default="")
    ap.add_argument("--launches", type=int, default=0,
                    help="operator launches in the profiled run (bench steps + warm-ups, times chunks per "
                         "step): lets bench.py turn per-dispatch bytes into bytes per operator launch when "
                         "an operator is several dispatches (the C5 backward runs in rounds)")
    ap.add_argument("dirs", nargs="+")
    opts = ap.parse_args(argv)
    dirs = opts.dirs
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                if not k:
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                t = acc[k]["duration_us"]
                t[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                t[1] += 1
    out = {}
    for k, cs in sorted(acc.items()):
        e = {"dispatches": max(v[1] for v in cs.values()),
             "dispatches_per_pass": max(v[1] for c, v in cs.items() if c != "duration_us")}
        for c, (tot, n) in cs.items():
            e[c + "_avg"] = tot / n
        if "FETCH_SIZE_avg" in e:
            e["read_bytes_raw"] = e["FETCH_SIZE_avg"] * 1024
            e["read_bytes_corrected"] = e["FETCH_SIZE_avg"] * 2048
        if "WRITE_SIZE_avg" in e:
            e["write_bytes"] = e["WRITE_SIZE_avg"] * 1024
        out[k] = e
    from bench import kernels_sha16
    out["_meta"] = {"workload": opts.workload, "kernels_sha16": kernels_sha16(), "command": opts.command, "operator_launches": opts.launches,
                    "git_sha": os.environ.get("F2N_GIT_SHA", "")}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
