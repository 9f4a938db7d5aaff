#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ counter passes per kernel (averages per dispatch, summed over the chip).

  python tools/sq_summary.py --command "..." DIR [DIR ...] > profiles/rNN_sq_counters_c2.json
"""
import argparse
import csv
import glob
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    m = re.search(r"(\w+_kernel<[^>]*>|\w+_kernel)\b", name.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else None


def main(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--command", default="")
    ap.add_argument("--include", default="shade_|bin_kernel|raytile|reduce_kernel")
    ap.add_argument("dirs", nargs="+")
    o = ap.parse_args(argv)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in o.dirs:
        for f in glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                if "anonymous" not in row["Kernel_Name"] or not re.search(o.include, row["Kernel_Name"]):
                    continue
                k = short(row["Kernel_Name"])
                if not k:
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                t = acc[k]["duration_us"]
                t[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                t[1] += 1
    out = {"_meta": {"command": o.command,
                     "note": "averages per dispatch; SQ counters are summed over the chip's 1024 SIMDs"}}
    for k, cs in sorted(acc.items()):
        e = {c: tot / n for c, (tot, n) in cs.items()}
        e["dispatches_sampled"] = max(n for _, n in cs.values())
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]:
            # MFMA_BUSY counts per SIMD, BUSY_CYCLES per SE-level SQ: report both raw; the kernel's own
            # duration x clock gives the per-SIMD denominator
            if e.get("duration_us"):
                e["mfma_busy_frac_of_duration"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["duration_us"] * 2400.0 * 1024)
        out[k] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
