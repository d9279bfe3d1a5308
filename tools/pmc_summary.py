#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes: per gfpp kernel, the sum of every counter over the dispatches of the LAST frame
(and the per-dispatch values of the trip kernel).  Usage: pmc_summary.py <pass dir> [<pass dir> ...]"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict


def short(name):
    if name.startswith("_Z"):
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    return name.split("(")[0].replace("void ", "").replace("gfpp::", "")


def main(dirs):
    for d in dirs:
        files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
        if not files:
            print("#", d, "no counter file")
            continue
        rows = [r for r in csv.DictReader(open(files[-1])) if "gfpp" in r["Kernel_Name"]]
        disp = defaultdict(dict)     # dispatch id -> {counter: value}
        names = {}
        for r in rows:
            k = int(r["Dispatch_Id"])
            disp[k][r["Counter_Name"]] = float(r["Counter_Value"])
            names[k] = short(r["Kernel_Name"])
        ids = sorted(disp)
        begins = [k for k in ids if "k_frame_begin" in names[k] or "k_begin_premarch" in names[k]]
        last = [k for k in ids if not begins or k >= begins[-1]]
        counters = sorted({c for k in last for c in disp[k]})
        print("#", d)
        tot = defaultdict(lambda: defaultdict(float))
        for k in last:
            for c, v in disp[k].items():
                tot[names[k]][c] += v
        for n, cs in tot.items():
            print(n, {c: round(v, 1) for c, v in cs.items()})
        trips = [k for k in last if "k_head_trip" in names[k] or "k_head_frame_persist" in names[k]]
        for c in counters:
            print("  per trip", c, [round(disp[k].get(c, 0.0), 1) for k in trips[:8]])


if __name__ == "__main__":
    main(sys.argv[1:])
