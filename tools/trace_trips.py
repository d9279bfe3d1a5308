#!/usr/bin/env python
"""Per-launch view of a rocprofv3 --kernel-trace CSV: durations of the gfpp kernels of the LAST rendered frame, in launch order
(one line per non-trivial launch), plus the per-kernel totals.  Usage: trace_trips.py <dir or *_kernel_trace.csv>"""
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


def main(path):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ours = [r for r in rows if "gfpp" in r["Kernel_Name"]]
    # last frame = from the last k_frame_begin on
    starts = [i for i, r in enumerate(ours) if "k_frame_begin" in r["Kernel_Name"] or "k_begin_premarch" in r["Kernel_Name"]]
    last = ours[starts[-1]:] if starts else ours
    t0 = int(last[0]["Start_Timestamp"])
    print(f"# {path}\n# last frame, launch order: start us | duration us | gap to previous end us | kernel")
    prev_end = t0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = short(r["Kernel_Name"])
        print(f"{(s - t0) / 1e3:9.1f} | {(e - s) / 1e3:8.1f} | {(s - prev_end) / 1e3:6.1f} | {name}")
        prev_end = e
    print(f"# frame span {(prev_end - t0) / 1e3:.1f} us")
    tot = defaultdict(lambda: [0, 0])
    for r in ours:
        name = short(r["Kernel_Name"])
        tot[name][0] += 1
        tot[name][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"# {k}: {n} launches, {t / 1e6:.3f} ms total, {t / n / 1e3:.1f} us avg")


if __name__ == "__main__":
    main(sys.argv[1])
