#!/usr/bin/env python
"""All kernels of the LAST rendered frame of a rocprofv3 --kernel-trace CSV (torch kernels included), in launch order, with gaps.
Usage: trace_frame_all.py <dir or csv>"""
import csv
import glob
import os
import subprocess
import sys


def short(name):
    if name.startswith("_Z"):
        try:
            name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    return name.replace("void ", "")[:100]


def main(path):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    torso = [i for i, r in enumerate(rows) if "k_torso" in r["Kernel_Name"] or "k_head_finish" in r["Kernel_Name"]]
    if len(torso) < 2:
        print("need two frames")
        return
    frame = rows[torso[-2] + 1:torso[-1] + 1]      # everything between the ends of the last two frames
    t0 = int(rows[torso[-2]]["End_Timestamp"])
    prev = t0
    busy = 0
    for r in frame:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        busy += e - s
        print(f"{(s - t0) / 1e3:9.1f} | {(e - s) / 1e3:8.1f} | gap {(s - prev) / 1e3:7.1f} | {short(r['Kernel_Name'])}")
        prev = e
    print(f"# frame period {(prev - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, {len(frame)} launches")


if __name__ == "__main__":
    main(sys.argv[1])
