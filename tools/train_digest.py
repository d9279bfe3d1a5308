#!/usr/bin/env python
"""gpurun_out/<tag>_stats (rocprofv3 --kernel-trace --stats of tools/profile_train.py) -> profiles/<tag>_kernel_stats.md.   python tools/train_digest.py <tag> "<note>" """
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
rows = list(csv.DictReader(open(os.path.join(ROOT, "gpurun_out", f"{tag}_stats", "t_kernel_stats.csv"))))
line = [l for l in open(os.path.join(ROOT, "gpurun_out", f"{tag}.log")) if l.startswith("profile_train")][-1].strip()
tot = sum(int(r["TotalDurationNs"]) for r in rows)


def name(n):
    if n.startswith("_Z"):
        try:        # (c++filt does not know the _Float16 mangling DF16_: hand it the IEEE-half code Dh)
            n = subprocess.run(["c++filt", n.replace("DF16_", "Dh")], capture_output=True, text=True).stdout.strip() or n
        except OSError:
            pass
    n = n.replace("__fp16", "_Float16").replace("half", "_Float16") if "gfpp::" in n else n
    return n if len(n) < 100 else n[:97] + "..."


out = [f"# rocprofv3 --kernel-trace --stats -- python tools/profile_train.py 65536 6 {'amp' if 'amp' in tag else ''}   ({tag}, MI355X; 14 training steps: 2 warm-up, 6 timed one by one, 6 back to back)", "",
       note, "", "```", line, "```", "", "| total ms | share | calls | avg us | kernel |", "|---|---|---|---|---|"]
for r in rows[:24]:
    out.append(f"| {int(r['TotalDurationNs']) / 1e6:.2f} | {float(r['Percentage']):.1f}% | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | `{name(r['Name'])}` |")
calls = sum(int(r["Calls"]) for r in rows)
out += ["", f"Sum over all kernels: {tot / 1e6:.1f} ms = {tot / 14e6:.2f} ms per step; {calls} launches = {calls / 14:.0f} per step."]
open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))
