#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace into the per-kernel stats table that `--stats` prints:
name, calls, total ms, average us, min us, max us, % of GPU kernel time.  Usage: rocpd_summary.py results.db [out.md]"""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), max(accum_vgpr_count), "
                       "max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch B |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx, vg, ag, sg, lds, scr in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append(f"| `{short}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.1f} | {vg} | {ag} | {sg} | {lds} | {scr} |")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
