#!/usr/bin/env python
"""Turn the raw outputs of tools/profile_round.sh (gpurun_out/<tag>_*) into the summaries committed under profiles/:
   profiles/<tag>_kernel_stats.md   rocprofv3 --kernel-trace --stats of the bench command (per-kernel table + the bench JSON line)
   profiles/<tag>_pmc.md            per-kernel PMC sums of the last rendered frame, per counter pass
   profiles/<round>_pmc_traffic_<precision>.json   fabric-side bytes of the trip launches per frame (read back by bench.py as roofline.traffic)
Usage: profile_digest.py <tag> <precision>"""
import csv
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(name):
    if name.startswith("_Z"):
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    if name.startswith("_ZN4gfpp"):   # llvm-cxxfilt of ROCm 7.2 does not know the _Float16 / __bf16 manglings (DF16_, DF16b)
        m = re.match(r"_ZN4gfpp\d+(k_\w+?)I(.*)EEvNS_", name)
        if m:
            args = m.group(2).replace("DF16_", "_Float16, ").replace("DF16b", "__bf16, ").replace("Li", "").replace("Lb0", "false").replace("Lb1", "true").replace("E", ", ")
            name = "gfpp::" + m.group(1) + "<" + args.strip(", ").replace(", ,", ",") + ">"
    return name


def main(tag, precision, bench_args=None):
    src = os.path.join(ROOT, "gpurun_out")
    dst = os.path.join(ROOT, "profiles")
    rows = list(csv.DictReader(open(os.path.join(src, f"{tag}_stats", "bench_kernel_stats.csv"))))
    bench_line = [l for l in open(os.path.join(src, f"{tag}_bench.log")) if l.startswith('{"metric')][-1].strip()
    out = [f"# rocprofv3 --kernel-trace --stats -- {tag}",
           "",
           f"Command (1x MI355X via gpurun): `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py "
           + (bench_args or f"--steps 40 --warmup 4 --precision {precision} --no-cpu-baseline --no-modes --no-configs") + "`",
           "",
           "bench line of the profiled run (profiling costs a few % of wall time):", "", "```", bench_line, "```", "",
           "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for r in rows[:22]:
        name = demangle(r["Name"])
        name = name if len(name) < 110 else name[:107] + "..."
        out.append(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | {int(r['MinNs']) / 1e3:.2f} | "
                   f"{int(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.1f} |")
    d = json.loads(bench_line)
    rf = d.get("roofline", {})
    is_head = lambda n: "k_head_trip" in n or "k_head_frame_persist" in n
    persist = any("k_head_frame_persist" in r["Name"] for r in rows)
    trip = [r for r in rows if is_head(r["Name"])]
    if trip:
        calls = sum(int(r["Calls"]) for r in trip)
        # launches per frame of the roofline section (one frame at a time): fp32 = one per trip; 16-bit = ONE persistent launch (round 3; 5 + 1 multi-trip before)
        per_frame = 16 if precision == "fp32" else (1 if persist else 6)
        tot_ms = sum(int(r["TotalDurationNs"]) for r in trip) / 1e6
        out += ["", f"Trip launches in this trace: {calls} dispatches, {tot_ms:.3f} ms in total.  The timed loop keeps several frames in flight (one stream per lane), so a launch there "
                    "shares the GPU with the other frame's kernels and its duration is not a property of the kernel alone; the roofline is therefore quoted on the launches of "
                    "bench.py's roofline section, which renders one frame at a time after the timed loop:"]
        tpath = os.path.join(src, f"{tag}_stats", "bench_kernel_trace.csv")
        if os.path.exists(tpath):
            # (without the ONE launch of the profiling instantiation k_head_frame_persist<.., PROF = true> that bench.py issues behind its five timed launches)
            tr = [r for r in csv.DictReader(open(tpath)) if is_head(r["Kernel_Name"]) and not re.search(r"Lb[01]ELb[01]ELb1EEEv", r["Kernel_Name"])]
            tr.sort(key=lambda r: int(r["Start_Timestamp"]))
            last = tr[-5 * per_frame:]
            dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
            busy = [d for d in dur if d > 15.0]
            fpl = int(rf.get("frames_per_launch") or 1)                    # round 4: a head launch renders a group of consecutive frames
            what = f"5 launches of {fpl} frames each" if fpl > 1 else "5 frames"
            out += ["", f"last {len(last)} head dispatches of the trace (= the roofline section's {what}): {len(busy)} non-empty launches, average {sum(busy) / max(len(busy), 1):.2f} us "
                        f"({sum(dur) / 5 / fpl:.1f} us of head launches per frame); bench.py's HIP-event measurement of the same launches: {rf.get('avg_launch_ms')} ms per non-empty launch, "
                        f"{rf.get('ms_per_frame_all_trips')} ms per frame."]
            rest = tr[:-5 * per_frame]
            rdur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rest]
            rbusy = [d for d in rdur if d > 15.0]
            if rbusy:
                out += [f"All earlier dispatches (warm-up + timed loop, frames overlapping): {len(rbusy)} non-empty launches, average {sum(rbusy) / len(rbusy):.2f} us."]
    open(os.path.join(dst, f"{tag}_kernel_stats.md"), "w").write("\n".join(out) + "\n")
    if not os.path.exists(os.path.join(src, f"{tag}_pmc.txt")):
        return

    pmc = open(os.path.join(src, f"{tag}_pmc.txt")).read()
    pmc = "\n".join(demangle(w) if w.startswith("_Z") else w for w in re.split(r"(\s+)", pmc)) if False else pmc
    lines = []
    for line in pmc.splitlines():
        m = re.match(r"^(_Z\S+)(.*)$", line)
        lines.append((demangle(m.group(1)).split("(")[0] + m.group(2)) if m else line)
    head = [f"# rocprofv3 --pmc passes -- {tag}", "",
            f"Each counter group in its own run (`rocprofv3 --pmc <group> --kernel-trace -- python tools/profile_frame.py may_torso 512 3 {precision}`), summed per kernel over the",
            ("dispatches of the LAST rendered frame (16 trip launches, 6 of them non-empty);" if precision == "fp32" else
             "dispatches of the LAST rendered frame (head pass: ONE persistent launch per frame since round 3; before: one launch per trip for the first five trips + one multi-trip launch);"),
            "`per trip` lists those launches in order.",
            "Units as rocprofv3 reports them: FETCH_SIZE / WRITE_SIZE in KiB of fabric-side (L2 <-> Infinity Cache / HBM) traffic -- on gfx950 a wide coalesced read",
            "is under-reported by 2x and other access shapes are uncalibrated (MI355X_MICROARCH.md, HBM section), so read them as lower bounds and compare runs, not absolutes;",
            "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over wavefronts; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; GRBM_GUI_ACTIVE in cycles summed over XCDs.", "",
            "```"]
    open(os.path.join(dst, f"{tag}_pmc.md"), "w").write("\n".join(head + lines + ["```"]) + "\n")

    def grab(counter):
        m = re.search(r"(?:k_head_trip|k_head_frame_persist)\S* \{[^}]*'" + counter + r"': ([0-9.]+)", pmc)
        return float(m.group(1)) if m else None
    fetch, write = grab("FETCH_SIZE"), grab("WRITE_SIZE")
    hit, miss = grab("TCC_HIT_sum"), grab("TCC_MISS_sum")
    m = re.search(r"per trip FETCH_SIZE \[([^\]]*)\]", pmc)
    per_trip = [float(v) for v in m.group(1).split(",")] if m else []
    nonempty = max(1, sum(1 for v in per_trip if v > 1000.0))
    # MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B fabric request for 16-B-per-lane reads -> double it.
    # WRITE_SIZE is uncalibrated and taken as reported.  KiB -> bytes.
    fetch_b = 2.0 * fetch * 1024 if fetch else None
    write_b = write * 1024 if write else None
    traffic = {"source": f"profiles/{tag}_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, separate passes, summed over the trip launches of one frame)",
               "bytes_per_launch": int((fetch_b + write_b) / nonempty) if fetch_b and write_b else None,
               "nonempty_launches_per_frame": nonempty,
               "bytes_per_frame": int(fetch_b + write_b) if fetch_b and write_b else None,
               "fetch_MB_per_frame": round(fetch_b / 1e6, 1) if fetch_b else None, "write_MB_per_frame": round(write_b / 1e6, 1) if write_b else None,
               "fetch_MB_per_frame_uncorrected": round(fetch * 1024 / 1e6, 1) if fetch else None,
               "l2_hit_rate": round(hit / (hit + miss), 4) if hit and miss else None,
               "note": "fabric-side (L2 <-> Infinity Cache / HBM) traffic; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane reads on gfx950 (calibrated there on "
                       "coalesced streams; the 16-byte gathers here are the same instruction width), Infinity-Cache hits are counted, not excluded.  Algorithmic gather bytes per "
                       "frame = samples x 2060 B (~1.9 GB): the tables are L2 / Infinity-Cache resident, so fabric traffic is a fraction of the algorithmic stream"}
    rnd = tag.split("_")[0] if re.match(r"r\d\d", tag) else "r01"
    json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic_{precision}.json"), "w"), indent=1)
    print(json.dumps(traffic))


if __name__ == "__main__":
    main(*sys.argv[1:4])
