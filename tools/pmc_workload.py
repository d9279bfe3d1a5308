#!/usr/bin/env python
"""Digest the passes of tools/pmc_workload.sh: counters of the LAST head launch (k_head_frame_persist / k_head_trip*) of every pass, its duration from the
counter-free trace of the same target, and the derived ratios -> gpurun_out/<tag>.json + <tag>.md (copy into profiles/ to commit; bench.py reads
profiles/r04_pmc_<variant>_<hw>_<precision>.json as roofline.pmc / roofline.traffic)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L2_PEAK_GBPS, L2_LINE = 34500.0, 128


def head(name):
    return "k_head_frame_persist" in name or "k_head_trip" in name


def main(tag, variant, hw, precision, group):
    src = os.path.join(ROOT, "gpurun_out")
    vals, kernel = {}, None
    for d in sorted(glob.glob(os.path.join(src, tag + "_p[0-9]"))):
        files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
        if not files:
            continue
        rows = [r for r in csv.DictReader(open(files[-1])) if head(r["Kernel_Name"])]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                vals[r["Counter_Name"]] = float(r["Counter_Value"])
                kernel = r["Kernel_Name"]
    dur_us, n_launch = None, 0
    for f in glob.glob(os.path.join(src, tag + "_ptrace", "**", "*kernel_trace.csv"), recursive=True):
        tr = [r for r in csv.DictReader(open(f)) if head(r["Kernel_Name"])]
        tr.sort(key=lambda r: int(r["Start_Timestamp"]))
        n_launch = len(tr)
        if tr:
            dur_us = (int(tr[-1]["End_Timestamp"]) - int(tr[-1]["Start_Timestamp"])) / 1e3
    g = vals.get
    out = {"workload": f"{variant} {hw}x{hw} {precision}, {group} frame(s) per head launch (tools/profile_clip.py: plain launches, one lane)", "kernel": kernel,
           "launch_us": dur_us, "head_launches_in_trace": n_launch, "counters_of_last_head_launch": vals}
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        # MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of 16-B-per-lane reads -> x 2
        out["fabric_bytes_per_launch"] = int(2 * g("FETCH_SIZE") * 1024 + g("WRITE_SIZE") * 1024)
        out["fetch_MB_uncorrected"], out["write_MB"] = round(g("FETCH_SIZE") * 1024 / 1e6, 1), round(g("WRITE_SIZE") * 1024 / 1e6, 1)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        req = g("TCC_HIT_sum") + g("TCC_MISS_sum")
        out["l2_hit_rate"] = round(g("TCC_HIT_sum") / max(req, 1), 4)
        out["l2_requests_per_launch"] = int(req)
        if dur_us:
            rate = req / (dur_us * 1e-6)
            out["l2_requests_per_s"] = rate
            out["l2_request_rate_frac"] = round(rate * L2_LINE / 1e9 / L2_PEAK_GBPS, 4)
            out["l2_request_rate_note"] = "TCC_HIT + TCC_MISS per launch / launch duration x one 128-B line per request, against the ~34.5 TB/s aggregate L2"
    if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
        out["valu_per_mfma"] = round(g("SQ_INSTS_VALU") / g("SQ_INSTS_MFMA"), 2)
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if g(k) is not None:
                out[k.lower() + "_frac_of_wave_cycles"] = round(g(k) / g("SQ_WAVE_CYCLES"), 4)
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
        # busy cycles summed over SIMDs (1024) / active cycles summed over XCDs (8)
        out["mfma_busy_frac"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / 1024.0 / (g("GRBM_GUI_ACTIVE") / 8.0), 4)
    if g("TCP_TOTAL_CACHE_ACCESSES_sum") is not None and g("SQ_INSTS_VMEM_RD"):
        out["l1_line_accesses_per_vmem_read"] = round(g("TCP_TOTAL_CACHE_ACCESSES_sum") / g("SQ_INSTS_VMEM_RD"), 2)
    json.dump(out, open(os.path.join(src, tag + ".json"), "w"), indent=1)
    md = [f"# counter passes -- {tag}", "", f"`tools/pmc_workload.sh {tag} {variant} {hw} {precision} {group}` on 1 x MI355X (gpurun): every counter group in its own rocprofv3 run "
          "(`--pmc <group> --kernel-trace`), values of the LAST head launch of the run; duration from a counter-free `--kernel-trace --stats` run of the same target.", "",
          "Units as rocprofv3 reports them: FETCH_SIZE / WRITE_SIZE in KiB of fabric-side traffic (gfx950: 16-B-per-lane reads are under-reported by 2 x, corrected in "
          "`fabric_bytes_per_launch`); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles summed over wavefronts; SQ_VALU_MFMA_BUSY_CYCLES summed over SIMDs; GRBM_GUI_ACTIVE over XCDs.", "",
          "```json", json.dumps(out, indent=1), "```"]
    open(os.path.join(src, tag + ".md"), "w").write("\n".join(md) + "\n")
    print(json.dumps({k: v for k, v in out.items() if k != "counters_of_last_head_launch"}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 1)
