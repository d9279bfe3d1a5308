#!/usr/bin/env python
"""Print the figures of a bench.py JSON line that a round's notes quote (value, check, roofline, modes, configs)."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d.get("roofline", {})
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "dtype", d["dtype"], "| data:", d["data"][:60])
print("  check", {k: v for k, v in d["config"].get("timed_frames_check", {}).items() if k in ("ok", "bytes_equal_per_frame_api", "psnr_vs_fp32_mode_db", "error")},
      "target psnr", d["config"].get("timed_frames_psnr_vs_analytic_target_db"), "ckpt parity", d["config"].get("ckpt_parity_fp32_vs_oracle"))
print("  roofline frac", r.get("frac"), "avg_launch_ms", r.get("avg_launch_ms"), "samples/frame", r.get("samples_per_frame"), "mfma", r.get("mfma", {}).get("frac"),
      "eff/period", r.get("effective_frac_per_frame_period"), "traffic", r.get("traffic"))
for k, v in d.get("modes", {}).items():
    print("  mode", k, v.get("value"), {a: b for a, b in v.items() if a in ("block_mean", "block_std", "error")})
for k, v in d.get("configs", {}).items():
    rr = v.get("roofline", {})
    print("  config", k, {a: b for a, b in v.items() if a in ("value", "latency_ms_p50", "psnr_vs_analytic_target_db", "torso_mask_share", "error", "rays_per_s")},
          "frac", rr.get("frac"), "launch_ms", rr.get("avg_launch_ms"), "samples/frame", rr.get("samples_per_frame"), "sr_us", v.get("sr_stage", {}).get("us_per_forward"))
if "cpu_baseline" in d:
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
