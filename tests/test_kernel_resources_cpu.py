"""Registers, spills and scratch of every shipped kernel, as the compiler reported them for the build that ships (round-4 review, item 7: the numbers in DESIGN.md had
drifted from HEAD).  genefaceplusplus_amd/csrc/Makefile compiles every .hip with -Rpass-analysis=kernel-resource-usage and leaves the report in <file>.res; this test
reads those reports -- it never compiles on its own unless they are missing -- and pins

  * ZERO scratch in every instantiation the shipped May models run (tiled grids: the SLOW = false family of the 16-bit head kernels, both operand types, ambient
    D = 2 / 3, frame groups or not; the torso kernels; the prologue / resolve / conditioning / SR-final kernels), and
  * the stated ceilings of the rest (hash-grid SLOW = true family, the exact-fp32 parity kernels, two SR convolution instantiations), so that a change which
    makes them worse -- or DESIGN.md quoting something else -- fails here.
"""
import glob
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "genefaceplusplus_amd", "csrc")

# kernels that are allowed scratch: mangled-name pattern -> ceiling in bytes per lane (what HEAD compiles to; DESIGN.md section 8 quotes these)
CEILINGS = [
    # exact-fp32 parity mode (1 248 v_mfma_f32_32x32x2_f32 per block; launch-lifetime values parked around the block loop)
    (r"k_head_tripILi3E", 148), (r"k_head_trip_wILi3E", 36), (r"k_head_trip_wpILi3E", 36),
    (r"k_head_frame_persistILi3EfLb0ELb0ELb0E", 148), (r"k_head_frame_persistILi2EfLb0ELb0ELb0E", 36),
    # hash-addressed / true-modulo grids: the generic lookup on fp32 tables (SLOW = true), no shipped model
    (r"k_head_trip_poolILi[23]EDF16[b_]Lb1E", 136), (r"k_head_frame_persistILi[23]EDF16[b_]Lb1E", 120),
    (r"k_head_eval_lpILi[23]EDF16bLb1E", 204),         # the per-sample test entry of hash-grid models in the bf16 mode (two operand types + the generic lookup)
    # the profiling instantiation of the persistent launch (bench.py's roofline section only: gfpp_frame_ws.phase_cycles) and the trip-launch A/B partner of the
    # ambient-D-2 (audio-conditioned) models: 1-3 launch-lifetime values of the ingest step, none inside a block
    (r"k_head_frame_persistILi[23]EDF16[b_]Lb0ELb[01]ELb1E", 16), (r"k_head_trip_poolILi2EDF16[b_]Lb0E", 12),
    # super-resolution: the two K-sliced convolution instantiations without the fused first layer
    (r"k_sr_conv3ILi128ELi4ELi[12]ELi1ELi2ELb0E", 16),
    # the polyphase up-sampling layer at the 128 registers that two workgroups per CU allow: four launch-lifetime values (tile origin, thread coordinates) parked
    # across the products -- three scratch stores before the first MFMA, three loads behind the last barrier of a K slice, none inside an MFMA run; its
    # phase-clock twin (tools/sr_up_phases.py only) parks the stamp address too
    (r"k_sr_up_polyILb0E", 20), (r"k_sr_up_polyILb1E", 52),
]


def _reports():
    files = sorted(glob.glob(os.path.join(CSRC, "*.res")))
    hips = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    stale = [h for h in hips if not os.path.exists(h[:-4] + ".res") or os.path.getmtime(h[:-4] + ".res") < os.path.getmtime(h)]
    if stale:
        subprocess.check_call(["make", "-s", "-C", CSRC, "-j8"] + (["-B"] if len(files) < len(hips) else []))
        files = sorted(glob.glob(os.path.join(CSRC, "*.res")))
    out = {}
    for f in files:
        lines = [l for l in open(f).read().splitlines() if l.strip()]
        i = 0
        while i < len(lines):
            assert lines[i].startswith("Function Name:"), (f, lines[i])
            name = lines[i].split(":", 1)[1].strip()
            rec = {}
            i += 1
            while i < len(lines) and not lines[i].startswith("Function Name:"):
                k, v = lines[i].split(":", 1)
                rec[k.strip()] = v.strip()
                i += 1
            out[name] = {"file": os.path.basename(f), "vgprs": int(rec["VGPRs"]), "scratch": int(rec["ScratchSize [bytes/lane]"]), "vgpr_spill": int(rec["VGPRs Spill"]),
                         "occupancy": int(rec["Occupancy [waves/SIMD]"]), "lds": int(rec["LDS Size [bytes/block]"])}
    return out


def test_every_source_has_a_report_and_the_head_kernels_are_in_it():
    rep = _reports()
    assert {r["file"] for r in rep.values()} >= {os.path.basename(h)[:-4] + ".res" for h in glob.glob(os.path.join(CSRC, "*.hip"))}
    persist = [n for n in rep if "k_head_frame_persist" in n]
    assert len(persist) == 26, persist                 # AMB_D 2/3 x {f16, bf16} x SLOW x MF (+ the profiling twin of the tiled-grid ones) + two fp32
    assert any("k_torso_mlp_group" in n for n in rep) and any("k_torso_compose_group" in n for n in rep) and any("k_group_begin" in n for n in rep)


def test_shipped_may_instantiations_use_no_scratch():
    rep = _reports()
    # what the shipped models run: the persistent launch without the phase clocks (tiled grids, both operand types, ambient D 2 / 3, frame groups or not), the
    # trip-launch A/B partner of the May models (ambient D 3) and the per-sample entry
    may = [n for n in rep if re.search(r"k_head_frame_persistILi[23]EDF16[b_]Lb0ELb[01]ELb0E", n) or re.search(r"k_head_trip_poolILi3EDF16[b_]Lb0E", n)
           or re.search(r"k_head_eval_lpILi[23]EDF16[b_]Lb0E", n)]
    assert len(may) == 8 + 4 + 4, sorted(may)
    for n in may:
        r = rep[n]
        assert r["scratch"] == 0 and r["vgpr_spill"] == 0, (n, r)
        assert r["vgprs"] <= 256 and r["occupancy"] == 2 and r["lds"] <= 163840, (n, r)         # two wavefronts per SIMD, one workgroup per CU
    for n, r in rep.items():
        if "k_torso" in n or "k_group_begin" in n or "k_begin_premarch" in n or "budget_resolve" in n or "k_cond_feat" in n or "k_sr_final" in n or "k_clip" in n:
            assert r["scratch"] == 0, (n, r)
    group = [r for n, r in rep.items() if "k_torso_mlp_group" in n]
    assert group and all(r["vgprs"] <= 168 and r["occupancy"] >= 3 for r in group), group      # three wavefronts per SIMD (launch bounds), no spill


def test_every_other_kernel_is_within_its_stated_ceiling():
    rep = _reports()
    over, unlisted = [], []
    for n, r in rep.items():
        if r["scratch"] == 0:
            continue
        caps = [c for pat, c in CEILINGS if re.search(pat, n)]
        if not caps:
            unlisted.append((n, r["scratch"]))
        elif r["scratch"] > max(caps):
            over.append((n, r["scratch"], max(caps)))
    assert not unlisted, f"kernels with scratch that no ceiling names (state them in DESIGN.md and here): {unlisted}"
    assert not over, over
