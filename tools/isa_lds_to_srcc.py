"""Static scan (no GPU): which kernels feed an MFMA's SrcC -- the accumulator it adds to -- straight from an LDS read, i.e. `ds_read* X; ...; s_waitcnt lgkmcnt;
v_mfma D, A, B, X` with no vector-ALU instruction writing X in between?  That is the one pattern the round-6 interference hunt points at in the victim kernels
(docs/LAB_NOTEBOOK.md: a bias vector read from LDS into the accumulator registers of `k_torso_lp`).  Compiles the given sources to gfx950 assembly with the
library's flags and prints, per kernel, how many MFMAs do that, into which register file, and the shortest distance in instructions between the read and the MFMA.

    python tools/isa_lds_to_srcc.py [file.hip ...]            (default: every .hip of genefaceplusplus_amd/csrc)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CSRC = os.path.join(ROOT, "genefaceplusplus_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-S", "--cuda-device-only", "-w"]


def regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b", tok):
        if m.group(1):
            out.update(m.group(1) + str(r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add(m.group(4) + m.group(5))
    return out


def scan(asm):
    """-> {kernel: (mfmas, mfmas with SrcC from an LDS read, of those into AGPRs, shortest distance)}"""
    out, name = {}, None
    for line in asm.splitlines():
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            name, from_lds, idx, rec = m.group(1), {}, 0, [0, 0, 0, None]
            out[name] = rec
            continue
        if name is None or not s or s[0] in ";." or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        idx += 1
        op = s.split()[0]
        ops = s[len(op):].split(",")
        if op == "s_endpgm":
            name = None
        elif op.startswith("ds_read"):
            for r in regs(ops[0]):
                from_lds[r] = idx
        elif op.startswith("v_mfma"):
            rec[0] += 1
            hit = [r for r in (regs(ops[3]) if len(ops) > 3 else ()) if r in from_lds]
            if hit:
                rec[1] += 1
                rec[2] += hit[0][0] == "a"
                d = idx - max(from_lds[r] for r in hit)
                rec[3] = d if rec[3] is None else min(rec[3], d)
            for r in regs(ops[0]):
                from_lds.pop(r, None)
        elif op.startswith(("v_", "global_load", "flat_load", "scratch_load", "buffer_load")):
            for r in regs(ops[0]):
                from_lds.pop(r, None)
    return out


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print("| kernel | MFMAs | SrcC straight from an LDS read | of those into AGPRs | shortest read -> MFMA distance (instructions) |\n|---|---|---|---|---|")
    for f in files:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            subprocess.check_call(["hipcc"] + FLAGS + ["-o", tmp.name, f], cwd=CSRC)
            res = scan(open(tmp.name).read())
        for k, (n, hit, agpr, dist) in sorted(res.items()):
            if hit:
                short = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("gfpp::", "").split("(")[0]
                print(f"| `{short}` | {n} | {hit} | {agpr} | {dist} |")


main()
