#!/usr/bin/env python3
"""Static VALU instruction mix of a kernel's hot loop, for the `valu_issue` roofline of kernels that are bound by vector
instruction issue instead of HBM (the direction-byte fills, the sweep, the walkers).

    python valu_mix.py csrc/sa_fill_dirs_x2.hip fill_nw_dirs_x2_kernel 'ILi3E' [--loop largest|innermost]

Compiles the file for gfx950 to assembly (device side only), finds the kernel whose mangled name contains every given
substring, takes its hot loop (by default the backward branch that encloses the most vector ALU instructions -- the row loop) and counts its vector ALU instructions in the two rate
classes tools/probes/valu_rate_probe.hip measured on MI355X (profiles/r03/r03_valu_rate_probe.txt, cycles per wave64
instruction per SIMD at 2.4 GHz):
    half : 2.1-2.4 cycles  v_add_u32 / v_sub_u32 / v_and / v_or / v_xor / v_lshrrev / v_ashrrev / v_mov / v_bitop3 /
                           unpacked 16-bit add / max -- WITHOUT a DPP or SDWA modifier
    full : 4.1 cycles      everything else: max / max3 / cndmask / cmp / bfi / or3 / and_or / lshlrev / add3 / lshl_or /
                           every packed 16-bit op / every DPP form
Prints JSON: counts per class, the loop's weighted issue cycles per iteration and the average cycles per VALU instruction.
"""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
HALF = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_bitop3_b32", "v_add_u16", "v_max_i16", "v_not_b32", "v_add_co_u32", "v_sub_co_u32", "v_xnor_b32",
        "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_accvgpr_mov_b32"}
CYCLES = {"half": 2.15, "full": 4.1}


def kernel_body(asm: str, needles):
    lines = asm.splitlines()
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m and all(n in m.group(1) for n in needles) and start is None:
            start, name = i + 1, m.group(1)
    if start is None:
        raise SystemExit(f"no kernel matching {needles}")
    body = []
    for l in lines[start:]:
        if l.strip().startswith("s_endpgm"):
            break
        body.append(l)
    return name, body


def loops(body):
    """(first, last) instruction index of every loop: a conditional / unconditional branch to a label above it."""
    label_at, insts = {}, []
    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            label_at[m.group(1)] = len(insts)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        insts.append(t.split(";")[0].strip())
    out = []
    for i, t in enumerate(insts):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", t)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            out.append((label_at[m.group(1)], i))
    return insts, out


def classify(inst: str):
    op = inst.split()[0]
    if not op.startswith("v_"):
        return None
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    modified = any(k in inst for k in ("row_shr", "row_shl", "row_bcast", "wave_shr", "wave_shl", "quad_perm", "row_mask", "dpp", "sdwa", "row_ror", "row_share"))
    if base in HALF and not modified:
        return "half"
    if base.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "full"
    return "full"


def main():
    src, needles = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
    which = "largest"
    for a in sys.argv[2:]:
        if a.startswith("--loop="):
            which = a.split("=", 1)[1]
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "k.s"
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
               f"-I{ROOT / 'include'}", f"-I{ROOT / 'seq-align_amd' / 'host'}", f"-I{ROOT / 'seq-align_amd' / 'csrc'}",
               str(src), "-o", str(out)]
        subprocess.run(cmd, check=True)
        asm = out.read_text()
    name, body = kernel_body(asm, needles)
    insts, lps = loops(body)
    if not lps:
        raise SystemExit("no loop found")
    size = lambda lp: lp[1] - lp[0]
    if which == "innermost":
        lp = max((l for l in lps if not any(o != l and o[0] >= l[0] and o[1] <= l[1] for o in lps)), key=size)
    else:   # (the loop with the most vector ALU instructions: a setup loop full of scalar code and loads may be longer)
        lp = max(lps, key=lambda l: (sum(1 for t in insts[l[0]:l[1] + 1] if classify(t)), size(l)))
    counts = {"half": 0, "full": 0}
    other = {"salu": 0, "lds": 0, "vmem": 0}
    for t in insts[lp[0]:lp[1] + 1]:
        c = classify(t)
        if c:
            counts[c] += 1
        elif t.startswith("ds_"):
            other["lds"] += 1
        elif t.startswith(("global_", "buffer_", "flat_", "scratch_")):
            other["vmem"] += 1
        elif t.startswith("s_"):
            other["salu"] += 1
    n = counts["half"] + counts["full"]
    cycles = counts["half"] * CYCLES["half"] + counts["full"] * CYCLES["full"]
    print(json.dumps({"kernel": name, "loop_instructions": lp[1] - lp[0] + 1, "loops_in_kernel": len(lps), "valu": counts, **other,
                      "valu_issue_cycles_per_iteration": round(cycles, 1),
                      "cycles_per_valu_instruction": round(cycles / n, 3) if n else None,
                      "classes": "half = 2.15, full = 4.1 cycles per wave64 instruction per SIMD at 2.4 GHz (profiles/r03/r03_valu_rate_probe.txt)"}))


if __name__ == "__main__":
    main()
