#!/usr/bin/env python3
"""Build-time lint for the kernels that track their own loads (sa_sw_sweep.hip: SweepRow; sa_reduce.hip: the three-deep stream).

Those kernels request rows with inline-asm `global_load_*` and claim them later with an inline-asm `s_waitcnt vmcnt(N)`; the
compiler does not know the destination registers are still being written between the two.  That is sound only while it never
touches them in between -- no copy, no SPILL (round 5: with 64-bit keys and 8 columns per lane it spilled a buffer to scratch
right behind the load and stored garbage).  This script reads the device assembly (`hipcc -save-temps=obj`) and fails the
build when, in a kernel that has such loads,
  * the kernel uses scratch at all (.amdhsa_private_segment_fixed_size != 0), or
  * between an asm load and the next asm `s_waitcnt vmcnt` (text order) an instruction outside the asm blocks names one of
    the load's destination registers, or another asm load's destination overlaps it, or
  * an asm block's first load reads an SGPR that a VALU instruction (v_readlane / v_readfirstlane / v_cmp ... writing an SGPR)
    wrote within the five instructions in front of the block, and the block does not open with `s_nop 4` (gfx9: five wait
    states between a VALU write of an SGPR and a VMEM read of it; the compiler's hazard recogniser does not look into asm).
Usage: check_inflight_loads.py file.s [...]      (exit status 1 and one line per finding)
"""
import re
import sys


def regs_of(tok):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path):
    findings, kernels = [], 0
    lines = open(path).read().splitlines()
    # split into functions: "name:" at column 0 ... ".end_amdhsa_kernel" (helpers without a kernel descriptor are skipped)
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    for si, s in enumerate(starts):
        e = starts[si + 1] if si + 1 < len(starts) else len(lines)
        body = lines[s:e]
        name = body[0].split(":")[0]
        in_asm, loads = False, []
        tagged = []
        for l in body:
            t = l.split(";")[0].strip() if not l.strip().startswith(";;#") else l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            tagged.append((in_asm, l.split(";")[0].strip()))
        idx_loads = [i for i, (a, t) in enumerate(tagged) if a and t.startswith("global_load")]
        if not idx_loads:
            continue
        kernels += 1
        for l in body:
            m = re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", l)
            if m and int(m.group(1)) != 0:
                findings.append(f"{name}: {m.group(1)} bytes of scratch in a kernel with hand-tracked loads (spills)")
        # the SGPR hazard: per asm block
        i = 0
        while i < len(tagged):
            if not tagged[i][0]:
                i += 1
                continue
            j = i
            while j < len(tagged) and tagged[j][0]:
                j += 1
            block = [t for _, t in tagged[i:j] if t]
            if any(t.startswith("global_load") for t in block):
                nop_first = bool(block) and re.match(r"s_nop\s+([4-9]|1\d)", block[0]) is not None
                sregs = set()
                for t in block:
                    if t.startswith("global_load"):
                        for m in re.finditer(r"s\[(\d+):(\d+)\]", t):
                            sregs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                before = [t for a, t in tagged[max(0, i - 8):i] if not a and t and not t.startswith(".")][-5:]
                for t in before:
                    if not t.startswith("v_"):
                        continue
                    first = t.split(",")[0]
                    wr = set()
                    for m in re.finditer(r"\bs(\d+)\b", first.split(None, 1)[1] if " " in first else ""):
                        wr.add(int(m.group(1)))
                    for m in re.finditer(r"s\[(\d+):(\d+)\]", first):
                        wr |= set(range(int(m.group(1)), int(m.group(2)) + 1))
                    if wr & sregs and not nop_first:
                        findings.append(f"{name}: '{t}' writes an SGPR that the asm load behind it reads inside 5 wait states (no s_nop 4)")
            i = j
        for i in idx_loads:
            dest = regs_of(tagged[i][1].split(",")[0])
            for j in range(i + 1, len(tagged)):
                a, t = tagged[j]
                if not t or t.startswith("."):
                    continue
                if a and t.startswith("s_waitcnt") and "vmcnt" in t:
                    break
                if a and t.startswith("global_load"):
                    if regs_of(t.split(",")[0]) & dest:
                        findings.append(f"{name}: asm load '{t}' overwrites registers still in flight from '{tagged[i][1]}'")
                    continue
                if not a and regs_of(t) & dest:
                    findings.append(f"{name}: '{t}' touches registers in flight from '{tagged[i][1]}'")
                    break
    return kernels, findings


def main():
    bad = 0
    for path in sys.argv[1:]:
        kernels, findings = check(path)
        for f in findings:
            print(f"check_inflight_loads: {path}: {f}", file=sys.stderr)
        bad += len(findings)
        print(f"check_inflight_loads: {path}: {kernels} kernels with hand-tracked loads, {len(findings)} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
