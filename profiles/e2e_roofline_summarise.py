#!/usr/bin/env python3
"""profiles/scripts/e2e_roofline.sh's summary step on its own: <collected dir> <repo root> <tag> -> JSON on stdout
(kernel-trace durations + SQ_INSTS_VALU per launch + the static VALU mix -> the valu_issue roofline of each host-level call's dominant kernel)."""
import csv, glob, json, os, sqlite3, sys
from collections import defaultdict
out, root, tag = sys.argv[1:4]   # <collected directory> <repo root> <tag of profiles/<tag>/>
# the static VALU mix of the kernels' row loops: the newest table (profiles/scripts/valu_mix_table.sh; it follows the source)
mix_files = sorted(glob.glob(os.path.join(root, "profiles", "r0*", "r0*_valu_mix.json")))
mix_path = sys.argv[4] if len(sys.argv) > 4 else mix_files[-1]
mix = json.load(open(mix_path))
N_SIMD, CLOCK = 1024, 2.4e9
# The walkers are bound by a chain of DEPENDENT accesses, one per step of the walk (src/alignment.c:244-350: the next cell
# follows from this one's direction), not by issue or bandwidth.  Their bound: rounds of resident walks x steps x the latency of
# one step's access (MI355X_MICROARCH.md: ds_read ~50 cycles, global_load L2 hit ~200 cycles, HBM miss ~900 cycles at 2.4 GHz).
#   tile walker (one wave per walk, 64 x 64-byte tiles of directions in LDS): a step is an LDS read; a tile costs one miss.
#   lane walker (one lane per walk, the next byte asked for ahead): a step is a byte load that misses (the bytes left L2 long
#   ago: C5's share writes 2.9 GB of them), two in flight per lane.
LAT = {"lds": 50 / CLOCK, "l2": 200 / CLOCK, "hbm": 900 / CLOCK}
def walker_bound(kernel, launches_items, steps, blocked=False):
    """(bound seconds, description) of one launch of a walker kernel over `launches_items` walks of `steps` steps each."""
    if "tile" in kernel or "group" in kernel:
        # Round 6: the tile walkers are bound by the LINES their tiles pull in, not by a latency chain or by instructions (a form with
        # 3-4 vector instructions per walk and step instead of 16 scalar ones took the same time on row-major bytes; TCC_MISS / FETCH_SIZE
        # of C2's launch: 278 MB -- profiles/r06/r06_walkers.txt).
        #   row-major direction bytes (the multi-hit path): a tile is 64 row pieces of 64 bytes at a pitch of len_a + 1 bytes, each on
        #     ~1.5 lines of 128 B, used for 63 .. 126 steps: measured 1.27 lines per step;
        #   blocked (NW, best hit; round 6): a tile is 8 x 4 whole blocks = 32 lines, good for >= 48 steps: (steps / 48 + 1) x 32 lines
        #     per walk -- and when the moves go in place to pinned host memory the kernel's end also waits for those PCIe writes
        #     (~9 GB/s on 24-byte pieces: 60-65 us for C2's 10 000 walks whatever the walker; not in this bound).
        if blocked and ", 32>" in kernel:
            # (round 6, second half: tiles of 32 x 32 bytes = 4 x 2 blocks = 8 lines, good for >= 16 steps -- the anchor rounds down to blocks of 8 x 16 cells)
            lines = launches_items * (steps / 16.0 + 1.0) * 8.0
            how = f"hbm lines: {launches_items:.0f} walks x ({steps:.0f} / 16 + 1) tiles x 8 lines of 128 B (blocked direction bytes, 32 x 32-byte tiles) at 8 TB/s; the moves' PCIe writes (in place) come on top"
        elif blocked:
            lines = launches_items * (steps / 48.0 + 1.0) * 32.0
            how = f"hbm lines: {launches_items:.0f} walks x ({steps:.0f} / 48 + 1) tiles x 32 lines of 128 B (blocked direction bytes) at 8 TB/s; the moves' PCIe writes (in place) come on top"
        else:
            lines = launches_items * steps * 1.27
            how = f"hbm lines: {launches_items:.0f} walks x {steps:.0f} steps x 1.27 lines of 128 B per step (measured: TCC_MISS, profiles/r06/r06_walkers.txt) at 8 TB/s"
        return lines * 128 / 8.0e12, how
    per_walk = steps * LAT["hbm"] / 2.0
    return per_walk, f"{steps:.0f} steps x 900-cycle miss / 2 loads in flight per lane (every walk has a lane: one round)"
res = {}
pmc_all = {}   # every counter of the --pmc passes, per workload and kernel: mean per launch -> <out>/e2e_pmc_insts.json
for key in ("C2_10000", "C5_125000", "C3_10000", "C4_4000", "C3_10000_hits4", "C4_4000_hits4"):
    dur = defaultdict(list)
    f = os.path.join(out, key + "_kernel_trace.csv")
    if not os.path.exists(f):
        continue
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "sa::" in n:
            dur[n.split("(")[0].replace("void ", "").strip()].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    if not dur:
        continue
    insts = defaultdict(list)
    for db in glob.glob(os.path.join(out, key + ".pmc", "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        for name, cname, val in con.execute("select kernel_name, counter_name, value from counters_collection"):
            if "sa::" not in name:
                continue
            kn = name.split("(")[0].replace("void ", "").strip()
            pmc_all.setdefault(key, {}).setdefault(kn, {}).setdefault(cname, []).append(float(val))
            if cname == "SQ_INSTS_VALU":
                insts[kn].append(float(val))
    # walks per launch and steps per walk: from the driver's own line in the trace log ("walks W steps S") when it prints one
    steps = walks = None
    logf = os.path.join(out, key + ".trace.log")
    if os.path.exists(logf):
        import re
        m = re.search(r"walks (\d+) mean_steps ([0-9.]+)", open(logf).read())
        if m:
            walks, steps = int(m.group(1)), float(m.group(2))
    kernels = {}
    for k, v in dur.items():
        v = v[len(v) // 3:]          # the first call sizes buffers: later launches only
        e = {"launches_seen": len(v), "kernel_ms": sum(v) / len(v), "total_ms_share": None}
        iv = insts.get(k)
        if iv:
            e["instructions"] = sum(iv) / len(iv)
        m = next((x for x in mix if x["demangled"] in k.replace("sa::", "") and k.rstrip().endswith(x.get("ends", ""))), None)
        if m and iv:
            cpi = m["mix"]["cycles_per_valu_instruction"]
            e["cycles_per_instruction"] = cpi
            e["frac"] = e["instructions"] * cpi / (N_SIMD * CLOCK * e["kernel_ms"] * 1e-3)
        if "traceback" in k and steps:
            launches_per_call = max(1, round(len(dur[k]) / 6)) if key.startswith(("C2", "C5")) else max(1, round(len(dur[k]) / 3))
            b, how = walker_bound(k, walks / launches_per_call, steps, blocked="hits4" not in key)
            e["bound"] = "hbm_lines" if ("tile" in k or "group" in k) else "dependent_latency"
            e["bound_ms"] = b * 1e3
            e["frac"] = b * 1e3 / e["kernel_ms"]
            e["bound_model"] = how
        kernels[k] = e
    total = sum(e["kernel_ms"] * e["launches_seen"] for e in kernels.values())
    for e in kernels.values():
        e["total_ms_share"] = round(e["kernel_ms"] * e["launches_seen"] / total, 3)
    dom = max(kernels, key=lambda k: kernels[k]["kernel_ms"] * kernels[k]["launches_seen"])
    d = kernels[dom]
    res[key.replace("_hits4", ":hits4").replace("_", ":", 1)] = {"bound": "valu_issue", "kernel": dom, "kernel_ms": round(d["kernel_ms"], 4),
                "instructions": d.get("instructions"), "cycles_per_instruction": d.get("cycles_per_instruction"),
                "frac": round(d["frac"], 3) if "frac" in d else None,
                "peak": "1024 SIMDs x 2.4 GHz; cycles per wave64 instruction by rate class (profiles/r03/r03_valu_rate_probe.txt) "
                        f"weighted by the static mix of the kernel's row loop ({os.path.relpath(mix_path, root)})",
                "source": f"profiles/{tag}/: rocprofv3 --kernel-trace (durations) and --pmc SQ_INSTS_VALU (own pass) over the host-level call",
                "kernels_of_the_call": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in e.items()} for k, e in kernels.items()}}
json.dump({k: {kn: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()} for kn, cs in ks.items()} for k, ks in pmc_all.items()},
          open(os.path.join(out, "e2e_pmc_insts.json"), "w"), indent=1)
res["_recorded"] = f"tag {tag}" + (f", round {int(tag[1:3])}" if tag[:1] == "r" and tag[1:3].isdigit() else "")
print(json.dumps(res, indent=1))
