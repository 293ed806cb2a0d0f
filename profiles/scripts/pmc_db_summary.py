#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3's rocpd SQLite output (the default format when --output-format is not given).
    python profiles/scripts/pmc_db_summary.py <dir with pmc*/..._results.db> [kernel-name filter ...]"""
import glob, json, os, sqlite3, sys
from collections import defaultdict

out = sys.argv[1]
filters = sys.argv[2:] or ["sweep", "fill_", "traceback", "sw_", "reduce"]
acc = defaultdict(lambda: defaultdict(list))
for db in glob.glob(os.path.join(out, "**", "*.db"), recursive=True):
    c = sqlite3.connect(db)
    try:
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        kn = "kernel_name" if "kernel_name" in cols else "name"
        # one row per (dispatch, counter, dimension instance): a dispatch's value is the sum over its instances
        for name, ctr, val in c.execute(f"select {kn}, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name"):
            acc[name.split("(")[0]][ctr].append(float(val))
    except sqlite3.Error as e:
        print("skip", db, e, file=sys.stderr)
res = {}
for k, cs in acc.items():
    if not any(f in k for f in filters):
        continue
    res[k] = {cn: {"launches": len(v), "mean": sum(v) / len(v)} for cn, v in sorted(cs.items())}
print(json.dumps(res, indent=1))
