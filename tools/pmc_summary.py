#!/usr/bin/env python3
"""Per-kernel PMC totals from a rocprofv3 rocpd database: python tools/pmc_summary.py <db> [kernel-substring]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); sub = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
rows = db.execute("select * from counters_collection").fetchall()
ix = {c: i for i, c in enumerate(cols)}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
seen = set()
for r in rows:
    k = r[ix["kernel_name"]] if "kernel_name" in ix else r[ix["name"]]
    if sub not in k: continue
    agg[k][r[ix["counter_name"]]] += r[ix["value"]]
    did = r[ix["dispatch_id"]]
    if (k, did) not in seen:
        seen.add((k, did)); cnt[k] += 1
for k, d in agg.items():
    print(k[:110], "dispatches", cnt[k])
    for c, v in sorted(d.items()):
        print(f"   {c:32s} {v / cnt[k]:16.1f} per dispatch")
