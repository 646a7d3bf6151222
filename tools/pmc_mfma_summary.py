#!/usr/bin/env python3
"""Per-kernel sums of the counters collected by tools/pmc_mfma.sh (one rocprofv3 --pmc pass each), as a markdown table.
usage: pmc_mfma_summary.py TAG"""
import collections, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_summary import demangle
tag = sys.argv[1]
COUNTERS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
data, calls = collections.defaultdict(dict), collections.defaultdict(int)
for c in COUNTERS:
    path = f"gpurun_out/pmcm_{tag}_{c}/pmc_results.db"
    if not os.path.exists(path):
        continue
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    ix = {n: i for i, n in enumerate(cols)}
    kcol = "kernel_name" if "kernel_name" in ix else "name"
    seen = collections.defaultdict(set)
    for r in db.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != c:
            continue
        k = demangle(r[ix[kcol]])
        data[k][c] = data[k].get(c, 0.0) + r[ix["value"]]
        seen[k].add(r[ix["dispatch_id"]])
    for k, s in seen.items():
        calls[k] = max(calls[k], len(s))
print("| kernel | dispatches | MFMA busy cycles | CU busy cycles | MFMA busy / (4 x CU busy) | LDS bank-conflict cycles / LDS active cycles |")
print("|---|---|---|---|---|---|")
rows = sorted(data.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0.0))
for k, d in rows[:24]:
    m, b = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), d.get("SQ_BUSY_CU_CYCLES", 0.0)
    lc, la = d.get("SQ_LDS_BANK_CONFLICT", 0.0), d.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"| `{k[:90]}` | {calls[k]} | {m:.3g} | {b:.3g} | {m / (4 * b) if b else 0:.3f} | {lc / la if la else 0:.3f} |")
