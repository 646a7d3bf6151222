#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 PMC counters: one `rocprofv3 --pmc <group> --kernel-trace` pass per counter group over the command given
(never combined with other trace domains), then one markdown table (counter value per dispatch).
usage: tools/pmc_any.py TAG "C1 C2 C3;C4 C5" -- <command ...>"""
import collections, os, sqlite3, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from rocpd_summary import demangle

tag, groups = sys.argv[1], [g.split() for g in sys.argv[2].split(";")]
cmd = sys.argv[sys.argv.index("--") + 1:]
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
data, calls = collections.defaultdict(dict), collections.defaultdict(int)
for gi, grp in enumerate(groups):
    out = os.path.join(root, f"gpurun_out/pmca_{tag}_{gi}")
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--pmc", *grp, "--kernel-trace", "-d", out, "-o", "pmc", "--", *cmd], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        print(f"group {grp}: rocprofv3 rc {r.returncode}\n{r.stderr[-2000:]}")
        continue
    path = None
    for dp, _, fs in os.walk(out):
        for f in fs:
            if f.endswith("_results.db"):
                path = os.path.join(dp, f)
    if path is None:
        print(f"group {grp}: no results db under {out}")
        continue
    db = sqlite3.connect(path)
    cols = [r_[1] for r_ in db.execute("pragma table_info(counters_collection)")]
    ix = {n: i for i, n in enumerate(cols)}
    kcol = "kernel_name" if "kernel_name" in ix else "name"
    seen = collections.defaultdict(set)
    for r_ in db.execute("select * from counters_collection"):
        c = r_[ix["counter_name"]]
        if c not in grp:
            continue
        k = demangle(r_[ix[kcol]])
        data[k][c] = data[k].get(c, 0.0) + r_[ix["value"]]
        seen[k].add(r_[ix["dispatch_id"]])
    for k, s in seen.items():
        calls[k] = max(calls[k], len(s))
allc = [c for g in groups for c in g]
print("| kernel | dispatches | " + " | ".join(allc) + " |")
print("|---|---|" + "---|" * len(allc))
for k, d in sorted(data.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    n = max(calls[k], 1)
    print(f"| `{k[:80]}` | {calls[k]} | " + " | ".join(f"{d.get(c, 0.0) / n:.4g}" for c in allc) + " |")
