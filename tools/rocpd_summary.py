#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as per-kernel stats, like --stats CSV.

    python tools/rocpd_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"^void\s+", "", name)
    m = re.search(r"conv_mfma_kernel<rc::ConvCfg<([^>]*)>", name)
    if m:
        return "conv_mfma_kernel<" + m.group(1).replace(" ", "") + ">"
    name = re.sub(r"\(.*", "", name)
    return name if len(name) <= 90 else name[:87] + "..."


def main(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {namecol}, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0, 10**18, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    print(f"| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / total:.1f} |")
    print(f"\ntotal kernel time {total / 1e6:.3f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1])
