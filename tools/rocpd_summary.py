#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as per-kernel stats, like --stats CSV.

    python tools/rocpd_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.md
"""
import re
import sqlite3
import sys


import shutil
import subprocess

_FILT = shutil.which("llvm-cxxfilt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
_DT = {"DF16b": "bf16", "f": "f32"}


def demangle(name: str) -> str:
    if not name.startswith("_Z"):
        return name
    m = re.match(r"_ZN2rc(\d+)(conv_mfma(?:_persist|_wsm|_ws)?_kernel)INS_7ConvCfgI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)E(?:Li\d+E)?EELb([01])E", name)
    if m:
        return f"rc::{m.group(2)}<{_DT[m.group(3)]},CK={m.group(4)},NT={m.group(5)},K={m.group(6)},gated={m.group(7)}>"
    m = re.match(r"_ZN2rc(\d+)(conv_mfma_auto(?:64)?_kernel)INS_7ConvCfgI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)E(?:Li\d+E)*EELi(\d)E", name)
    if m:       # kernels 6 / 7: mode 0 plain epilogues, 1 carried channel sums, 2 prefetched residual
        return f"rc::{m.group(2)}<{_DT[m.group(3)]},CK={m.group(4)},NT={m.group(5)},K={m.group(6)},mode={m.group(7)}>"
    m = re.match(r"_ZN2rc(\d+)(conv_mfma_wst_kernel)INS_7ConvCfgI(DF16b|f)Li(\d+)ELi(\d+)ELi(\d+)E(?:Li\d+E)?EELb([01])E", name)
    if m:       # kernel 4b (one barrier per stage): form 3 = tiles by LDS-DMA two stages ahead (<= 20 KB of weights a chunk), form 2 = tile through registers
        ck, nt, k = int(m.group(4)), int(m.group(5)), int(m.group(6))
        steps = (k * k * (ck // 8) + 3) // 4
        return f"rc::{m.group(2)}<{_DT[m.group(3)]},CK={ck},NT={nt},K={k},form={3 if steps * nt <= 20 else 2}>"
    m = re.match(r"_ZN2rc13conv32_kernelINS_6C32CfgILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)EEELb([01])ELb([01])E", name)
    if m:
        return f"rc::conv32_kernel<bf16,CK={m.group(1)},TH={m.group(2)},NCW={m.group(3)},NT32={m.group(4)},gated={m.group(5)},defer={m.group(6)}>"
    m = re.match(r"_ZN2rc(\d+)", name)          # generic rc::[ns::]<kernel><dtype, ints...> (c++filt cannot parse DF16b)
    if m:
        n = int(m.group(1)); start = m.end()
        base, rest = name[start:start + n], name[start + n:]
        m2 = re.match(r"(\d+)", rest)            # nested namespace (rc::gf::kernel): the next length-prefixed component is the kernel
        while m2:
            k = int(m2.group(1))
            base, rest = base + "::" + rest[m2.end():m2.end() + k], rest[m2.end() + k:]
            m2 = re.match(r"(\d+)", rest)
        args = []
        if rest.startswith("I"):
            rest = rest[1:]
            while rest and not rest.startswith("E"):
                for pat, fn in ((r"DF16b", lambda g: "bf16"), (r"f(?![a-z])", lambda g: "f32"), (r"Li(\d+)E", lambda g: g.group(1)),
                                (r"Lb([01])E", lambda g: g.group(1))):
                    mm = re.match(pat, rest)
                    if mm:
                        args.append(fn(mm)); rest = rest[mm.end():]
                        break
                else:
                    break
        return f"rc::{base}<{','.join(args)}>" if args else f"rc::{base}"
    return name


def short(name: str) -> str:
    name = demangle(name)
    name = re.sub(r"^void\s+", "", name)
    m = re.search(r"conv_mfma_kernel<rc::ConvCfg<([^>]*)>", name)
    if m:
        return "conv_mfma_kernel<" + m.group(1).replace(" ", "") + ">"
    name = re.sub(r"\(.*", "", name)
    return name if len(name) <= 90 else name[:87] + "..."


def main(path, last_forwards=0, marker="raw_ingest_kernel"):
    """last_forwards = N > 0: steady state only -- the dispatches from the N-th last launch of `marker` (a kernel every forward starts with) on, so the
    first forward's weight packing (H2D copies, bf16 -> f32 conversions) is out of the table."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    if last_forwards > 0:
        marks = [s for name, s, e in rows if marker in name]
        if len(marks) >= last_forwards:
            t0 = marks[-last_forwards]
            rows = [r for r in rows if r[1] >= t0]
            print(f"steady state: the last {last_forwards} forwards (from the {len(marks) - last_forwards + 1}-th of {len(marks)} `{marker}` launches on)\n")
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
    if "--last-forwards" in sys.argv:
        i = sys.argv.index("--last-forwards")
        main(sys.argv[1], int(sys.argv[i + 1]), *(sys.argv[i + 2:i + 3]))
    else:
        main(sys.argv[1])
