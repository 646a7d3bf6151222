#!/usr/bin/env python3
"""Static scan of hipcc -S listings for the accumulator-reuse pattern that produced run-to-run differences on gfx950 (tools/ubench/gi_experiment.hip):
a v_mfma whose SrcC is the destination of a v_mfma issued a FEW instructions earlier with at least one other v_mfma in between (the back-to-back case, distance 1,
is interlocked by the hardware).  Reports, per kernel, every such pair with the number of MFMAs and of other instructions between them.
usage: mfma_hazard_scan.py file.s [max_mfma_between=3]      (scan(path, max_between) is what tests/test_abi_host.py imports)"""
import re, sys

_rng = re.compile(r"[va]\[(\d+):(\d+)\]")


def _regs(tok):
    m = _rng.match(tok)
    return (int(m.group(1)), int(m.group(2))) if m else None


def scan(path, maxb=3):
    """{kernel: [(mfmas between, other instructions between, the consuming instruction), ...]}"""
    kern, hist, found = None, [], {}
    for line in open(path):
        l = line.strip()
        mk = re.match(r"^(_Z\w+):", l)
        if mk:
            kern, hist = mk.group(1), []
            continue
        if kern is None or not l or l.startswith((";", ".", "//")):
            if l.startswith(".LBB"): hist = []          # conservative: a branch target starts a new window
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"):
            parts = [p.strip() for p in l[len(op):].split(",")]
            dst, srcc = _regs(parts[0]), _regs(parts[3]) if len(parts) > 3 else None
            if srcc is not None:
                between_m, between_o = 0, 0
                for (kind, d) in reversed(hist):
                    if kind == "m":
                        if d == srcc:
                            if 1 <= between_m <= maxb:
                                found.setdefault(kern, []).append((between_m, between_o, l))
                            break
                        between_m += 1
                        if between_m > maxb: break
                    else:
                        between_o += 1
            hist.append(("m", dst))
        elif op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
            hist = []
        else:
            n = 1
            m = re.match(r"s_nop\s+(\d+)", l)
            if m: n = int(m.group(1)) + 1
            hist.extend([("o", None)] * n)
    return found


if __name__ == "__main__":
    path = sys.argv[1]
    maxb = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    found, tot = scan(path, maxb), 0
    for k, v in found.items():
        by = {}
        for bm, bo, l in v: by[(bm, min(bo, 9))] = by.get((bm, min(bo, 9)), 0) + 1
        tot += len(v)
        print(f"{len(v):5d}  {k[:110]}  " + " ".join(f"[{bm} mfma + {bo}{'+' if bo == 9 else ''} other: {c}]" for (bm, bo), c in sorted(by.items())))
    print(f"{tot} suspicious pairs in {len(found)} kernels ({path})")
