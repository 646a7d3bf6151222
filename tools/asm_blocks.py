#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S listing (static; for finding per-tile overhead).
usage: asm_blocks.py file.s <mangled-substring> [min_instrs]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l and not l.startswith("_ZN2rc.*\$"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
def cat(op):
    if "mfma" in op: return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"
blocks, cur, name = [], {}, "entry"
order = ["mfma", "valu", "salu", "lds", "vmem", "smem", "wait", "barrier", "branch", "nop"]
tot = {}
for i in range(start + 1, end):
    l = lines[i].strip()
    if not l or l.startswith((";", ".", "//")):
        if re.match(r"^\.LBB\d+_\d+:", l):
            blocks.append((name, cur)); name, cur = l.split(":")[0], {}
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); name, cur = m.group(1), {}
        continue
    op = l.split()[0]
    c = cat(op)
    cur[c] = cur.get(c, 0) + 1
    tot[c] = tot.get(c, 0) + 1
    if c == "branch":
        cur.setdefault("_tgt", []).append(l.split()[-1])
blocks.append((name, cur))
print("block".ljust(12) + "".join(k.rjust(8) for k in order) + "   branches")
for n, c in blocks:
    k = sum(v for kk, v in c.items() if kk != "_tgt")
    if k >= minn:
        print(n.ljust(12) + "".join(str(c.get(k2, 0)).rjust(8) for k2 in order) + "   " + ",".join(c.get("_tgt", [])))
print("TOTAL".ljust(12) + "".join(str(tot.get(k2, 0)).rjust(8) for k2 in order))
