#!/usr/bin/env python3
"""List kernels that use scratch memory or spill registers (hipcc -S of each translation unit, gfx950).
A conv kernel whose accumulators land in scratch still passes every parity test and silently runs 20-50 % slower
(round 1: the fp32 fast-epilogue switch), so run this after touching conv_kernel.hpp.
usage: python tools/check_scratch.py [file.hip ...]   (default: every .hip under realcamnet_amd/csrc)"""
import concurrent.futures as cf, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "realcamnet_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from rocpd_summary import demangle
files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))

def scan(f):
    with tempfile.NamedTemporaryFile(suffix=".s") as t:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", CSRC, "-S",
                            "--cuda-device-only", "-o", t.name, os.path.join(CSRC, os.path.basename(f))], capture_output=True, text=True)
        if r.returncode != 0:
            return [(f, "COMPILE ERROR " + r.stderr[-300:], 0, 0, 0)]
        txt = open(t.name).read()
    out = []
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:\s+\.\w+:.*\n)+)", txt):
        blk = m.group(2)
        def g(k):
            mm = re.search(rf"\.{k}:\s+(\d+)", blk)
            return int(mm.group(1)) if mm else 0
        if g("private_segment_fixed_size") or g("vgpr_spill_count"):
            out.append((f, demangle(m.group(1)), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("vgpr_count")))
    return out

with cf.ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
    rows = [r for rs in ex.map(scan, files) for r in rs]
for f, k, scratch, spills, vgprs in rows:
    print(f"{os.path.basename(f):32s} scratch {scratch:5} B  vgpr spills {spills:4}  vgprs {vgprs:4}  {str(k)[:110]}")
print(f"{len(rows)} kernels with scratch / spills")
