"""Build librealcam_hip.so (hipcc, gfx950 only) in-tree.  No CUDA, no hipify, no multi-arch."""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = PKG / "_build"
LIB = PKG / "librealcam_hip.so"
SOURCES = ["lib.hip", "conv.hip", "conv_dispatch.hip", "conv_pair.hip", "chain.hip", "pointwise.hip", "cond.hip", "gma.hip", "gma_fused.hip", "wmsa.hip", "entropy.hip", "rans.hip"] + sorted(p.name for p in CSRC.glob("conv_inst_*.hip")) + sorted(p.name for p in CSRC.glob("conv32_inst_*.hip"))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Rpass-analysis=kernel-resource-usage"]          # per-kernel registers / scratch / spills -> _build/resources.json (tests/test_abi_host.py holds the bench path to zero)
FLAGS += os.environ.get("RC_EXTRA_HIPCC_FLAGS", "").split()   # experiments only


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: librealcam_hip.so cannot be built on this machine")
    return exe


def _digest(paths) -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in sorted(paths):
        h.update(p.name.encode())
        h.update(p.read_bytes())
    return h.hexdigest()


def _deps():
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    return srcs, srcs + sorted(CSRC.glob("*.hpp")) + [PKG.parent / "include" / "realcam_hip.h"]


def source_digest() -> str:
    """SHA-256 over the compile flags and every kernel source / header: identifies the code a measurement was taken on
    (profiles/*_pmc_bench.json carries it; bench.py reports PMC traffic only for a matching digest)."""
    return _digest(_deps()[1])


def _parse_resource_remarks(stderr: str, tu: str, into: dict) -> str:
    """Collect hipcc's kernel-resource-usage remarks of one translation unit into `into` (mangled kernel name -> counts) and return the
    rest of stderr (warnings and their notes) for display."""
    import re
    names = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
             "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
    blocks, cur_block = [], None                      # a diagnostic = its "file:line:col: kind:" line + the lines that follow it
    pending = []                                      # "In file included from" lines in front of the next diagnostic
    for line in stderr.splitlines(keepends=True):
        if re.match(r"\S+:\d+:\d+: (warning|error|remark|note|fatal error):", line):
            cur_block = {"kind": line.split(": ")[1].split(":")[0] if ": " in line else "", "lines": pending + [line]}
            m = re.match(r"\S+:\d+:\d+: (\w+(?: error)?):", line)
            cur_block["kind"] = m.group(1)
            pending = []
            blocks.append(cur_block)
        elif line.startswith("In file included from"):
            pending.append(line)
            cur_block = None
        elif cur_block is not None:
            cur_block["lines"].append(line)
    cur = None
    keep = []
    last_kept = False
    for blk in blocks:
        head = blk["lines"][-1] if False else [l for l in blk["lines"] if not l.startswith("In file included from")][0]
        if blk["kind"] == "remark":
            m = re.search(r"remark: +([^:]+): *(\S+)", head)
            if m:
                k, v = m.group(1).strip(), m.group(2)
                if k == "Function Name":
                    cur = into.setdefault(v, {"tu": tu})
                elif cur is not None and k in names:
                    cur[names[k]] = int(v)
            last_kept = False
        elif blk["kind"] == "note":
            if last_kept:
                keep.extend(blk["lines"])
        else:
            keep.extend(blk["lines"])
            last_kept = True
    return "".join(keep)


def kernel_resources() -> dict:
    """{mangled kernel name: {tu, vgprs, agprs, sgprs, scratch, vgpr_spill, sgpr_spill, occupancy, lds}} of the built library (build() first)."""
    import json
    build(verbose=False)
    f = OBJ / "resources.json"
    if not f.exists():
        build(force=True, verbose=False)
    return json.loads(f.read_text())


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs, deps = _deps()
    stamp = OBJ / "stamp.txt"
    dig = _digest(deps)
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == dig:
        return LIB
    OBJ.mkdir(exist_ok=True)
    hipcc = _hipcc()

    resources = {}

    def compile_one(src: Path) -> Path:
        obj = OBJ / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        rest = _parse_resource_remarks(r.stderr, src.name, resources)
        if verbose and rest.strip():
            sys.stderr.write(rest)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    import json
    (OBJ / "resources.json").write_text(json.dumps(resources, indent=0, sort_keys=True))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
