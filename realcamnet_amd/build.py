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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
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


def build(force: bool = False, verbose: bool = True) -> Path:
    srcs, deps = _deps()
    stamp = OBJ / "stamp.txt"
    dig = _digest(deps)
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == dig:
        return LIB
    OBJ.mkdir(exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: Path) -> Path:
        obj = OBJ / (src.stem + ".o")
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            sys.stderr.write(r.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
