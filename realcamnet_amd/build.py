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
SOURCES = ["lib.hip", "conv.hip", "conv_dispatch.hip", "conv_pair.hip", "wino.hip", "chain.hip", "pointwise.hip", "cond.hip", "gma.hip", "gma_fused.hip", "wmsa.hip", "entropy.hip", "rans.hip"] + sorted(p.name for p in CSRC.glob("conv_inst_*.hip")) + sorted(p.name for p in CSRC.glob("conv32_inst_*.hip"))
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
    """Collect hipcc's kernel-resource-usage remarks of one translation unit into `into` (mangled kernel name -> counts) and return
    everything else on stderr unchanged (warnings with their notes and source excerpts, driver / linker lines, "N warnings generated")."""
    import re
    names = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
             "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
    diag = re.compile(r"\S+:\d+:\d+: (warning|error|remark|note|fatal error):")
    keep, pending = [], []            # pending: "In file included from" lines in front of the next diagnostic
    in_remark = False                 # the lines after a remark's head (source excerpt, caret) belong to it
    cur = None
    for line in stderr.splitlines(keepends=True):
        m = diag.match(line)
        if m:
            in_remark = m.group(1) == "remark"
            if in_remark:
                pending = []
                r = re.search(r"remark: +([^:]+): *(\S+)", line)
                if r:
                    k, v = r.group(1).strip(), r.group(2)
                    if k == "Function Name":
                        cur = into.setdefault(v, {"tu": tu})
                    elif cur is not None and k in names:
                        cur[names[k]] = int(v)
            else:
                keep.extend(pending + [line])
                pending = []
        elif line.startswith("In file included from"):
            pending.append(line)
            in_remark = False
        elif not in_remark:
            keep.extend(pending + [line])
            pending = []
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
    """Compile what changed and link.  A translation unit is recompiled when its own source, any csrc/*.hpp, the public header or the flags
    changed (per-object stamps under _build/); the library is relinked when any object was."""
    import json
    srcs, deps = _deps()
    stamp = OBJ / "stamp.txt"
    dig = _digest(deps)
    if not force and LIB.exists() and stamp.exists() and stamp.read_text() == dig:
        return LIB
    OBJ.mkdir(exist_ok=True)
    hipcc = _hipcc()
    shared = [d for d in deps if d not in srcs]            # headers: every object depends on them
    res_file = OBJ / "resources.json"
    try:
        resources = json.loads(res_file.read_text()) if res_file.exists() and not force else {}
    except ValueError:
        resources = {}

    def compile_one(src: Path):
        obj = OBJ / (src.stem + ".o")
        ostamp = OBJ / (src.stem + ".stamp")
        odig = _digest([src] + shared)
        if not force and obj.exists() and ostamp.exists() and ostamp.read_text() == odig and any(v.get("tu") == src.name for v in resources.values()):
            return obj, None
        cmd = [hipcc, *FLAGS, "-c", str(src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        mine = {}
        rest = _parse_resource_remarks(r.stderr, src.name, mine)
        if verbose and rest.strip():
            sys.stderr.write(rest)
        ostamp.write_text(odig)
        return obj, mine

    with cf.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 8, len(srcs))) as ex:
        done = list(ex.map(compile_one, srcs))
    for src, (_, mine) in zip(srcs, done):
        if mine is not None:
            for k in [k for k, v in resources.items() if v.get("tu") == src.name]:
                del resources[k]
            resources.update(mine)
    live = {s.name for s in srcs}
    resources = {k: v for k, v in resources.items() if v.get("tu") in live}
    res_file.write_text(json.dumps(resources, indent=0, sort_keys=True))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *(str(o) for o, _ in done)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    stamp.write_text(dig)
    return LIB


def build_variant(name: str, defines, only=None) -> Path:
    """Kernel experiments: recompile `only` with extra -D flags and link them with the library's other objects into realcamnet_amd/_alt/lib_<name>.so
    (select it with RC_HIP_LIB=...; git-ignored, travels with gpurun)."""
    build(verbose=False)
    only = only or tuple(os.environ.get("RC_VARIANT_TUS", "wino.hip").split(","))
    alt = PKG / "_alt"
    alt.mkdir(exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for s_ in _deps()[0]:
        if s_.name in only:
            o = alt / f"{s_.stem}_{name}.o"
            r = subprocess.run([hipcc, *[f for f in FLAGS if not f.startswith("-Rpass")], *[f"-D{d}" for d in defines], "-c", str(s_), "-o", str(o)], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stderr)
            objs.append(o)
        else:
            objs.append(OBJ / (s_.stem + ".o"))
    out = alt / f"lib_{name}.so"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(out), *map(str, objs)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":          # python -m realcamnet_amd.build --variant NAME DEFINE [DEFINE ...]
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        print(build(force="--force" in sys.argv))
