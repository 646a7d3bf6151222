#!/usr/bin/env python3
"""Is the level-0 48 -> 48 layer power / clock bound?  Runs the layer back to back for ~1.2 s per variant and samples rocm-smi (sclk, power) in the middle:
kernel 2 and kernel 6 as shipped, kernel 6 with its stores / MFMAs / tile loads knocked out (conv_flags 1 / 2 / 4), a plain copy, and a pure MFMA loop."""
import os, sys, time, subprocess, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops, _lib
L = _lib.load()
c = N.Conv2d(48, 48, 3, 1, 1).to("cuda", torch.bfloat16)
x = torch.rand(8, 1088, 1920, 48, device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)


def smi():
    """'sclk <MHz>, <W>' from one rocm-smi call (socket graphics package power; the reading averages over the tool's own window)."""
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
    sclk = power = "?"
    for l in r.stdout.splitlines():
        if "sclk" in l and "(" in l:
            sclk = l.split("(")[-1].split("Mhz")[0]
        elif "Power" in l and "===" not in l:
            power = l.split(":")[-1].strip()
    return f"sclk {sclk} MHz, {power} W"


def run(tag, fn, secs=1.2):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(50, int(secs * 1e3 / max(e0.elapsed_time(e1), 1e-3)))
    out = {}
    th = threading.Thread(target=lambda: (time.sleep(secs * 0.55), out.setdefault("smi", smi())))
    e0.record()
    th.start()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); th.join()
    print(f"{tag:34s} {e0.elapsed_time(e1) / n:7.3f} ms   {out.get('smi')}", flush=True)


def knobs(auto, flags):
    assert L.rc_debug_set(b"persist_auto", auto) == 0 and L.rc_debug_set(b"conv_flags", flags) == 0


print("idle:", smi())
with torch.no_grad():
    for tag, auto, flags in (("kernel 2", 0, 0), ("kernel 6", 2, 0), ("kernel 6, no stores", 2, 1), ("kernel 6, no MFMA", 2, 2), ("kernel 6, no tile loads", 2, 4),
                             ("kernel 6, no loads, no stores", 2, 5), ("kernel 6, no MFMA no loads", 2, 6)):
        knobs(auto, flags)
        run(tag, lambda: ops.conv2d(x, c, act="relu"))
    knobs(1, 0)
    # the multi-chunk kernel (wsm) on the two layer shapes that carry its time
    for cin, H, W in ((192, 544, 960), (512, 136, 240), (128, 272, 480)):
        cm = N.Conv2d(cin, cin, 3, 1, 1).to("cuda", torch.bfloat16)
        xm = torch.rand(8, H, W, cin, device="cuda").to(torch.bfloat16)
        fl = 2 * 8 * H * W * cin * cin * 9
        for tag, flags, thin in ((" kernel 4b", 0, 2), (" kernel 4", 0, 0), (" kernel 4, no MFMA", 2, 0), (" kernel 4, no stores", 1, 0)):     # the knock-outs exist in kernel 4 only
            knobs(1, flags)
            assert L.rc_debug_set(b"thin", thin) == 0
            run(f"{cin}->{cin} {H}x{W}{tag}", lambda: ops.conv2d(xm, cm, act="relu"))
        L.rc_debug_set(b"thin", 2)
        print(f"    ({fl / 1e12:.3f} TFLOP per launch)")
    knobs(1, 0)
    import ctypes as C
    for name, fn, wps in (("16x16x32 const", L.rc_debug_mfma_peak, 2), ("16x16x32 random", L.rc_debug_mfma_peak, 12), ("32x32x16 const", L.rc_debug_mfma_peak32, 2),
                          ("32x32x16 random", L.rc_debug_mfma_peak32, 12), ("32x32x16 random, 1 wave/SIMD", L.rc_debug_mfma_peak32, 11)):
        tf, tk = C.c_double(), C.c_double()
        out = {}
        out = []
        th = threading.Thread(target=lambda: [out.append((time.sleep(0.7), smi())[1]) for _ in range(3)])
        th.start()
        assert fn(wps, 1600000, C.byref(tf), C.byref(tk)) == 0      # three launches of ~0.7-1 s each: the first samples fall inside them
        th.join()
        print(f"pure MFMA {name:30s} {tf.value:7.1f} TF/s   {out}", flush=True)
    run("torch copy 1.6 -> 1.6 GB", lambda: y.copy_(x))
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16); b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    run("torch.matmul 8192^3 bf16", lambda: torch.matmul(a, b))
