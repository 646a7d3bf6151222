#!/usr/bin/env python3
"""pair2 (csrc/conv_pair.hip, weights in registers / two teams) against the first pair kernel and against two rc_conv2d launches:
steady-state time per 4K x 8 pair and the team phase stamps.  usage: pair2_probe.py [sums|gated|plain|film]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
L = ops.lib()
modes = sys.argv[1:] or ["sums", "gated"]
c1 = N.Conv2d(48, 48, 3, 1, 1).to("cuda", torch.bfloat16); c2 = N.Conv2d(48, 48, 3, 1, 1).to("cuda", torch.bfloat16)
x = torch.rand(8, 1088, 1920, 48, device="cuda").to(torch.bfloat16)
for mode in modes:
    kw = dict(want_sums=True) if mode in ("sums", "gated") else {}
    if mode == "gated":
        kw.update(gate=torch.rand(8, 48, device="cuda"), skip=torch.rand_like(x), store_input=True)
    film = (torch.rand(8, 48, device="cuda"), torch.rand(8, 48, device="cuda")) if mode == "film" else None
    def run():
        if mode == "film":
            return ops.conv_pair(x, c1, c2, act="leaky", slope=0.01, film=film, residual=x)
        return ops.conv_pair(x, c1, c2, act="relu", **kw)
    def run2():
        if mode == "film":
            t = ops.conv2d(x, c1, act="leaky", slope=0.01, film=film)
            return ops.conv2d(t, c2, residual=x)
        t = ops.conv2d(x, c1, act="relu", **{k: v for k, v in kw.items() if k != "want_sums"})
        t = t[0] if isinstance(t, tuple) else t
        return ops.conv2d(t, c2, want_sums="want_sums" in kw)
    res = {}
    for name, impl, f in (("two launches", 2, run2), ("pair (LDS weights)", 1, run), ("pair2 (register weights)", 2, run)):
        L.rc_debug_set(b"pair_impl", impl)
        for _ in range(30): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(60): f()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 60
    print(f"{mode}: " + "   ".join(f"{k} {v:.3f} ms" for k, v in res.items()), flush=True)
    dbg = torch.zeros(2048, dtype=torch.int64, device="cuda")
    L.rc_debug_set(b"pair_impl", 2)
    L.rc_debug_set_ptr(b"conv_phase_timing", dbg.data_ptr())
    run(); torch.cuda.synchronize()
    L.rc_debug_set_ptr(b"conv_phase_timing", None)
    d = dbg.cpu()
    a = d[:480].view(60, 8)[6:54, :4].float().mean(0); b = d[512:992].view(60, 8)[6:54, :4].float().mean(0)
    print("   team A cycles/tile: commit0+issue1+pass0 %d  commit1+issue0 %d  passes1,2 %d  barrier %d | sum %d" % (*a.tolist(), a.sum()))
    print("   team B cycles/tile: commit0+issue1+row0-mfma %d  commit1+issue0 %d  row0-epilogue+row1 %d  barrier %d | sum %d" % (*b.tolist(), b.sum()))
