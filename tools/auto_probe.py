#!/usr/bin/env python3
"""The wave-autonomous kernels (6: 48 / 32 channels, 7: 64 channels; rc_debug_set("persist_auto", ...)) against the kernels they replace, on C -> C layers
(argv[1] = C, default 48) at the cfg3 sizes: bit-equality of every operand form, then ms per layer (>= 50 warm-up launches: the GPU idles at low clocks) and the RCAGroup."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops, _lib
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
L = _lib.load()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 48


VARIANTS = (("persist_auto 0", 0, 0), ("persist_auto 2", 2, 0))     # 48 / 32 channels: kernel 2 vs kernel 6; 64 channels: the general kernel vs kernel 7


def knob(v, flags=0):
    assert L.rc_debug_set(b"persist_auto", v) == 0 and L.rc_debug_set(b"conv_flags", flags) == 0


def timed(fn, n=20, warm=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    conv = N.Conv2d(C, C, 3, 1, 1).to(dev, bf).eval()
    # ---- parity: ragged and multi-image shapes, every form, kernel 6 == kernel 2 bit for bit
    bad = 0
    for (B, H, W) in ((1, 8, 32), (2, 23, 70), (3, 40, 97), (2, 64, 64), (9, 17, 33), (1, 130, 260)):
        x = torch.randn(B, H, W, C, device=dev, dtype=bf)
        res = torch.randn(B, H, W, C, device=dev, dtype=bf)
        g = torch.rand(B, C, device=dev, dtype=torch.float32)
        forms = {"plain": dict(), "relu": dict(act="relu"), "leaky": dict(act="leaky", slope=0.2), "sums": dict(want_sums=True), "relu+sums": dict(act="relu", want_sums=True),
                 "leaky+sums": dict(act="leaky", slope=0.01, want_sums=True),
                 "res": dict(residual=res), "gate+res": dict(residual=res, out_scale=g)}
        for name, kw in forms.items():
            outs = []
            for _, v, fl in VARIANTS:
                knob(v, fl)
                o = conv._nhwc(x, **kw)
                outs.append(o if isinstance(o, tuple) else (o,))
            torch.cuda.synchronize()
            for i in range(1, len(VARIANTS)):
                ok = all(torch.equal(a, b) for a, b in zip(outs[0], outs[i]))
                if "sums" in name:          # slots are written differently (carried runs): compare the fold
                    ok = torch.equal(outs[0][0], outs[i][0]) and torch.allclose(outs[0][1].sum(1), outs[i][1].sum(1), rtol=1e-5, atol=1e-3)
                if not ok:
                    bad += 1
                    d = (outs[0][0].float() - outs[i][0].float()).abs().max().item()
                    print(f"MISMATCH {B}x{H}x{W} {name} {VARIANTS[i][0]}: max |diff| {d}")
    print(f"parity: {'all forms bit-identical' if bad == 0 else str(bad) + ' MISMATCHES'}")

    # ---- timing
    for (B, H, W) in ((8, 1088, 1920), (8, 544, 960)):
        x = torch.randn(B, H, W, C, device=dev, dtype=bf)
        res = torch.randn(B, H, W, C, device=dev, dtype=bf)
        g = torch.rand(B, C, device=dev, dtype=torch.float32)
        gb = B * H * W * C * 2 / 1e9
        for name, kw, maps in (("relu", dict(act="relu"), 2), ("relu+sums", dict(act="relu", want_sums=True), 2), ("res", dict(residual=res), 3), ("gate+res", dict(residual=res, out_scale=g), 3)):
            t = []
            for _, v, fl in VARIANTS:
                knob(v, fl)
                t.append(timed(lambda: conv._nhwc(x, **kw)))
            print(f"{B}x{H}x{W} {C}->{C} {name:10s}: " + "   ".join(f"{VARIANTS[i][0]} {t[i]:.3f}" for i in range(len(VARIANTS))) + f"   best x{t[0] / min(t[1:]):.3f}")
        if C == 48:
            rg = N.RCAGroup(48, 48, nb=4).to(dev, bf).eval()
            t = []
            for _, v, fl in VARIANTS:
                knob(v, fl)
                t.append(timed(lambda: rg._nhwc(x), n=8, warm=10))
            print(f"{B}x{H}x{W} RCAGroup: " + "   ".join(f"{VARIANTS[i][0]} {t[i]:.3f}" for i in range(len(VARIANTS))) + f"   best x{t[0] / min(t[1:]):.3f}")
    knob(0)
