#!/usr/bin/env python3
"""GPU probe of the 32x32x16 conv (conv32_kernel.hpp): exact-integer parity in every operand form, then layer timings against the
16x16x32 kernels at the flagship shapes.  usage: python tools/conv32_probe.py [--no-time] [--no-check]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from realcamnet_amd import networks as N, ops

L = ops.lib()
dev, bf = "cuda", torch.bfloat16


def setk(conv32=4, persist=1, flags=0):
    L.rc_debug_set(b"conv32", conv32); L.rc_debug_set(b"persist", persist); L.rc_debug_set(b"conv_flags", flags)


def mk(cin, cout, g, dense=96.0):
    c = N.Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        sparse = (torch.rand(c.weight.shape, generator=g) < dense / cin).float()
        c.weight.copy_(torch.randint(-1, 2, c.weight.shape, generator=g).float() * sparse)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    return c


def report(tag, y, ref):
    if torch.equal(y, ref):
        print(f"  ok    {tag}")
        return True
    bad = (y != ref)
    nb = int(bad.sum())
    idx = bad.nonzero()[:6].tolist()
    ch = bad.any(dim=0).any(dim=-1).any(dim=-1) if bad.dim() == 4 else None
    print(f"  FAIL  {tag}: {nb}/{bad.numel()} differ, max|d| {float((y - ref).abs().max())}, first {idx}")
    if bad.dim() == 4:
        print(f"        bad channels {bad.any(dim=0).flatten(1).any(1).nonzero().flatten().tolist()[:40]}")
        print(f"        bad rows {bad.any(dim=0).any(dim=0).any(dim=1).nonzero().flatten().tolist()[:40]}")
        print(f"        bad cols {bad.any(dim=0).any(dim=0).any(dim=0).nonzero().flatten().tolist()[:40]}")
    return False


def check():
    ok = True
    for variant in (1, 2, 3):
        for shape in [(128, 64, 16, 40), (192, 192, 9, 33), (512, 128, 8, 32), (128, 128, 37, 100), (48, 192, 16, 40), (48, 96, 21, 70)]:
            cin, cout, h, w = shape
            if cin == 48 and variant != 1:
                continue
            for mode in ("plain", "relu", "gated", "res", "sums", "film", "ps", "relu_post", "gelu"):
                if mode == "ps" and (cout // 4) % 16 != 0:
                    continue
                g = torch.Generator().manual_seed(cin * 7 + cout)
                c = mk(cin, cout, g)
                x = torch.randint(-1, 2, (2, cin, h, w), generator=g).float()
                xin, kw = x, {}
                wt, bs = c.weight.detach().clone(), c.bias.detach().clone()
                if mode == "gated":
                    r = torch.randint(-1, 2, (2, cin, h, w), generator=g).float()
                    gate = torch.randint(0, 2, (2, cin), generator=g).float()
                    xin = r * gate[:, :, None, None] + x
                ref = F.conv2d(xin, wt, bs, padding=1)
                if mode == "relu":
                    ref = ref.relu(); kw = dict(act="relu")
                if mode == "gelu":
                    kw = dict(act="gelu")
                res = None
                if mode in ("res", "relu_post"):
                    res = torch.randint(-2, 3, (2, cout, h, w), generator=g).float()
                    ref = ref + res
                    if mode == "relu_post":
                        ref = ref.relu(); kw = dict(act="relu_post")
                if mode == "film":
                    fs = torch.randint(0, 2, (2, cout), generator=g).float(); ft = torch.randint(-1, 2, (2, cout), generator=g).float()
                    ref = ref * fs[:, :, None, None] + ft[:, :, None, None] + ref
                    ref = torch.where(ref > 0, ref, ref * 0.5); kw = dict(act="leaky", slope=0.5, film=(fs.to(dev), ft.to(dev)))
                if mode == "ps":
                    ref = F.pixel_shuffle(ref, 2); kw = dict(out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)
                setk(conv32=variant)
                try:
                    cd = c.to(dev, bf)
                    with torch.no_grad():
                        if mode == "gated":
                            y, stored = ops.conv2d(ops.to_nhwc(r.to(dev, bf)), cd, gate=gate.to(dev), skip=ops.to_nhwc(x.to(dev, bf)), store_input=True)
                            ok &= report(f"v{variant} {shape} stored-input", ops.to_nchw(stored).float().cpu(), xin)
                        elif mode in ("res", "relu_post"):
                            y = ops.conv2d(ops.to_nhwc(x.to(dev, bf)), cd, residual=ops.to_nhwc(res.to(dev, bf)), **kw)
                        elif mode == "sums":
                            y, sums = ops.conv2d(ops.to_nhwc(x.to(dev, bf)), cd, want_sums=True)
                            tot = sums.float().reshape(2, -1, cout).sum(1).cpu()
                            ok &= report(f"v{variant} {shape} chan_sums", tot, ref.sum(dim=(2, 3)))
                        else:
                            y = ops.conv2d(ops.to_nhwc(x.to(dev, bf)), cd, **kw)
                        y = ops.to_nchw(y).float().cpu()
                    torch.cuda.synchronize()
                finally:
                    setk()
                if mode == "gelu":
                    refg = F.gelu(ref)
                    e = float((y - refg).abs().max() / refg.abs().max())
                    print(f"  {'ok  ' if e < 1e-2 else 'FAIL'}  v{variant} {shape} gelu rel err {e:.2e}"); ok &= e < 1e-2
                else:
                    ok &= report(f"v{variant} {shape} {mode}", y, ref)
    print("CHECK", "PASSED" if ok else "FAILED")
    return ok


def timeit(fn, iters=30, warm=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def timing():
    cases = [  # cin, cout, h, w, b, kwargs-name
        (192, 192, 544, 960, 8, "plain"), (192, 192, 544, 960, 8, "res"), (128, 128, 272, 480, 8, "relu"), (128, 128, 272, 480, 8, "gated"),
        (128, 128, 272, 480, 8, "sums"), (512, 512, 136, 240, 8, "plain"), (192, 128, 272, 480, 8, "plain"), (128, 192, 272, 480, 8, "res"),
        (48, 192, 1088, 1920, 8, "ps"), (48, 192, 544, 960, 8, "res"),
    ]
    for cin, cout, h, w, b, mode in cases:
        x = torch.rand(b, h, w, cin, device=dev).to(bf)
        kw = {}
        if mode == "res":
            kw = dict(residual=torch.rand(b, h, w, cout, device=dev).to(bf))
        if mode == "relu":
            kw = dict(act="relu")
        if mode == "gated":
            kw = dict(gate=torch.rand(b, cin, device=dev), skip=torch.rand_like(x), store_input=True)
        if mode == "sums":
            kw = dict(want_sums=True)
        if mode == "ps":
            kw = dict(out_mode=ops.RC_OUT_PIXEL_SHUFFLE2)
        fl = 2.0 * b * h * w * cin * cout * 9
        line = f"{cin:3d}->{cout:3d} {h}x{w}x{b} {mode:6s}:"
        variants = [("old", 0, 0), ("staged", 1, 0), ("ck32", 2, 0)]
        if cin == 48:
            variants = [("old", 0, 0), ("v1", 1, 0), ("v1-nodefer", 1, 8), ("v1-nostore", 1, 1), ("old-nostore", 0, 1)]
        else:
            variants += [("staged-nomfma", 1, 2), ("staged-nostore", 1, 1)]
        for name, v, fl_ in variants:
            setk(conv32=v, flags=fl_)
            c = N.Conv2d(cin, cout, 3, 1, 1).to(dev, bf)
            try:
                t = timeit(lambda: ops.conv2d(x, c, **kw))
                line += f"  {name} {t*1e3:.3f} ms ({fl/t/1e12:.0f} TF)"
            except Exception as e:
                line += f"  {name} ERR {str(e)[:60]}"
            setk()
        print(line, flush=True)
        del x, kw
        torch.cuda.empty_cache()


if __name__ == "__main__":
    torch.zeros(1, device=dev)
    ok = True
    if "--no-check" not in sys.argv:
        ok = check()
    if "--no-time" not in sys.argv:
        timing()
    sys.exit(0 if ok else 1)
