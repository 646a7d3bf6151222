#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache (MALL) pay for a producer -> consumer hop?  (probe before any kernel work.)

 A. streaming rate by working-set size: copy / read / write of 16 MB .. 1 GB looped in place (rc_debug_hbm_probe).
 B. one RCAB body (conv 48->48 + ReLU -> conv 48->48 + CALayer sums) at level 0 (8 x 1088 x 1920) and level 1 (8 x 544 x 960),
    and a 192 -> 192 pair at level 1: whole batch per launch (as shipped) against frame-major / band-major launch order
    (bands faked with shorter images: the timing question is the reuse distance, not the halo).
 C. the whole cfg3 forward: 1 x B=8 against 2 x B=4, 4 x B=2, 8 x B=1.
"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import networks as N, ops
L = ops.lib()
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
MB = 1 << 20


def timed(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def part_a():
    print("== A. streaming rate by working-set size (looped in place, 30 iterations after 20 warm-ups)")
    big = 1024 * MB
    a = torch.empty(big, dtype=torch.uint8, device=dev).random_(0, 255); b = torch.empty_like(a)
    for mb in (16, 32, 64, 96, 128, 192, 256, 384, 512, 1024):
        row = []
        for mode, name in ((0, "copy"), (1, "read"), (2, "write")):
            ms = C.c_double()
            nbytes = mb * MB // (2 if mode == 0 else 1)          # copy: src + dst together are the working set
            ops.check(L.rc_debug_hbm_probe(a.data_ptr(), b.data_ptr(), nbytes, mode, 0, 0, 0, 30, C.byref(ms)), "probe")
            moved = nbytes * (2 if mode == 0 else 1)
            row.append(f"{name} {moved / ms.value / 1e9:6.2f} TB/s ({ms.value * 1e3:7.1f} us)")
        print(f"  working set {mb:5d} MB: " + "   ".join(row))
    del a, b


def chain(convs, x_all, group):
    """Run the layer chain group by group (group = images per launch)."""
    outs = None
    for g0 in range(0, x_all.shape[0], group):
        t = x_all[g0:g0 + group]
        for c, kw in convs:
            t = c._nhwc(t, **kw)
            if isinstance(t, tuple):
                t = t[0]
        outs = t
    return outs


def part_b():
    print("== B. producer -> consumer pairs: launch order")
    with torch.no_grad():
        for label, shape, cin, kws in (
                ("level 0 RCAB body 48->48 relu, 48->48 +sums", (8, 1088, 1920), 48, (dict(act="relu"), dict(want_sums=True))),
                ("level 0, 4-conv chain 48->48", (8, 1088, 1920), 48, (dict(act="relu"), dict(), dict(act="relu"), dict())),
                ("level 1 RCAB body 48->48", (8, 544, 960), 48, (dict(act="relu"), dict(want_sums=True))),
                ("level 1 pair 192->192 (Res_GFM)", (8, 544, 960), 192, (dict(act="relu"), dict())),
                ("level 2 pair 128->128", (8, 272, 480), 128, (dict(act="relu"), dict()))):
            B, H, W = shape
            convs = [(N.Conv2d(cin, cin, 3, 1, 1).to(dev, bf).eval(), kw) for kw in kws]
            x = torch.randn(B, H, W, cin, device=dev, dtype=bf)
            for _ in range(20): chain(convs, x, B)                                  # clocks up
            base = timed(lambda: chain(convs, x, B))
            print(f"  {label}: batch of {B} per launch {base:7.3f} ms  ({B * H * W * cin * 2 / MB:.0f} MB per map)")
            for group in (4, 2, 1):
                t = timed(lambda: chain(convs, x, group))
                print(f"      {group} frame(s) per launch ({group * H * W * cin * 2 / MB:5.0f} MB per map): {t:7.3f} ms  x{base / t:.2f}")
            for bands in (2, 4, 8):                                                 # fake bands: B*bands images of H/bands rows
                if H % (bands * 8):
                    continue
                xb = x.view(B * bands, H // bands, W, cin)
                t = timed(lambda: chain(convs, xb, 1))
                print(f"      1/{bands} frame per launch ({H // bands * W * cin * 2 / MB:5.0f} MB per map): {t:7.3f} ms  x{base / t:.2f}")
            del x, convs


def part_c():
    print("== C. whole cfg3 forward (LiteISPNet_GFM_LSC_GMA, 8 frames of 4K, bf16): frames per launch")
    net = M.LiteISPNet_GFM_LSC_GMA().eval().to(dev, bf)
    g = torch.Generator().manual_seed(1234)
    mosaic = torch.rand(8, 1, 2160, 3840, generator=g).to(dev, bf)
    coord = ops.make_coord(8, 1080, 1920, device=dev, dtype=bf)
    with torch.no_grad():
        def run(group):
            for g0 in range(0, 8, group):
                y = net.forward_mosaic(mosaic[g0:g0 + group], None, coord[g0:g0 + group])
            return y
        for _ in range(3): run(8)
        base = timed(lambda: run(8), n=4, warm=1)
        print(f"  8 frames per forward: {base:7.2f} ms")
        for group in (4, 2, 1):
            t = timed(lambda: run(group), n=4, warm=1)
            print(f"  {group} frame(s) per forward: {t:7.2f} ms  x{base / t:.2f}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["a", "b", "c"]
    if "a" in which: part_a()
    if "b" in which: part_b()
    if "c" in which: part_c()
