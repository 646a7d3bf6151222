#!/usr/bin/env python3
"""The folded tail (ops.tail_fold: one 5x5 conv 48 -> 12 + the exact border ring) against the two launches it replaces
(conv 48 -> 192 + PixelShuffle, conv 48 -> 3), at 8 x 1088 x 1920 -> 8 x 3 x 2160 x 3840: time per form of the 5x5 kernel, and agreement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from realcamnet_amd import networks as N, ops
from realcamnet_amd._lib import RC_OUT_NCHW, RC_OUT_PIXEL_SHUFFLE2, RC_OUT_PIXEL_SHUFFLE2_NCHW
L = ops.lib()
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)


def timed(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def psnr(a, b):
    return float(10 * torch.log10((b.max() - b.min()) ** 2 / ((a - b) ** 2).mean()))


tail = N.seq(N.conv(48, 192, mode="C"), torch.nn.PixelShuffle(2), N.conv(48, 3, mode="C")).to(dev, bf).eval()
c1, c2 = tail[0], tail[2]
with torch.no_grad():
    # ---- agreement on a small ragged case against an fp64 CPU reference, ring and interior separately
    for (b, H, W, crop) in ((2, 37, 70, None), (1, 16, 32, (30, 64)), (2, 24, 40, (47, 79))):
        x = torch.randn(b, H, W, 48, device=dev, dtype=bf)
        two = c2._nhwc(c1._nhwc(x, out_mode=RC_OUT_PIXEL_SHUFFLE2), out_mode=RC_OUT_NCHW, crop_hw=crop)
        one = ops.tail_fold(x, c1, c2, crop_hw=crop)
        xr = x.double().cpu().permute(0, 3, 1, 2)
        ref = F.conv2d(F.pixel_shuffle(F.conv2d(xr, c1.weight.double().cpu(), c1.bias.double().cpu(), padding=1), 2),
                       c2.weight.double().cpu(), c2.bias.double().cpu(), padding=1)
        if crop: ref = ref[:, :, :crop[0], :crop[1]]
        ring = torch.zeros_like(ref, dtype=torch.bool)
        ring[:, :, 0] = True; ring[:, :, :, 0] = True
        if not crop or crop[0] == 2 * H: ring[:, :, -1] = True
        if not crop or crop[1] == 2 * W: ring[:, :, :, -1] = True
        o, t = one.double().cpu(), two.double().cpu()
        print(f"{b}x{H}x{W} crop {crop}: two-step {psnr(t, ref):.1f} dB, folded {psnr(o, ref):.1f} dB; ring max|err| folded {(o - ref)[ring].abs().max():.4f} "
              f"two-step {(t - ref)[ring].abs().max():.4f}; interior max|err| folded {(o - ref)[~ring].abs().max():.4f} two-step {(t - ref)[~ring].abs().max():.4f}")
    # ---- time at the bench size
    B, H, W = 8, 1088, 1920
    x = torch.randn(B, H, W, 48, device=dev, dtype=bf)
    crop = (2160, 3840)
    for _ in range(10): c1._nhwc(x, out_mode=RC_OUT_PIXEL_SHUFFLE2)
    t_two = timed(lambda: c2._nhwc(c1._nhwc(x, out_mode=RC_OUT_PIXEL_SHUFFLE2), out_mode=RC_OUT_NCHW, crop_hw=crop))
    print(f"two launches (48->192 + PixelShuffle, 48->3): {t_two:.3f} ms")
    view = ops._folded_tail(c1, c2)[0]
    for persist, name in ((1, "persistent, 2 blocks/CU"), (2, "producer/consumer"), (0, "general")):
        L.rc_debug_set(b"persist", persist)
        t = timed(lambda: ops.conv2d(x, view, out_mode=RC_OUT_PIXEL_SHUFFLE2_NCHW, crop_hw=crop))
        print(f"folded 5x5 48->12, {name}: {t:.3f} ms")
    L.rc_debug_set(b"persist", 1)
    t_all = timed(lambda: ops.tail_fold(x, c1, c2, crop_hw=crop))
    print(f"ops.tail_fold (5x5 + ring: gather, 4 strip convs, scatter): {t_all:.3f} ms")
    one = ops.tail_fold(x, c1, c2, crop_hw=crop); two = c2._nhwc(c1._nhwc(x, out_mode=RC_OUT_PIXEL_SHUFFLE2), out_mode=RC_OUT_NCHW, crop_hw=crop)
    print(f"full size: folded vs two-step PSNR {psnr(one.float(), two.float()):.1f} dB, max|diff| {(one.float() - two.float()).abs().max():.4f}")
