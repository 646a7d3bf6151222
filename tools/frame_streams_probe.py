#!/usr/bin/env python3
"""cfg3 step (8 frames of 4K, bf16, LiteISPNet_GFM_LSC_GMA) as ONE batch on one stream vs k groups of 8 / k frames on k streams: does a second stream's queue
fill the launch tails / hide the tiny dependent kernels?  Frames are independent (frame i of a batch == frame i alone, bitwise: tests/test_full_size.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops
dev, dt = torch.device("cuda:0"), torch.bfloat16
torch.manual_seed(0)
net = M.LiteISPNet_GFM_LSC_GMA().eval().to(dev, dt)
g = torch.Generator(device=dev).manual_seed(1234)
B, H2, W2 = 8, 2160, 3840
mosaic = torch.rand(B, 1, H2, W2, generator=g, device=dev).to(dt)
coord = ops.make_coord(B, H2 // 2, W2 // 2, device=dev, dtype=dt)
streams = [torch.cuda.Stream() for _ in range(4)]


def step(k):
    with torch.no_grad():
        if k == 1:
            return [net.forward_mosaic(mosaic, None, coord)]
        outs, main = [], torch.cuda.current_stream()
        n = B // k
        for i in range(k):
            s = streams[i]
            s.wait_stream(main)
            with torch.cuda.stream(s):
                outs.append(net.forward_mosaic(mosaic[i * n:(i + 1) * n], None, coord[i * n:(i + 1) * n]))
        for i in range(k):
            main.wait_stream(streams[i])
        return outs


def timed(k, n=6, warm=3):
    for _ in range(warm): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): step(k)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


ref = torch.cat(step(1)).clone()
for k in (1, 2, 4, 1, 2, 4):
    t = timed(k)
    same = torch.equal(torch.cat(step(k)), ref)
    print(f"{k} stream(s) x {B // k} frames: {t:7.2f} ms per 8 frames   (output bit-identical to one batch: {same})", flush=True)
