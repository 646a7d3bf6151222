#!/usr/bin/env python3
"""One 4K frame through raw_compression_tcm_final (bf16): forward eager vs HIP-graph replay (realcamnet_amd.GraphedCall), compress, decompress -- wall ms per call,
host-side (perf_counter around call + synchronize), median of n.  VERDICT r5 item 5."""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H2, W2 = 2160, 3840
torch.manual_seed(0)
net = M.raw2bit.raw_compression_tcm_final().eval().to(device=dev, dtype=dt)
net.update()
g = torch.Generator(device=dev).manual_seed(4321)
mosaic = torch.rand(frames, 1, H2, W2, generator=g, device=dev).to(dt)
coord = ops.make_coord(frames, H2 // 2, W2 // 2, device=dev, dtype=dt)


def wall(fn, n=7, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    return statistics.median(ts)


with torch.no_grad():
    eager = lambda: net.forward_mosaic(mosaic, None, coord)
    print(f"forward eager           {wall(eager):7.2f} ms")
    gf = M.GraphedCall(lambda m, c: net.forward_mosaic(m, None, c))
    ref = eager()["x_hat"].clone()
    out = gf(mosaic, coord)
    torch.cuda.synchronize()
    print(f"forward graph replay    {wall(lambda: gf(mosaic, coord)):7.2f} ms   (x_hat bit-identical to eager: {torch.equal(out['x_hat'], ref)})")
    ops.GRAPH_FORK = False
    gf1 = M.GraphedCall(lambda m, c: net.forward_mosaic(m, None, c))
    print(f"forward graph, 1 stream {wall(lambda: gf1(mosaic, coord)):7.2f} ms")
    ops.GRAPH_FORK = True
    if frames == 1 and "--codec" in sys.argv:
        a, cond = M.LiteISP._ingest(net, mosaic, None, dt, 128, 0.0, 1.0, (256, 256))
        x = [ops.to_nchw(a), cond, ops.to_nchw(ops.to_nhwc(coord, dtype=dt, pad_hw=(a.shape[1], a.shape[2])))]
        enc = net.compress(x)
        print(f"compress                {wall(lambda: net.compress(x)):7.2f} ms   ({sum(len(s) for s in enc['strings'][0]) + sum(len(s) for s in enc['strings'][1])} bytes)")
        encg = net.compress(x, graph=True)
        print(f"compress, graph replay  {wall(lambda: net.compress(x, graph=True)):7.2f} ms   (strings identical: {encg['strings'] == enc['strings']})")
        print(f"decompress              {wall(lambda: net.decompress(enc['strings'], enc['shape'])):7.2f} ms")
