#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
b, H, W, c = int(os.environ.get("B", 8)), 544, 960, 80
blk = M.GMA_Block(c, 8).to("cuda", torch.bfloat16).eval()
x = torch.randn(b, H * W, c, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    for _ in range(10): y = blk(x, (H, W))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): y = blk(x, (H, W))
    torch.cuda.synchronize()
print(f"GMA_Block(80,8) B={b} N={H*W}: {(time.perf_counter()-t0)/20*1e3:.2f} ms")
