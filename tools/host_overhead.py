#!/usr/bin/env python3
"""Host-side cost of one forward: tiny frames, so GPU time is negligible and wall time ~= Python/ctypes launch overhead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, liteisp_oracle as O, realcamnet_amd as M
name = sys.argv[1] if len(sys.argv) > 1 else "LiteISPNet_GFM_LSC_GMA"
net = getattr(M, name)().eval().to("cuda", torch.bfloat16)
mosaic = torch.rand(1, 1, 128, 128, device="cuda").bfloat16(); cond = torch.rand(1, 4, 64, 64, device="cuda").bfloat16()
coord = O.make_coord(1, 64, 64).to("cuda", torch.bfloat16)
with torch.no_grad():
    for _ in range(3): net.forward_mosaic(mosaic, cond, coord)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): net.forward_mosaic(mosaic, cond, coord)
    t_issue = (time.perf_counter() - t0) / 20
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 20
print(f"{name}: host issue {t_issue*1e3:.2f} ms/forward, wall {t_all*1e3:.2f} ms/forward (OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')})")
