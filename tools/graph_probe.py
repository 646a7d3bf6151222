#!/usr/bin/env python3
"""Capture one forward in a HIP graph (torch.cuda.CUDAGraph) and replay it: checks the path is capturable (no host sync, no
allocation outside the caching allocator, everything on the current stream) and times eager vs replay for small batches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import realcamnet_amd as M
import liteisp_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev, dt = "cuda", torch.bfloat16
torch.manual_seed(0)
net = M.LiteISPNet_GFM_LSC_GMA().eval().to(dev, dt)
g = torch.Generator(device=dev).manual_seed(1)
mosaic = torch.rand(B, 1, 2160, 3840, generator=g, device=dev).to(dt)
cond = torch.rand(B, 4, 256, 256, generator=g, device=dev).to(dt)
coord = O.make_coord(B, 1080, 1920).to(dev, dt)
def fwd():
    with torch.no_grad():
        return net.forward_mosaic(mosaic, cond, coord)
for _ in range(3): y_eager = fwd()
torch.cuda.synchronize()
def timeit(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t_eager = timeit(fwd)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): fwd()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    y_graph = fwd()
graph.replay(); torch.cuda.synchronize()
same = torch.equal(y_graph, y_eager)
t_graph = timeit(graph.replay)
print(f"B={B}: eager {t_eager:.2f} ms/forward, graph replay {t_graph:.2f} ms/forward, outputs bitwise equal: {same}")
