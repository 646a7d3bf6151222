"""Is the codec forward bound by the host's launch rate?  Times raw_compression_tcm_final.forward_mosaic (4 frames, 4K) three ways:
eager (two-stream branches), eager on one stream, and as a replayed HIP graph (one stream, no host work at all)."""
import sys, time
sys.path.insert(0, ".")
import torch
import realcamnet_amd as M
from realcamnet_amd import ops

dev, dt = torch.device("cuda:0"), torch.bfloat16
frames, H2, W2 = 4, 2160, 3840
torch.manual_seed(0)
net = M.raw2bit.raw_compression_tcm_final().eval().to(device=dev, dtype=dt)
g = torch.Generator(device=dev).manual_seed(4321)
mosaic = torch.rand(frames, 1, H2, W2, generator=g, device=dev).to(dt)
coord = ops.make_coord(frames, H2 // 2, W2 // 2, device=dev, dtype=dt)


def step():
    with torch.no_grad():
        return net.forward_mosaic(mosaic, None, coord)


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def host_only(n=3):       # host time to ENQUEUE one forward (no sync inside): the launch rate the GPU has to be fed at
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return 1e3 * (t1 - t0) / n


print("eager, two streams  ms", round(timed(step), 3))
print("host enqueue time   ms", round(host_only(), 3))
ops.BRANCH_STREAMS = False
print("eager, one stream   ms", round(timed(step), 3))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = step()
print("graph replay        ms", round(timed(graph.replay), 3))
