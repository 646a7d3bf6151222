import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import realcamnet_amd.raw2bit as RB
from realcamnet_amd import ops
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
H, W = 1152, 1920
m = RB.raw_compression_tcm_final().eval(); m.update(); m = m.to("cuda", torch.bfloat16)
x = [torch.rand(1, 4, H, W, generator=g).to("cuda", torch.bfloat16), torch.rand(1, 4, 256, 256, generator=g).to("cuda", torch.bfloat16), ops.make_coord(1, H, W, "cuda", torch.bfloat16)]
with torch.no_grad():
    for _ in range(2): m(x)
    for _ in range(2): enc = m.compress(x)
    torch.cuda.synchronize()
    t0=time.perf_counter(); enc = m.compress(x); torch.cuda.synchronize(); print("compress ms", (time.perf_counter()-t0)*1e3)
    t0=time.perf_counter(); m(x); torch.cuda.synchronize(); print("forward ms", (time.perf_counter()-t0)*1e3)
    t0=time.perf_counter(); m.decompress(enc["strings"], enc["shape"]); torch.cuda.synchronize(); print("decompress ms", (time.perf_counter()-t0)*1e3)
