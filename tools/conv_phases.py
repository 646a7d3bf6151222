#!/usr/bin/env python3
"""Per-phase s_memtime cycle counts of the producer/consumer conv kernel (one compute wave, one loader wave)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
gated = "--gated" in sys.argv
flags = int(os.environ.get("FLAGS", "0"))
c = N.Conv2d(48, 48, 3, 1, 1).to("cuda", torch.bfloat16)
x = torch.rand(8, 1088, 1920, 48, device="cuda").to(torch.bfloat16)
kw = dict(gate=torch.rand(8, 48, device="cuda"), skip=torch.rand_like(x), store_input=True) if gated else {}
dbg = torch.zeros(1024, dtype=torch.int64, device="cuda")
ops.lib().rc_debug_set(b"persist", 2)
for _ in range(60): ops.conv2d(x, c, act="relu", **kw)
torch.cuda.synchronize()
ops.lib().rc_debug_set(b"conv_flags", flags)
ops.lib().rc_debug_set(b"persist", 2)
ops.lib().rc_debug_set_ptr(b"conv_phase_timing", dbg.data_ptr())
import time; t0 = time.perf_counter()
ops.conv2d(x, c, act="relu", **kw); torch.cuda.synchronize()
dt = time.perf_counter() - t0
ops.lib().rc_debug_set_ptr(b"conv_phase_timing", None)
ops.lib().rc_debug_set(b"conv_flags", 0)
ops.lib().rc_debug_set(b"persist", 1)
d = dbg.cpu()
cw = d[:256].view(64, 4)[4:60].float(); lw = d[256:384].view(64, 2)[4:60].float()
print(f"flags={flags} wall {dt*1e3:.2f} ms gated={gated}  compute wave: mma {cw[:,0].mean():.0f}  epilogue {cw[:,1].mean():.0f}  barrier-wait {cw[:,2].mean():.0f} cycles/tile")
print(f"             loader wave:  commit+issue {lw[:,0].mean():.0f}  barrier-wait {lw[:,1].mean():.0f} cycles/tile")

