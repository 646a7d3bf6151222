#!/usr/bin/env python3
"""Does an HBM-bound conv on one half of the CUs overlap with a matrix-bound conv on the other half?  """
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
dt = torch.bfloat16
c48 = N.Conv2d(48, 48, 3, 1, 1).to("cuda", dt); x48 = torch.rand(4, 1088, 1920, 48, device="cuda").to(dt)
c192 = N.Conv2d(192, 192, 3, 1, 1).to("cuda", dt); x192 = torch.rand(4, 544, 960, 192, device="cuda").to(dt)
def hbm(n):
    for _ in range(n): ops.conv2d(x48, c48, act="relu")
def mfma(n):
    for _ in range(n): ops.conv2d(x192, c192)
for _ in range(30): hbm(1); mfma(1)
torch.cuda.synchronize()
def timed(f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
N_IT = 60
print(f"full chip, back to back: hbm {timed(lambda: hbm(N_IT)) / N_IT:.3f} ms  mfma {timed(lambda: mfma(N_IT)) / N_IT:.3f} ms  (half batches of 4 frames)")
for kinds in ((0, 1), (2, 3)):
    ptrs = []
    for k in kinds:
        p = C.c_void_p()
        ops.check(ops.lib().rc_debug_stream_create_masked(k, C.byref(p)), "masked stream")
        ptrs.append(torch.cuda.ExternalStream(p.value))
    sa, sb = ptrs
    def on(s, f, n):
        with torch.cuda.stream(s): f(n)
    on(sa, hbm, 5); on(sb, mfma, 5); torch.cuda.synchronize()
    ta = timed(lambda: on(sa, hbm, N_IT)) / N_IT
    tb = timed(lambda: on(sb, mfma, N_IT)) / N_IT
    def both():
        for _ in range(N_IT):
            on(sa, hbm, 1); on(sb, mfma, 1)
    tt = timed(both) / N_IT
    print(f"mask kinds {kinds}: alone hbm {ta:.3f} ms, alone mfma {tb:.3f} ms, concurrent pair {tt:.3f} ms (sum of full-chip times above is the number to beat)")
