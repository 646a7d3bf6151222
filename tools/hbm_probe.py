#!/usr/bin/env python3
"""What streaming rate does this MI355X sustain?  Copy / read / write of a 1.6 GB buffer by grid shape and cache policy."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import ops
GB = 1088 * 1920 * 8 * 48 * 2
a = torch.empty(GB, dtype=torch.uint8, device="cuda").random_(0, 255); b = torch.empty_like(a)
L = ops.lib()
def run(mode, nt, contiguous, blocks, nbytes=GB):
    ms = C.c_double()
    ops.check(L.rc_debug_hbm_probe(a.data_ptr(), b.data_ptr(), nbytes, mode, nt, contiguous, blocks, 30, C.byref(ms)), "probe")
    moved = nbytes * (2 if mode == 0 else 1)
    return ms.value, moved / ms.value / 1e9
for mode, name in ((0, "copy"), (1, "read"), (2, "write")):
    for nt in (0, 1):
        for contiguous, blocks in ((0, 0), (0, 256 * 8), (0, 256 * 16), (1, 256 * 8), (1, 256 * 32)):
            ms, tb = run(mode, nt, contiguous, blocks)
            print(f"{name:5s} nt={nt} {'contig' if contiguous else 'stride'} blocks={blocks or 'one-shot':>8}: {ms:.3f} ms  {tb:.2f} TB/s")
for frac in (8, 64):
    ms, tb = run(0, 0, 0, 0, GB // frac // 16 * 16)
    print(f"copy of {GB // frac / 1e6:.0f} MB: {ms:.4f} ms {tb:.2f} TB/s")
t = torch.empty(GB // 2, dtype=torch.bfloat16, device="cuda"); u = torch.empty_like(t)
for _ in range(20): u.copy_(t)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): u.copy_(t)
e1.record(); torch.cuda.synchronize()
print(f"torch copy_: {e0.elapsed_time(e1) / 30:.3f} ms  {2 * GB / (e0.elapsed_time(e1) / 30) / 1e9:.2f} TB/s")
