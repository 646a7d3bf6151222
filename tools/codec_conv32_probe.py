"""Codec forward (4 frames, 4K) under the conv32 knob: 4 = auto (48 -> 192 only), 1 = staged CK16 where eligible, 2 / 3 = CK32 two-barrier forms."""
import sys, time
sys.path.insert(0, ".")
import torch
import realcamnet_amd as M
from realcamnet_amd import ops, _lib

L = _lib.load()
dev, dt = torch.device("cuda:0"), torch.bfloat16
frames, H2, W2 = 4, 2160, 3840
g = torch.Generator(device=dev).manual_seed(4321)
mosaic = torch.rand(frames, 1, H2, W2, generator=g, device=dev).to(dt)
coord = ops.make_coord(frames, H2 // 2, W2 // 2, device=dev, dtype=dt)
ref = None
for knob in (4, 1, 2, 3, 0):
    L.rc_debug_set(b"conv32", knob)
    torch.manual_seed(0)
    net = M.raw2bit.raw_compression_tcm_final().eval().to(device=dev, dtype=dt)
    with torch.no_grad():
        for _ in range(2):
            out = net.forward_mosaic(mosaic, None, coord)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = net.forward_mosaic(mosaic, None, coord)
        torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 5
    x = out["x_hat"].float()
    if ref is None:
        ref = x
    print(f"conv32={knob}  {ms:7.3f} ms   max|x_hat - auto| {float((x - ref).abs().max()):.4g}")
    del net
L.rc_debug_set(b"conv32", 4)
