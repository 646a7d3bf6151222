import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops
R = torch.ops.realcam
torch.manual_seed(0)
blk = M.GMA_Block(80, 8).to("cuda", torch.bfloat16).eval()
for (b, H, W) in [(1, 16, 32), (2, 37, 53), (1, 130, 201), (8, 544, 960)]:
    x = torch.randn(b, H, W, 80, device="cuda").to(torch.bfloat16)
    with torch.no_grad():
        wq, bq = ops.packed_chain(blk.att.qkv)
        qkv = R.gma_ln_qkv(x, wq, bq, ops.f32_param(blk.norm1, "weight"), ops.f32_param(blk.norm1, "bias"), 1e-5)
        want = blk.att.aggregator._run_fused(qkv)
        got = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
        again = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
    torch.cuda.synchronize()
    print(f"--- {b} x {H} x {W}")
    for name, w_, g_, a_ in zip(("qkvp", "loc", "kmax"), want, got, again):
        d1 = (w_.float() - g_.float()); d2 = (g_.float() - a_.float())
        print(f"  {name}: vs two-launch: {int((d1 != 0).sum())} of {d1.numel()} differ (max {d1.abs().max().item():.4g}); run-to-run: {int((d2 != 0).sum())} differ")
        if name == "qkvp" and (d1 != 0).any():
            per_seg = (d1 != 0).reshape(12, -1).sum(1).tolist()
            print("    per segment:", per_seg)
            idx = (d1 != 0).nonzero()[:6].tolist()
            print("    first:", idx)
        if name == "loc" and (d1 != 0).any():
            print("    first:", (d1 != 0).nonzero()[:6].tolist())


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(8, 544, 960, 80, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    print(f"front: {timeit(lambda: blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)):8.1f} us")

