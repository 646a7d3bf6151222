"""Time the GroupMix block's launches one by one at the cfg3 size (8 x 544 x 960 tokens, dim 80, bf16) with HIP events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops, networks as N

dev = "cuda"
torch.manual_seed(0)
B, H, W = 8, 544, 960
blk = M.GMA_Block(80, 8).to(dev, torch.bfloat16).eval()
gin = N.Conv2d(192, 80, 1, 1, 0).to(dev, torch.bfloat16)
gout = N.Conv2d(80, 192, 1, 1, 0).to(dev, torch.bfloat16)
d1 = torch.randn(B, H, W, 192, device=dev).to(torch.bfloat16)
R = torch.ops.realcam
f32 = ops.f32_param


def timeit(name, fn, n=10):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) / n * 1e3:9.1f} us")
    return out


with torch.no_grad():
    t = timeit("gma_in 1x1 192->80", lambda: gin._nhwc(d1))
    x = timeit("cpe dw3x3 + identity", lambda: blk.cpe._nhwc(t))
    timeit("gma_in + cpe in ONE launch", lambda: blk._entry(d1, gin))
    wq, bq = ops.packed_chain(blk.att.qkv)
    qkv = timeit("ln_qkv", lambda: R.gma_ln_qkv(x, wq, bq, f32(blk.norm1, "weight"), f32(blk.norm1, "bias"), 1e-5))
    qkvp, loc, kmax = timeit("aggregate (+ per-channel max of k)", lambda: blk.att.aggregator._run_fused(qkv))
    timeit("ln_qkv + aggregate in ONE launch", lambda: blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv))
    convv = timeit("crpe", lambda: blk.att.crpe._conv_v(qkvp))
    timeit("kv, two-pass VALU form (max, sums, merge)", lambda: R.gma_kv(qkvp, 8, 8, float(blk.att.scale)))
    ktv = timeit("kv on the matrix cores (sums, merge)", lambda: R.gma_kv_mfma(qkvp, kmax, float(blk.att.scale)))
    wp, bp = ops.packed_chain(blk.att.proj); w1, b1 = ops.packed_chain(blk.mlp.fc1); w2, b2 = ops.packed_chain(blk.mlp.fc2)
    wo, bo = ops.packed_chain(gout)
    timeit("tail (+ out conv)", lambda: R.gma_tail(qkvp, convv, loc, x, ktv, wp, bp, f32(blk.norm2, "weight"), f32(blk.norm2, "bias"), 1e-5,
                                                   w1, b1, w2, b2, d1, wo, bo))
    timeit("whole block (+in/out conv), entry as two launches", lambda: blk._nhwc(gin._nhwc(d1), post=(gout, d1)))
    timeit("whole block (+in/out conv)", lambda: blk._nhwc(d1, pre=gin, post=(gout, d1)))
