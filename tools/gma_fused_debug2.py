import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops
torch.manual_seed(0)
dev = "cuda"
blk = M.GMA_Block(80, 8).to(dev, torch.bfloat16).eval()
with torch.no_grad():
    for p in blk.parameters():
        p.zero_()
a = torch.randn(2, 24, 40, 80, device=dev).to(torch.bfloat16)
R = torch.ops.realcam
f32 = ops.f32_param
with torch.no_grad():
    # pollute LDS / registers with a real layer-by-layer forward of another block first
    other = M.GMA_Block(80, 8).to(dev, torch.bfloat16).eval()
    ops.FUSE_GMA = False
    other._nhwc(torch.randn(2, 24, 40, 80, device=dev).to(torch.bfloat16))
    ops.FUSE_GMA = True
    x = a.clone()
    qkvp = torch.randn(2, 24, 40, 3, 64, device=dev).to(torch.bfloat16)
    convv = torch.randn(2, 24, 40, 64, device=dev).to(torch.bfloat16)
    loc = torch.randn(2, 24, 40, 16, device=dev).to(torch.bfloat16)
    ktv = torch.randn(2, 8, 8, 8, device=dev)
    wp, bp = ops.packed_chain(blk.att.proj); w1, b1 = ops.packed_chain(blk.mlp.fc1); w2, b2 = ops.packed_chain(blk.mlp.fc2)
    of = R.gma_tail(qkvp, convv, loc, x, ktv, wp, bp, f32(blk.norm2, "weight"), f32(blk.norm2, "bias"), 1e-5, w1, b1, w2, b2, None, None, None)
    torch.cuda.synchronize()
    d = (of.float() - x.float()).abs().reshape(-1, 80)
    bad = (d > 0)
    print("RC_TAIL_DBG", os.environ.get("RC_TAIL_DBG"), "mismatches", int(bad.sum()), "channels hit", bad.any(dim=0).nonzero().flatten().tolist()[:40])
