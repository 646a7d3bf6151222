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
        if p.dim() == 1:
            p.add_(torch.randn_like(p) * 0.1)
a = torch.randn(2, 24, 40, 80, device=dev).to(torch.bfloat16)
R = torch.ops.realcam
f32 = ops.f32_param
dbg = int(os.environ.get("RC_TAIL_DBG", "0"))
with torch.no_grad():
    x = blk.cpe._nhwc(a)
    qkv_l = ops.conv2d(ops.layernorm(x, blk.norm1), blk.att.qkv)
    qkvp, loc, convv, ktv = blk.att._context(qkv_l)
    wp, bp = ops.packed_chain(blk.att.proj); w1, b1 = ops.packed_chain(blk.mlp.fc1); w2, b2 = ops.packed_chain(blk.mlp.fc2)
    of = R.gma_tail(qkvp, convv, loc, x, ktv, wp, bp, f32(blk.norm2, "weight"), f32(blk.norm2, "bias"), 1e-5, w1, b1, w2, b2, None, None, None)
    y = R.gma_apply(qkvp, convv, loc, ktv, 8, 8, 16)
    x2 = ops.conv2d(y, blk.att.proj, residual=x)
    full = blk.mlp._nhwc(ops.layernorm(x2, blk.norm2), residual=x2)
    b2f = blk.mlp.fc2.bias.float()
    if dbg & 1:
        ref = (x2.float() + b2f).to(torch.bfloat16) if not (dbg & 2) else None
        if dbg & 2:   # no proj: x + b_proj, then + b_fc2
            t = (x.float() + blk.att.proj.bias.float()).to(torch.bfloat16)
            ref = (t.float() + b2f).to(torch.bfloat16)
    else:
        ref = full
    torch.cuda.synchronize()
    g, w = of.float(), ref.float()
    bad = ~torch.isfinite(g)
    d = (g - w).abs(); d[bad] = 0
    tokbad = ((d > 0.05 * (1 + w.abs())) | bad).reshape(-1, 80).any(dim=1).nonzero().flatten()
    print("RC_TAIL_DBG", dbg, "nonfinite", int(bad.sum()), "max|diff|", d.max().item(), "bad tokens", tokbad[:20].tolist(), "total", tokbad.numel())
