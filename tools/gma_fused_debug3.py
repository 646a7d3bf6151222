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
mode = os.environ.get("MODE", "real")
with torch.no_grad():
    x = blk.cpe._nhwc(a)
    qkv_l = ops.conv2d(ops.layernorm(x, blk.norm1), blk.att.qkv)
    qkvp, loc, convv, ktv = blk.att._context(qkv_l)
    print("ranges: qkvp", qkvp.float().abs().max().item(), "convv", convv.float().abs().max().item(), "loc", loc.float().abs().max().item(),
          "ktv", ktv.abs().max().item(), "finite", bool(torch.isfinite(qkvp.float()).all() and torch.isfinite(ktv).all()))
    if mode.startswith("lnqkv_first"):
        wq, bq = ops.packed_chain(blk.att.qkv)
        junk = R.gma_ln_qkv(x, wq, bq, f32(blk.norm1, "weight"), f32(blk.norm1, "bias"), 1e-5)
        if mode.endswith("sync"):
            torch.cuda.synchronize()
    if mode == "rand_q":
        qkvp = torch.randn_like(qkvp.float()).to(torch.bfloat16)
    if mode == "rand_cv":
        convv = torch.randn_like(convv.float()).to(torch.bfloat16)
    if mode == "rand_loc":
        loc = torch.randn_like(loc.float()).to(torch.bfloat16)
    if mode == "rand_ktv":
        ktv = torch.randn_like(ktv)
    if mode == "clone_all":
        qkvp, convv, loc, ktv, x = qkvp.clone(), convv.clone(), loc.clone(), ktv.clone(), x.clone()
    b2k = copy.deepcopy(blk)
    for p in b2k.parameters():
        p.zero_()
    ops.invalidate_caches(b2k)
    wp, bp = ops.packed_chain(b2k.att.proj); w1, b1 = ops.packed_chain(b2k.mlp.fc1); w2, b2 = ops.packed_chain(b2k.mlp.fc2)
    of = R.gma_tail(qkvp, convv, loc, x, ktv, wp, bp, f32(b2k.norm2, "weight"), f32(b2k.norm2, "bias"), 1e-5, w1, b1, w2, b2, None, None, None)
    torch.cuda.synchronize()
    d = (of.float() - x.float()).abs().reshape(-1, 80)
    bad = (d > 0)
    print("MODE", mode, "RC_TAIL_DBG", os.environ.get("RC_TAIL_DBG"), "mismatches", int(bad.sum()), "channels hit", bad.any(dim=0).nonzero().flatten().tolist()[:24],
          "wp sum", wp.float().sum().item(), "w1 sum", w1.float().sum().item(), "w2 sum", w2.float().sum().item())
