"""Debug aid: fused GroupMix stages (csrc/gma_fused.hip) against the layer-by-layer ops, stage by stage."""
import os, sys
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
H, W = 24, 40
a = torch.randn(2, H, W, 80, device=dev).to(torch.bfloat16)
R = torch.ops.realcam
f32 = ops.f32_param


def to_planar(t):      # (B,H,W,C) -> (C/16,B,H,W,16)
    b, h, w, c = t.shape
    return t.reshape(b, h, w, c // 16, 16).permute(3, 0, 1, 2, 4).contiguous()


def from_planar(t):    # (S,B,H,W,16) -> (B,H,W,16 S)
    s_, b, h, w, _ = t.shape
    return t.permute(1, 2, 3, 0, 4).reshape(b, h, w, 16 * s_)


def cmp(name, got, want):
    g, w = got.float(), want.float()
    bad = ~torch.isfinite(g)
    d = (g - w).abs()
    d[bad] = 0
    print(f"{name}: shape {tuple(g.shape)} nonfinite {int(bad.sum())} max|diff| {d.max().item():.4g} (max|ref| {w.abs().max().item():.4g})")
    if bad.any():
        idx = bad.reshape(-1, g.shape[-1]).any(dim=1).nonzero().flatten()
        print("   tokens with non-finite values:", idx[:40].tolist(), "... total", idx.numel())
        ch = bad.reshape(-1, g.shape[-1]).any(dim=0).nonzero().flatten()
        print("   channels:", ch[:40].tolist(), "total", ch.numel())
    else:
        worst = d.reshape(-1, g.shape[-1]).max(dim=1).values.argmax().item()
        print("   worst token", worst, "channel", d.reshape(-1, g.shape[-1])[worst].argmax().item())


with torch.no_grad():
    x = blk.cpe._nhwc(a)
    # stage 1: LN1 + qkv
    wq, bq = ops.packed_chain(blk.att.qkv)
    qkv_f = R.gma_ln_qkv(x, wq, bq, f32(blk.norm1, "weight"), f32(blk.norm1, "bias"), float(blk.norm1.eps))
    qkv_l = ops.conv2d(ops.layernorm(x, blk.norm1), blk.att.qkv)
    qkv_planar = qkv_f
    qkv_f = qkv_f.permute(1, 2, 3, 0, 4).reshape(*qkv_l.shape)
    cmp("ln_qkv", qkv_f, qkv_l)
    qkvp, loc, convv, ktv = blk.att._context(qkv_l)
    # stage 2: tail
    wp, bp = ops.packed_chain(blk.att.proj)
    w1, b1 = ops.packed_chain(blk.mlp.fc1)
    w2, b2 = ops.packed_chain(blk.mlp.fc2)
    qkvp_p, convv_p = to_planar(qkvp.reshape(*qkvp.shape[:3], 192)), to_planar(convv)
    cmp("kv planar vs token-major", R.gma_kv(qkvp_p, 8, 8, float(blk.att.scale)), ktv)
    out_f = R.gma_tail(qkvp_p, convv_p, loc, x, ktv, wp, bp, f32(blk.norm2, "weight"), f32(blk.norm2, "bias"), float(blk.norm2.eps), w1, b1, w2, b2,
                       None, None, None)
    y = R.gma_apply(qkvp, convv, loc, ktv, 8, 8, 16)
    x2 = ops.conv2d(y, blk.att.proj, residual=x)
    out_l = blk.mlp._nhwc(ops.layernorm(x2, blk.norm2), residual=x2)
    cmp("tail", out_f, out_l)
    torch.cuda.synchronize()

# ---- isolate stages by zeroing layers -------------------------------------------------------------------------------------------
import copy
def tail_both(b2k):
    with torch.no_grad():
        wp, bp = ops.packed_chain(b2k.att.proj); w1, b1 = ops.packed_chain(b2k.mlp.fc1); w2, b2 = ops.packed_chain(b2k.mlp.fc2)
        of = R.gma_tail(qkvp_p, convv_p, loc, x, ktv, wp, bp, f32(b2k.norm2, "weight"), f32(b2k.norm2, "bias"), float(b2k.norm2.eps), w1, b1, w2, b2,
                        None, None, None)
        y = R.gma_apply(qkvp, convv, loc, ktv, 8, 8, 16)
        x2 = ops.conv2d(y, b2k.att.proj, residual=x)
        ol = b2k.mlp._nhwc(ops.layernorm(x2, b2k.norm2), residual=x2)
    return of, ol
for name, edit in (("fc2=0 (out = x2)", lambda m: (m.mlp.fc2.weight.zero_(), m.mlp.fc2.bias.zero_())),
                   ("fc1=0", lambda m: (m.mlp.fc1.weight.zero_(),)),
                   ("proj=0,fc2=0 (out = x + b)", lambda m: (m.att.proj.weight.zero_(), m.mlp.fc2.weight.zero_(), m.mlp.fc2.bias.zero_()))):
    b2k = copy.deepcopy(blk)
    with torch.no_grad():
        edit(b2k)
    ops.invalidate_caches(b2k)
    of, ol = tail_both(b2k)
    cmp(name, of, ol)
# hidden pre-activation range
with torch.no_grad():
    y = R.gma_apply(qkvp, convv, loc, ktv, 8, 8, 16)
    x2 = ops.conv2d(y, blk.att.proj, residual=x)
    hpre = ops.conv2d(ops.layernorm(x2, blk.norm2), blk.mlp.fc1)
    print("fc1 pre-activation range", hpre.float().min().item(), hpre.float().max().item())

b2k = copy.deepcopy(blk)
with torch.no_grad():
    for p in b2k.parameters():
        p.zero_()
ops.invalidate_caches(b2k)
of, ol = tail_both(b2k)
cmp("all parameters zero (out = x)", of, x)
cmp("   layer path, same", ol, x)
of2, _ = tail_both(b2k)
print("run-to-run equal:", torch.equal(of, of2))
d = (of.float() - x.float()).abs().reshape(-1, 80)
bad = (d > 0).nonzero()
print("mismatching (token, channel) pairs:", bad[:30].tolist(), "total", bad.shape[0])

# ---- aggregator: one launch vs dwconv2d + gma_pointwise -----------------------------------------------------------------------------
with torch.no_grad():
    agg = blk.att.aggregator
    for n_ in (agg.norm0, agg.norm1, agg.norm2, agg.norm3):
        n_.running_mean.normal_(0, 0.2); n_.running_var.uniform_(0.5, 1.5)
    ops.invalidate_caches(blk)
    for shape in ((2, 24, 40), (1, 37, 29), (1, 16, 32)):
        qkv_t = torch.randn(*shape, 240, device=dev).to(torch.bfloat16)
        ops.FUSE_GMA = True
        qf, lf = agg._run(qkv_t.reshape(*shape, 15, 16).permute(3, 0, 1, 2, 4).contiguous())
        ops.FUSE_GMA = False
        ql, ll = agg._run(qkv_t)
        ops.FUSE_GMA = True
        qf = from_planar(qf).reshape(*ql.shape)
        cmp(f"aggregate qkvp {shape}", qf, ql)
        cmp(f"aggregate loc  {shape}", lf, ll)
        for gidx in range(4):
            d = (qf.float() - ql.float())[..., 16 * gidx:16 * gidx + 16].abs().max().item()
            print(f"   group {gidx}: max|diff| {d:.4g}")

with torch.no_grad():
    for shape in ((2, 24, 40), (1, 37, 29)):
        qp = torch.randn(*shape, 3, 64, device=dev).to(torch.bfloat16)
        ops.FUSE_GMA = True
        cf = from_planar(blk.att.crpe._conv_v(to_planar(qp.reshape(*shape, 192))))
        ops.FUSE_GMA = False
        cl = blk.att.crpe._conv_v(qp)
        ops.FUSE_GMA = True
        cmp(f"crpe {shape}", cf, cl)
