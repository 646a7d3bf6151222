import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops
torch.manual_seed(0)
blk = M.GMA_Block(80, 8).to("cuda", torch.bfloat16).eval()
x = torch.randn(8, 544, 960, 80, device="cuda").to(torch.bfloat16)
with torch.no_grad():
    from realcamnet_amd import ops
    R = torch.ops.realcam
    for _ in range(2):
        got = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
        wq, bq = ops.packed_chain(blk.att.qkv)
        qkv = R.gma_ln_qkv(x, wq, bq, ops.f32_param(blk.norm1, "weight"), ops.f32_param(blk.norm1, "bias"), 1e-5)
        want = blk.att.aggregator._run_fused(qkv)
torch.cuda.synchronize()
