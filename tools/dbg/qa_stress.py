"""Run-to-run bitwise stability of rc_gma_qkv_aggregate under load: N launches on the cfg3-size input, every output compared with the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
torch.manual_seed(0)
blk = M.GMA_Block(80, 8).to("cuda", torch.bfloat16).eval()
x = torch.randn(8, 544, 960, 80, device="cuda").to(torch.bfloat16)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with torch.no_grad():
    ref = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
    bad = 0
    for i in range(n):
        out = blk.att.aggregator._run_front(x, blk.norm1, blk.att.qkv)
        for r_, o_ in zip(ref, out):
            if not torch.equal(r_, o_):
                bad += 1
                print("launch", i, "differs:", int((r_.float() != o_.float()).sum()))
torch.cuda.synchronize()
print(f"{n} launches, {bad} mismatching tensors")
