"""Run-to-run bitwise stability of the whole fused GroupMix block (gma_in conv, cpe, qkv_aggregate, crpe, kv, tail<192> with the out conv) and of the lens-shading chain under
load: N forwards on the cfg3-size input, every output compared with the first.  (The tail's ISA has accumulator chains revisited one MFMA later: tools/mfma_hazard_scan.py.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import torch
import realcamnet_amd as M
from realcamnet_amd import ops
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
net = M.LiteISPNet_GFM_LSC_GMA().eval().to("cuda", torch.bfloat16)
d1 = torch.randn(8, 544, 960, 192, device="cuda").to(torch.bfloat16)
a = torch.rand(8, 1088, 1920, 4, device="cuda").to(torch.bfloat16)
coord = ops.to_nhwc(ops.make_coord(8, 1088, 1920, device="cuda", dtype=torch.bfloat16), dtype=torch.bfloat16)
with torch.no_grad():
    for name, fn in (("GroupMix block (_refine_d1)", lambda: net._refine_d1(d1)), ("lens-shading head", lambda: ops.lsc_chain(net.lsc, coord, net.head, a))):
        ref = fn().clone()
        bad, elems = 0, 0
        for i in range(n):
            out = fn()
            k = int((out.view(torch.int16) != ref.view(torch.int16)).sum())
            bad += k > 0; elems += k
        torch.cuda.synchronize()
        print(f"{name}: {bad} of {n} launches differ from the first ({elems} elements in all)")
