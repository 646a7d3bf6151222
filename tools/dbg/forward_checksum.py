import os, sys, hashlib
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/realcamnet_amd") else os.environ["GRAFT_REPO_ROOT"])
import torch
import realcamnet_amd as M
from realcamnet_amd import ops
torch.manual_seed(0)
for cls, dt in ((M.LiteISPNet_GFM_LSC_GMA, torch.bfloat16), (M.LiteISPNet, torch.float32), (M.LiteISPNet, torch.bfloat16)):
    net = cls().eval().to("cuda", dt)
    g = torch.Generator(device="cuda").manual_seed(1234)
    B, H2, W2 = 2, 544, 960
    mosaic = torch.rand(B, 1, H2, W2, generator=g, device="cuda").to(dt)
    coord = ops.make_coord(B, H2 // 2, W2 // 2, device="cuda", dtype=dt)
    with torch.no_grad():
        y = net.forward_mosaic(mosaic, None, coord)
    torch.cuda.synchronize()
    print(cls.__name__, dt, hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest()[:16])
