import sys; sys.path.insert(0, "/root/repo")
import torch, realcamnet_amd.raw2bit as RB
from realcamnet_amd import ops
torch.manual_seed(0)
m = RB.HybridConditionModule(out_channels=64, init_mid_channels=16).to("cuda", torch.bfloat16).eval()
a = torch.rand(4, 1152, 1920, 4, device="cuda").bfloat16()
ev = []
def T(name, fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = fn(); e1.record(); ev.append((name, e0, e1)); return out
def run():
    ev.clear()
    with torch.no_grad():
        x1 = T("in_conv 4->16 @full", lambda: m.in_conv._nhwc(a))
        x2 = T("enc_1 (s2 16->32, 32->32) @1/2", lambda: m.enc_1._nhwc(x1))
        x3 = T("enc_2 @1/4", lambda: m.enc_2._nhwc(x2))
        x4 = T("enc_3 @1/8", lambda: m.enc_3._nhwc(x3))
        y = T("dec_1 @1/4", lambda: m.dec_1._nhwc(x4, x3))
        y = T("dec_2 @1/2", lambda: m.dec_2._nhwc(y, x2))
        y = T("dec_3 @full (up 32, conv 32->16, cat, conv 32->16)", lambda: m.dec_3._nhwc(y, x1))
        y = T("out_conv 16->64 @full", lambda: m.out_conv._nhwc(y))
        s2d = T("space_to_depth(y)", lambda: ops.space_to_depth2(y))
        T("CondNet1", lambda: m._cond(m.CondNet1, y, s2d)); T("CondNet2", lambda: m._cond(m.CondNet2, y, s2d)); T("CondNet3", lambda: m._cond(m.CondNet3, y, s2d))
    torch.cuda.synchronize()
for _ in range(3): run()
for n, e0, e1 in ev: print(f"{n:60s} {e0.elapsed_time(e1):7.3f} ms")
