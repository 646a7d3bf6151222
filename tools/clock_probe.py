import os, sys, time, subprocess, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import networks as N, ops
CIN, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (48, 1088, 1920)))
c = N.Conv2d(CIN, CIN, 3, 1, 1).to("cuda", torch.bfloat16)
x = torch.rand(8, H, W, CIN, device="cuda").to(torch.bfloat16)
for _ in range(2): ops.conv2d(x, c, act="relu")
torch.cuda.synchronize()
def smi(tag):
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True)
    keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "junction"))]
    print(tag, " | ".join(keep), flush=True)
smi("idle")
time.sleep(2)
evs = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
evs[0].record()
for i in range(12):
    for _ in range(25): ops.conv2d(x, c, act="relu")
    evs[i + 1].record()
th = threading.Thread(target=lambda: smi("load")); th.start()
torch.cuda.synchronize(); th.join()
print("ms/iter per block of 25:", [round(evs[i].elapsed_time(evs[i + 1]) / 25, 3) for i in range(12)])
time.sleep(3)
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); ops.conv2d(x, c, act="relu"); e1.record(); torch.cuda.synchronize()
print("single after 3 s idle:", round(e0.elapsed_time(e1), 3), "ms")
y = torch.empty_like(x)
evs[0].record()
for _ in range(200): y.copy_(x)
evs[1].record(); torch.cuda.synchronize()
print("copy ms:", round(evs[0].elapsed_time(evs[1]) / 200, 3))
