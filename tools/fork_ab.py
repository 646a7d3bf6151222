import sys, time, torch
sys.path.insert(0, ".")
import realcamnet_amd as M
from realcamnet_amd import ops
torch.manual_seed(0)
net = M.raw2bit.raw_compression_tcm_final().eval().to("cuda", torch.bfloat16)
g = torch.Generator(device="cuda").manual_seed(4321)
for frames in (8, 1):
    mosaic = torch.rand(frames, 1, 2160, 3840, generator=g, device="cuda").to(torch.bfloat16)
    coord = ops.make_coord(frames, 1080, 1920, device="cuda", dtype=torch.bfloat16)
    outs = {}
    for depth in (1, 3, 1, 3):
        ops.FORK_DEPTH = depth
        with torch.no_grad():
            for _ in range(2): o = net.forward_mosaic(mosaic, None, coord)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): o = net.forward_mosaic(mosaic, None, coord)
            torch.cuda.synchronize(); el = (time.perf_counter() - t0) / 4
        outs[depth] = o
        print(f"frames {frames} fork depth {depth}: {1e3 * el:.2f} ms  {frames * 2160 * 3840 / 1e6 / el:.1f} MP/s", flush=True)
    same = all(torch.equal(outs[1][k], outs[3][k]) for k in ("x_hat", "y"))
    print("bit-identical outputs:", same)
