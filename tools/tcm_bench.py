#!/usr/bin/env python3
"""Codec forward (likelihood path) at the cfg5 size, random-init weights, bf16 or fp32 (not the headline bench: see bench.py).
   python tools/tcm_bench.py [--model tcm|raw] [--frames 1] [--dtype bf16] [--steps 5]
   tcm: TCM.forward on sRGB 3840x2160 padded to 2176 rows;  raw: raw_compression_tcm_final.forward on the packed RAW of a 4K
   mosaic (4 x 1080 x 1920 padded to 1152 rows) + cond 256x256 + coord, producing the 2304 x 3840 sRGB."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import realcamnet_amd.tcm as T

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--model", default="tcm")
ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--height", type=int, default=0); ap.add_argument("--width", type=int, default=0)
ap.add_argument("--graph", action="store_true", help="capture the forward in a HIP graph and time replays")
ap.add_argument("--cpu-baseline", action="store_true", help="also time the CPU oracle (fp32) on a 512x512-mosaic sample with the same weights")
a = ap.parse_args()
dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
g = torch.Generator().manual_seed(1)
torch.manual_seed(0)
if a.model == "raw":
    import realcamnet_amd.raw2bit as RB
    a.height, a.width = a.height or 1152, a.width or 1920
    m = RB.raw_compression_tcm_final().eval()
    x = [torch.rand(a.frames, 4, a.height, a.width, generator=g).to("cuda", dt), torch.rand(a.frames, 4, 256, 256, generator=g).to("cuda", dt),
         torch.stack(torch.meshgrid(torch.linspace(-1, 1, a.height), torch.linspace(-1, 1, a.width), indexing="ij"))[None].expand(a.frames, -1, -1, -1).contiguous().to("cuda", dt)]
    label = "megapixels/sec packed RAW -> raw_compression_tcm_final.forward (likelihood path), per 4K mosaic frame"
else:
    a.height, a.width = a.height or 2176, a.width or 3840
    m = T.TCM().eval()
    x = torch.rand(a.frames, 3, a.height, a.width, generator=g).to("cuda", dt)
    label = "megapixels/sec sRGB -> TCM.forward (likelihood path)"
cpu = None                               # weights: seed-0 default initialisation of the mirror modules
if a.cpu_baseline:                       # oracle on the host cores, bounded sample (the codec's CPU restatement, kind "port")
    import liteisp_oracle as LO, raw2bit_oracle as RO, tcm_oracle as TO   # the oracle is used in this leg only, as in bench.py
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    with torch.no_grad():
        if a.model == "raw":
            xs = [torch.rand(1, 4, 256, 256, generator=g), torch.rand(1, 4, 256, 256, generator=g), LO.make_coord(1, 256, 256)]
            f = lambda: RO.raw_compression_tcm_final(sd, xs)
            px = 512 * 512
        else:
            xs = torch.rand(1, 3, 512, 512, generator=g)
            f = lambda: TO.tcm_forward(sd, xs)
            px = 512 * 512
        f(); t0 = time.perf_counter(); f(); tc = time.perf_counter() - t0
    cpu = {"value": round(px / 1e6 / tc, 4), "unit": "MP/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"1 frame, 512x512 output pixels, fp32, oracle/{'raw2bit' if a.model == 'raw' else 'tcm'}_oracle.py, {tc:.1f} s"}
m = m.to("cuda", dt)
with torch.no_grad():
    for _ in range(a.warmup):
        out = m(x)
    torch.cuda.synchronize()
    if a.graph:                          # all launches go to torch's current stream and allocate through its caching allocator
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = m(x)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(gr):
            out = m(x)
        gr.replay(); torch.cuda.synchronize()
        step = gr.replay
    else:
        step = lambda: m(x)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step()
        out = r if r is not None else out
    torch.cuda.synchronize()
t = (time.perf_counter() - t0) / a.steps
bpp = float((-torch.log2(out["likelihoods"]["y"])).sum() + (-torch.log2(out["likelihoods"]["z"])).sum()) / (a.frames * a.height * a.width)
print(json.dumps({"metric": label, "model": a.model, "value": round(a.frames * 3840 * 2160 / 1e6 / t, 2), "unit": "MP/s",
                  "ms_per_step": round(t * 1e3, 2), "frames": a.frames, "dtype": a.dtype, "hip_graph": bool(a.graph), "padded": [a.height, a.width],
                  "bits_per_pixel_random_weights": round(bpp, 4), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2), "cpu_baseline": cpu}))
