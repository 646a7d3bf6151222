#!/usr/bin/env python3
"""Where the RAW codec's forward goes, phase by phase (HIP events): analysis transform, hyper-prior, slice loop, synthesis.
   python tools/codec_phase_bench.py [--frames 4]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import realcamnet_amd.raw2bit as RB
import realcamnet_amd.tcm as T
from realcamnet_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=4); ap.add_argument("--one-stream", action="store_true")
a = ap.parse_args()
ops.BRANCH_STREAMS = not a.one_stream
torch.manual_seed(0)
g = torch.Generator().manual_seed(1)
H, W, B, dt = 1152, 1920, a.frames, torch.bfloat16
m = RB.raw_compression_tcm_final().eval().to("cuda", dt)
raw = torch.rand(B, 4, H, W, generator=g).to("cuda", dt)
cond = torch.rand(B, 4, 256, 256, generator=g).to("cuda", dt)
coord = ops.make_coord(B, H, W, "cuda", dt)
marks = []


def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))


def run():
    marks.clear()
    with torch.no_grad():
        mark("start")
        an, cn = ops.to_nhwc(raw, dtype=dt), ops.to_nhwc(coord, dtype=dt)
        y, local, lsc = m._analysis(an, cond, cn); mark("analysis (ingest layout, lsc, prior, hycond, g_a)")
        z = m.h_a._nhwc(y); z_hat, z_lik = m.entropy_bottleneck._nhwc(z); mark("h_a + entropy bottleneck")
        ls, lm = T._hyper_synthesis(m, z_hat); mark("h_scale_s | h_mean_s")
        per = y.shape[-1] // m.num_slices
        ys = []
        for i in range(m.num_slices):
            ms, mu, sc = T._slice_params(m, i, lm, ls, ys)
            yh, lik = m.gaussian_conditional._nhwc(ops.channel_slice(y, i * per, per), sc, mu)
            ys.append(T._refine(m, i, ms, yh))
        mark("slice loop (5 slices)")
        out = T._synthesis_nchw(m, ops.channel_concat(ys)); mark("g_s (x_hat written as NCHW by the closing shuffle)")
    torch.cuda.synchronize()


for _ in range(3):
    run()
tot = marks[0][1].elapsed_time(marks[-1][1])
for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
    print(f"{n1:55s} {e0.elapsed_time(e1):8.2f} ms")
print(f"{'total':55s} {tot:8.2f} ms   (B={B}, two streams: {ops.BRANCH_STREAMS})")

# ---- finer: every top-level stage of the analysis transform and of g_s
if not a.one_stream:
    def timed(label, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record()
        fine.append((label, e0, e1))
        return out

    for rep in range(2):
        fine = []
        with torch.no_grad():
            an, cn = ops.to_nhwc(raw, dtype=dt), ops.to_nhwc(coord, dtype=dt)
            lsc = timed("lsc chain (2 -> 128, 4 layers)", lambda: m.lsc._nhwc(cn))
            vec = timed("colour prior", lambda: m.classifier._vec(cond))
            local = timed("HybridConditionModule", lambda: m.local_condition._nhwc(an))
            fea = timed("conv_first * (lsc + 1)", lambda: m.conv_first._nhwc(an, mul_plus1=lsc))
            fea = timed("conv_down (RBWS s2)", lambda: m.conv_down._nhwc(fea))
            for k, (gfm, blocks, down, c) in enumerate(((m.gfm1, m.m_down1, m.m_down1_down, local[0]), (m.gfm2, m.m_down2, m.m_down2_down, local[1]),
                                                         (m.gfm3, m.m_down3, m.m_down3_down, local[2]))):
                fea = timed(f"stage {k + 1}: Res_GFM", lambda: gfm[0]._nhwc((fea, vec))[0])
                for j, blk in enumerate(blocks):
                    fea = timed(f"stage {k + 1}: ConvTransBlock_mzj {j} @ {fea.shape[1]}x{fea.shape[2]}", lambda: blk._nhwc((fea, c))[0])
                fea = timed(f"stage {k + 1}: down", lambda: down._nhwc(fea))
            t = torch.randn(B, H // 16, W // 16, 320, device="cuda").to(dt)
            for j, mod in enumerate(m.g_s):
                t = timed(f"g_s[{j}] {type(mod).__name__} -> {t.shape[1]}x{t.shape[2]}", lambda: mod._nhwc(t))
        torch.cuda.synchronize()
    for label, e0, e1 in fine:
        print(f"{label:60s} {e0.elapsed_time(e1):8.2f} ms")
