#!/usr/bin/env python3
"""RAW->sRGB throughput benchmark (BASELINE.json metric: megapixels/sec at 4K; PSNR vs CPU reference).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: re-executes itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (RAW ingest: Bayer unshuffle + pad + bilinear cond resize -> LiteISPNet_GFM_LSC_GMA ->
cropped sRGB) over one batch of `--frames` synthetic 4K mosaics per GPU (weak scaling: cfg3 at N=1, cfg4 = 64 frames at
N=8).  Inputs are resident in HBM before the timed region.  At N>1 every rank's sRGB frames are all-gathered to all ranks
over RCCL/xGMI on a side stream, overlapped with the next step's forward; the last gather is inside the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch

PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(name, sd, frame_hw=(2160, 3840), budget_s=20.0):
    """Time the CPU oracle (kind 'port': oracle/liteisp_oracle.py, fp32) on the host's cores on a bounded
    sample of the same workload; returns (dict, (mosaic, cond, coord, ref_out)) for the PSNR leg.
    The thread count is probed (oneDNN collapses when a 256-thread pool is thrown at small convs)."""
    import liteisp_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    g = torch.Generator().manual_seed(1234)

    def sample(h2, w2):
        mosaic = torch.rand(1, 1, h2, w2, generator=g)
        cond = O.raw_ingest(mosaic)[1]                    # cond = bilinear resize of the packed RAW to 256x256
        coord = O.make_coord(1, h2 // 2, w2 // 2)
        return mosaic, cond, coord

    def run(mosaic, cond, coord):
        with torch.no_grad():
            t0 = time.perf_counter()
            out = O.run_padded(name, sd, O.bayer_unshuffle(mosaic), cond, coord)
            return time.perf_counter() - t0, out

    probe = sample(256, 256)
    best_t, threads = None, 1
    for n in (8, 16, 32, 64, 128):
        if n > avail and n != 8:
            break
        torch.set_num_threads(min(n, avail))
        run(*probe)                                       # warm-up for this pool size
        t, _ = run(*probe)
        if best_t is None or t < best_t:
            best_t, threads = t, min(n, avail)
        if t > 3.0:
            break
    torch.set_num_threads(threads)
    rate = 256 * 256 / best_t                             # output pixels / s at small size
    # largest 16:9 crop of a 4K frame that fits the budget (big images run somewhat slower per pixel)
    fh, fw = frame_hw
    frac = min(1.0, (rate * budget_s * 0.5) / (fh * fw))
    scale = frac ** 0.5
    h2 = max(256, int(fh * scale) // 32 * 32)
    w2 = max(256, int(fw * scale) // 32 * 32)
    if scale >= 1.0:
        h2, w2 = fh, fw
    mosaic, cond, coord = sample(h2, w2)
    t, out = run(mosaic, cond, coord)
    mp = h2 * w2 / 1e6
    info = {"value": round(mp / t, 4), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": f"1 frame {w2}x{h2} mosaic ({mp:.2f} MP) fp32, oracle/liteisp_oracle.py, {t:.1f} s, "
                      f"{threads} of {avail} host threads (probed)"}
    return info, (mosaic, cond, coord, out)


# algorithmic FLOPs per padded output pixel (2 MAC of every conv / linear of the REFERENCE net, hook-counted there), and the GroupMix block's
# 25.3 C^2 + 176 C FLOP per token (C = 80) plus the build's two 1x1 projections 192 -> 80 -> 192 around it.  `flops_per_step` (executed) is lower where a fold
# removed work (the tail as one 5x5 convolution): both are reported so that the fold's credit is visible and the denominator is not silently moving.
ALGO_FLOP_PER_PX = {"LiteISPNet": 716896.0, "LiteISPNet_GFM_LSC": 622084.0, "LiteISPNet_GFM_LSC_GMA": 622084.0, "ISPUNet_GFM_LSC": 378922.0}
GMA_FLOP_PER_TOKEN = 25.3 * 80 * 80 + 176 * 80 + 2 * 2 * 192 * 80


def flops_algorithmic(model, frames, H2, W2):
    if model not in ALGO_FLOP_PER_PX:
        return None
    pad = 16                                              # packed RAW padded to a multiple of 16 (pad_to_multiple_of_16, LiteISP.py:84-128)
    hp, wp = -(-(H2 // 2) // pad) * pad, -(-(W2 // 2) // pad) * pad
    fl = ALGO_FLOP_PER_PX[model] * (2 * hp) * (2 * wp)
    if model.endswith("_GMA"):
        fl += GMA_FLOP_PER_TOKEN * (hp // 2) * (wp // 2)
    return frames * fl


class PowerSampler:
    """One rocm-smi sample (socket power, shader clock) taken INSIDE the timed region, from a side thread: the conv kernels run this part at its board power
    cap with the clock throttled (profiles/r05_power_wall.md), which is what bounds roofline.frac.  Diagnostics only: never fails the run."""

    def __init__(self, delay_s):
        import subprocess, threading
        self.out, self.t_done = {}, None

        def work():
            try:
                time.sleep(max(0.0, delay_s))
                r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower"], capture_output=True, text=True, timeout=20)
                self.t_done = time.perf_counter()
                for l in r.stdout.splitlines():
                    v = l.split(":")[-1].strip()
                    if "Max Graphics Package Power" in l:
                        self.out["cap_W"] = float(v)
                    elif "Graphics Package Power" in l and "Max" not in l:
                        self.out["socket_W"] = float(v)
                    elif "sclk" in l and "(" in l:
                        self.out["sclk_MHz"] = int(l.split("(")[-1].split("Mhz")[0])
            except Exception as e:                        # noqa: BLE001 -- diagnostics only
                self.out["error"] = str(e)[:60]
        self.th = threading.Thread(target=work, daemon=True)
        self.th.start()

    def result(self, t_region_end):
        self.th.join(timeout=25)
        if not self.out or "error" in self.out:
            return None
        self.out["sampled_inside_timed_region"] = bool(self.t_done is not None and self.t_done <= t_region_end)
        return self.out


def load_pmc_summary():
    """The newest profiles/rNN_pmc_bench.json whose `source_digest` equals this build's kernel-source digest, else None."""
    from realcamnet_amd import build as rb
    dig = rb.source_digest()
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted((f for f in os.listdir(pdir) if f.endswith("_pmc_bench.json")), reverse=True):
        try:
            with open(os.path.join(pdir, name)) as f:
                pmc = json.load(f)
        except (OSError, ValueError):
            continue
        if pmc.get("source_digest") == dig:
            pmc["_file"] = name
            return pmc
    return None


from torch.utils._python_dispatch import TorchDispatchMode


class _LaunchCounter(TorchDispatchMode):
    """Counts dispatched ops of one forward: every realcam:: op is one C-ABI kernel launch; ATen ops that touch data are counted beside them."""
    def __init__(self):
        super().__init__()
        self.realcam = 0
        self.aten = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name.startswith("realcam."):
            self.realcam += 1
        elif not any(k in name for k in ("view", "reshape", "permute", "transpose", "slice", "select", "detach", "alias", "expand", "squeeze",
                                         "unsqueeze", "empty", "as_strided", "t.default", "size", "stride", "unbind", "split", "chunk", "_unsafe_view")):
            self.aten += 1
        return func(*args, **(kwargs or {}))


def codec_leg(dev, dt, H2, W2, frames=8, steps=4, warmup=2, with_psnr=True):
    """SURVEY cfg5's one-GPU shape inside the default run: raw_compression_tcm_final.forward_mosaic (models/raw2bit.py:1766-1855, likelihood
    path, no entropy coder) on `frames` 4K mosaics, timed after a warm-up that also packs the weights.  Returned as the extra key
    `codec_leg`; the contract keys of the headline line are untouched."""
    import realcamnet_amd as M
    from realcamnet_amd import ops
    torch.manual_seed(0)
    net = M.raw2bit.raw_compression_tcm_final().eval()
    sd_cpu = {k: v.clone() for k, v in net.state_dict().items()} if with_psnr else None
    net = net.to(device=dev, dtype=dt)
    g = torch.Generator(device=dev).manual_seed(4321)
    mosaic = torch.rand(frames, 1, H2, W2, generator=g, device=dev).to(dt)
    coord = ops.make_coord(frames, H2 // 2, W2 // 2, device=dev, dtype=dt)

    def step():
        with torch.no_grad():
            return net.forward_mosaic(mosaic, None, coord)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    ops.prof_enable(True)
    step()
    n_conv, conv_ms, conv_flops = ops.prof_collect()
    ops.prof_enable(False)
    launches = None
    try:
        with _LaunchCounter() as lc:
            step()
        launches = {"realcam_ops": lc.realcam, "aten_ops": lc.aten}
    except Exception as e:                                   # counting is diagnostics only
        launches = {"error": str(e)[:80]}
    torch.cuda.synchronize()
    leg = {"metric": f"megapixels/sec RAW mosaic {W2}x{H2} -> raw_compression_tcm_final.forward (likelihood path)",
           "value": round(frames * H2 * W2 / 1e6 / el, 2), "unit": "MP/s", "ms_per_step": round(1e3 * el, 3), "frames": frames, "steps": steps,
           "warmup": warmup, "launches_per_forward": launches, "conv_launches": int(n_conv), "conv_ms_per_step": round(conv_ms, 3),
           "flops_necessary": conv_flops, "conv_TFLOPs": round(conv_flops / (conv_ms * 1e-3) / 1e12, 1) if conv_ms > 0 else 0.0}
    if with_psnr:
        import liteisp_oracle as O                          # the oracle: checker only
        import raw2bit_oracle as RO
        gc = torch.Generator().manual_seed(1234)
        S = 1024                                            # bounded: one 1024 x 1024 mosaic (~3 s of CPU work)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        with torch.no_grad():
            mos = torch.rand(1, 1, S, S, generator=gc)
            raw, cond = O.raw_ingest(mos); co = O.make_coord(1, S // 2, S // 2)
            ref = RO.raw_compression_tcm_final(sd_cpu, [raw, cond, co])
            y = net([raw.to(dev, dt), cond.to(dev, dt), co.to(dev, dt)])
        leg["psnr_y"] = round(O.psnr(y["para"]["y"].float().cpu(), ref["para"]["y"]), 2)
        leg["psnr_x_hat"] = round(O.psnr(y["x_hat"].float().cpu(), ref["x_hat"]), 2)
        leg["psnr_sample"] = f"1 frame, {S}x{S} mosaic, fp32 CPU oracle (oracle/raw2bit_oracle.py)"
    del net, mosaic, coord
    torch.cuda.empty_cache()
    return leg


def cfg2_leg(dev, steps=20, warmup=5, with_psnr=True):
    """SURVEY cfg2 inside the default run: one 1920 x 1080 mosaic through LiteISPNet in fp32 (the configuration whose 3x3 layers take the Winograd F(2x2,3x3)
    kernel, csrc/wino.hip), timed like the headline (synchronize on both sides, K steps after W warm-up steps), with the same forward's PSNR against the
    fp32 CPU oracle on the whole frame (~4 s of CPU work).  Returned as the extra key `cfg2_leg`; the contract keys of the headline line are untouched."""
    import realcamnet_amd as M
    from realcamnet_amd import ops
    H2, W2 = 1080, 1920
    torch.manual_seed(0)
    net = M.LiteISPNet().eval()
    sd_cpu = {k: v.clone() for k, v in net.state_dict().items()} if with_psnr else None
    net = net.to(device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev).manual_seed(1234)
    mosaic = torch.rand(1, 1, H2, W2, generator=g, device=dev)
    coord = ops.make_coord(1, H2 // 2, W2 // 2, device=dev, dtype=torch.float32)

    def step():
        with torch.no_grad():
            return net.forward_mosaic(mosaic, None, coord)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    ops.prof_enable(True)
    y = step()
    n_conv, conv_ms, conv_flops = ops.prof_collect()
    ops.prof_enable(False)
    leg = {"metric": "megapixels/sec RAW->sRGB at 1920x1080", "value": round(H2 * W2 / 1e6 / el, 2), "unit": "MP/s", "ms_per_step": round(1e3 * el, 3), "steps": steps, "warmup": warmup,
           "dtype": "f32", "workload": "cfg2: 1920x1080 Bayer mosaic -> unshuffle+pad16 -> LiteISPNet -> sRGB 1920x1080, 1 frame, fp32",
           "conv_launches": int(n_conv), "conv_ms_per_step": round(conv_ms, 3), "flops_executed": conv_flops,
           "flops_algorithmic": 4 * 716896.0 * (H2 // 2 + (-(H2 // 2)) % 16) * (W2 // 2 + (-(W2 // 2)) % 16),        # 716 896 FLOP per OUTPUT pixel = 4x per packed pixel of the padded frame
           "winograd": bool(ops.WINOGRAD)}
    leg["frac_algorithmic_of_fp32_mfma_peak"] = round(leg["flops_algorithmic"] / el / 157.3e12, 4)
    if with_psnr:
        import liteisp_oracle as O                          # the oracle: checker only
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        with torch.no_grad():
            packed, cond = O.raw_ingest(mosaic.cpu())
            ref = O.run_padded("LiteISPNet", sd_cpu, packed, cond, O.make_coord(1, H2 // 2, W2 // 2))
        leg["psnr_db_vs_cpu_fp32"] = round(O.psnr(y.float().cpu(), ref), 2)
    del net, mosaic, coord
    torch.cuda.empty_cache()
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=8, help="frames per GPU per step (cfg3: 8)")
    ap.add_argument("--height", type=int, default=2160, help="mosaic height")
    ap.add_argument("--width", type=int, default=3840, help="mosaic width")
    ap.add_argument("--dtype", choices=["bf16", "f32"], default="bf16")
    ap.add_argument("--model", default="LiteISPNet_GFM_LSC_GMA", choices=["LiteISPNet_GFM_LSC_GMA", "LiteISPNet_GFM_LSC", "LiteISPNet", "ISPUNet_GFM_LSC", "raw_compression_tcm_final"],
                    help="default = cfg3: the flagship net plus one GroupMix GMA_Block(80,8) at H/2 (build-defined placement); "
                         "raw_compression_tcm_final = the RAW codec's forward (likelihood path), SURVEY cfg5's codec leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sample-power", action="store_true", help="diagnostic: one rocm-smi sample (socket W, shader clock) from a side thread INSIDE the timed region (rank 0)")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the all-gather of the output frames (replicas only)")
    ap.add_argument("--no-codec-leg", action="store_true", help="default cfg3 run: skip the extra codec_leg key (raw_compression_tcm_final at 4 frames)")
    ap.add_argument("--layer-by-layer-tail", action="store_true", help="A/B: the tail as the module list's two launches (ops.FOLD_TAIL = False) instead of the folded 5x5 conv")
    ap.add_argument("--staged-gate", action="store_true", help="A/B: round 3's RCAB schedule (CALayer gate folded into the next conv's staging, ops.EARLY_GATE = False)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: the launch / rendezvous / barrier / max-over-ranks skeleton with a trivial CPU step over gloo; prints a line marked as a self-test (tests only)")
    args = ap.parse_args()

    from realcamnet_amd import shard
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:     # plain `python bench.py --gpus N`: become the N-rank torchrun job (does not return)
        shard.relaunch_under_torchrun(os.path.abspath(__file__), sys.argv[1:], args.gpus)
    if args.launcher_selftest:
        return launcher_selftest(args)

    import realcamnet_amd as M
    from realcamnet_amd import ops
    ops.FOLD_TAIL, ops.EARLY_GATE = not args.layer_by_layer_tail, not args.staged_gate

    rank, world, local_rank = shard.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32

    torch.manual_seed(0)                                   # random-init weights of the named architecture
    codec = args.model == "raw_compression_tcm_final"
    net = (M.raw2bit.raw_compression_tcm_final() if codec else getattr(M, args.model)()).eval()
    sd_cpu = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(device=dev, dtype=dt)

    B, H2, W2 = args.frames, args.height, args.width
    total_frames = B * world                               # weak scaling: rank r owns frames [r*B, (r+1)*B)
    s, e = shard.frame_shard(total_frames, rank, world)
    assert e - s == B
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    mosaic = torch.rand(B, 1, H2, W2, generator=g, device=dev).to(dt)
    coord = ops.make_coord(B, H2 // 2, W2 // 2, device=dev, dtype=dt)
    cond = None                                            # forward_mosaic's ingest kernel makes it: the packed RAW resized to 256x256

    def step():
        with torch.no_grad():
            return net.forward_mosaic(mosaic, cond, coord)

    if codec:
        return bench_codec(args, net, sd_cpu, step, (mosaic, cond, coord), rank, world, dev, dt)

    import torch.distributed as dist
    gather = shard.OverlappedGather(total_frames) if (dist.is_available() and dist.is_initialized() and not args.no_gather) else None
    est = 0.0
    for i in range(args.warmup):
        if i == args.warmup - 1:                           # the last warm-up step is timed on its own: it sizes the power sample's delay (no extra forward)
            torch.cuda.synchronize()
            tw = time.perf_counter()
        out = step()
        if gather is not None:
            gather.submit(out)
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if args.warmup > 0:
        est = (time.perf_counter() - tw) * args.steps
    shard.barrier()
    torch.cuda.synchronize()
    # --sample-power: one rocm-smi sample inside the timed region (a subprocess on rank 0 + SMI queries against the GPU being timed: a diagnostic that
    # perturbs what it measures, so it is OFF for the recorded metric -- ADVICE r5)
    sampler = PowerSampler(0.1 * est) if (args.sample_power and rank == 0 and est > 0.25) else None     # rocm-smi itself takes ~0.2 s: only runs long enough to contain it
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
        if gather is not None:
            gather.submit(out)                             # side stream: overlaps the next step's forward
    gathered = gather.wait() if gather is not None else out
    torch.cuda.synchronize()
    shard.barrier()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    elapsed = shard.max_over_ranks(t_end - t0, dev)
    power = sampler.result(t_end) if sampler is not None else None
    assert out.shape == (B, 3, H2, W2) and gathered.shape == (total_frames if gather is not None else B, 3, H2, W2)
    if gather is not None:                                 # the gathered payload is this rank's own frames where they belong
        assert torch.equal(gathered[s:e], out)

    # dominant kernel (MFMA conv): HIP-event time of every launch on its stream, one extra step
    ops.prof_enable(True)
    step()
    n_launch, conv_ms, conv_flops = ops.prof_collect()
    rows = ops.prof_rows()
    ops.prof_enable(False)

    if rank != 0:
        return
    cfg_name = ("cfg3" if world == 1 else "cfg4") if (args.model.endswith("GMA") and (H2, W2) == (2160, 3840)) else "custom"
    if args.model == "LiteISPNet" and args.dtype == "f32" and (H2, W2, B) == (1080, 1920, 1):
        cfg_name = "cfg2"
    mp_per_step = total_frames * H2 * W2 / 1e6
    value = mp_per_step * args.steps / elapsed
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = PEAK_TFLOPS[args.dtype]
    res = {
        "metric": "megapixels/sec RAW->sRGB at 4K" if (H2, W2) == (2160, 3840) else f"megapixels/sec RAW->sRGB at {W2}x{H2}", "value": round(value, 2), "unit": "MP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (uniform[0,1) mosaics, seed-0 random-init weights)",
        "config": {"workload": f"{cfg_name}: {W2}x{H2} Bayer mosaic -> unshuffle+pad16 -> {args.model} -> sRGB {W2}x{H2}, "
                               f"{B} frames/GPU, {args.dtype} storage / fp32 accumulate",
                   "frames_per_gpu": B, "global_frames": total_frames, "parallelism": f"frame-shard x{world}",
                   "collective": (f"all_gather of the sRGB frames over RCCL, {out.numel() * out.element_size()} bytes per rank per step, on a side "
                                  "stream overlapped with the next step's forward") if gather is not None else "none",
                   **({"schedule": ("tail as two launches; " if args.layer_by_layer_tail else "") + ("CALayer gate staged in the next conv" if args.staged_gate else "")}
                      if (args.layer_by_layer_tail or args.staged_gate) else {})},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": None,
                     "kernel": "conv_mfma_kernel (all instantiations)", "launches_per_step": int(n_launch),
                     "kernel_ms_per_step": round(conv_ms, 3), "flops_per_step": conv_flops},
    }
    # what bounds what: the family's frac above is priced against the MFMA peak, but the layer shape with the most time is HBM-paced -- name it, with both
    # of its fractions (flat scalar keys: the driver's parser keeps those)
    rf = res["roofline"]
    fa = flops_algorithmic(args.model, B, H2, W2)
    if fa is not None:
        rf["flops_algorithmic"] = fa
        rf["frac_algorithmic_whole_step"] = round(fa / (elapsed / args.steps) / 1e12 / peak, 4)
        rf["flops_note"] = ("flops_per_step = executed by the conv launches (the folded 5x5 tail executes 4.6x fewer than the two convolutions it replaces; a Winograd F(2x2,3x3) "
                            "launch -- the fp32 3x3 layers with 64 | cout -- executes 4/9 of its layer's multiplications, so frac_algorithmic_whole_step can exceed frac and, in fp32, 1); "
                            "flops_algorithmic = SURVEY 8(d): reference net at the padded size + GroupMix block")
    if rows:
        d0 = rows[0]
        es = 2 if args.dtype == "bf16" else 4
        rf["dominant_kernel"] = f"conv {d0['ksize']}x{d0['ksize']} {d0['cin']}->{d0['cout']} ({d0['launches']} launches per step)"
        rf["dominant_ms_per_step"] = round(d0["ms"], 3)
        rf["dominant_tflops"] = round(d0["flops"] / (d0["ms"] * 1e-3) / 1e12, 1)
        rf["dominant_mfma_frac"] = round(d0["flops"] / (d0["ms"] * 1e-3) / 1e12 / peak, 4)
        rf["dominant_hbm_TBps"] = round(d0["bytes"] / (d0["ms"] * 1e-3) / 1e12, 2)
        rf["dominant_hbm_frac_of_8TBps"] = round(d0["bytes"] / (d0["ms"] * 1e-3) / 8e12, 4)
        rf["dominant_bound"] = "hbm" if rf["dominant_hbm_frac_of_8TBps"] > rf["dominant_mfma_frac"] else "mfma"
        rf["by_shape_top4"] = "; ".join(f"{r['cin']}->{r['cout']} k{r['ksize']}: {r['ms']:.2f} ms, {r['flops'] / (r['ms'] * 1e-3) / 1e12:.0f} TF/s, "
                                        f"{r['bytes'] / (r['ms'] * 1e-3) / 1e12:.2f} TB/s" for r in rows[:4])
    if power is not None:
        res["power"] = power
        rf["power_note"] = (f"rocm-smi mid-run: socket {power.get('socket_W')} W (its reading averages over a window that includes the non-conv kernels) of a "
                            f"{power.get('cap_W')} W cap, sclk {power.get('sclk_MHz')} MHz of 2400; the conv kernels alone sit AT the cap (profiles/r05_power_wall.md)")
    # HBM traffic comes from PMC counters, which cannot be read inside a timed run: tools/pmc_bench.sh collects them in separate
    # rocprofv3 --pmc passes of this same default command and commits the summary under profiles/.  The summary carries the digest of
    # the kernel sources it was measured on; a file taken on other sources is ignored (traffic stays null) rather than replayed.
    pmc = load_pmc_summary() if (cfg_name == "cfg3" and args.dtype == "bf16" and B == 8) else None
    if pmc is not None:
        res["roofline"]["traffic"] = round(pmc["conv_kernels_all"]["hbm_bytes_per_dispatch"])
        res["roofline"]["traffic_note"] = ("HBM bytes per conv launch (mean over the step's conv launches), FETCH_SIZE x2 + WRITE_SIZE from "
                                           f"profiles/{pmc['_file']} (same kernel-source digest {pmc['source_digest'][:12]} as this build)")
        # the mean hides very different layers: also the three kernels that move the most, per launch and per forward (same PMC passes)
        fw = pmc.get("forwards", 3)
        top = sorted(((k, v["dispatches"], v["fetch_bytes_per_dispatch"] + v["write_bytes_per_dispatch"]) for k, v in pmc["kernels"].items()),
                     key=lambda t: -t[1] * t[2])[:3]
        res["roofline"]["traffic_by_kernel"] = [{"kernel": k.split("(")[0][-70:], "launches_per_forward": round(n / fw, 1), "hbm_bytes_per_launch": round(bpl),
                                                 "hbm_GB_per_forward": round(n * bpl / fw / 1e9, 1)} for k, n, bpl in top]
        res["roofline"]["traffic_top3"] = "; ".join(f"{k.split('(')[0].split('::')[-1][:40]} x{n / fw:.0f}: {n * bpl / fw / 1e9:.1f} GB per forward" for k, n, bpl in top)
        step_bytes = (pmc["all_kernels_total_bytes"]["fetch"] + pmc["all_kernels_total_bytes"]["write"]) / pmc.get("forwards", 3)
        res["hbm_whole_step"] = {"bytes_per_step": round(step_bytes), "achieved_TBps": round(step_bytes / (elapsed / args.steps) / 1e12, 2),
                                 "copy_rate_TBps": pmc.get("copy_rate_TBps", 5.9), "peak_TBps": 8.0}
    else:
        res["roofline"]["traffic_note"] = "null: no PMC summary under profiles/ was taken on this build's kernel sources (tools/pmc_bench.sh)"
    if world == 1 and cfg_name == "cfg3" and not args.no_codec_leg:
        del out, gathered
        torch.cuda.empty_cache()
        res["codec_leg"] = codec_leg(dev, dt, H2, W2, frames=args.frames, steps=4, with_psnr=not args.no_cpu_baseline)     # cfg5: the same frames per GPU as the headline (8)
        res["cfg2_leg"] = cfg2_leg(dev, with_psnr=not args.no_cpu_baseline)
    if world == 1 and not args.no_cpu_baseline:
        info, (m_c, c_c, co_c, ref) = cpu_baseline(args.model, sd_cpu, (H2, W2))
        with torch.no_grad():
            y = net.forward_mosaic(m_c.to(dev, dt), c_c.to(dev, dt), co_c.to(dev, dt))
        torch.cuda.synchronize()
        res["cpu_baseline"] = info
        import liteisp_oracle as O
        res["psnr_db_vs_cpu_fp32"] = round(O.psnr(y.float().cpu(), ref), 2)
    print(json.dumps(res), flush=True)


def launcher_selftest(args):
    """The multi-rank skeleton of main() without a GPU: rendezvous (gloo), frame shard, K trivial CPU steps with the all-gather of their output
    overlapped, barrier, max-over-ranks.  Rank 0 prints one JSON line whose metric says it is a self-test, never a measurement."""
    import torch.distributed as dist
    from realcamnet_amd import shard
    rank, world, _ = shard.init_distributed("gloo")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    B = args.frames
    s, e = shard.frame_shard(B * world, rank, world)
    x = torch.arange(s, e, dtype=torch.float32).view(B, 1, 1, 1).expand(B, 3, 4, 4).contiguous()
    gather = shard.OverlappedGather(B * world) if dist.is_initialized() else None
    shard.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = x * 2.0
        if gather is not None:
            gather.submit(out)
    got = gather.wait() if gather is not None else out
    shard.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0)
    ok = bool(torch.equal(got[:, 0, 0, 0], 2.0 * torch.arange(B * world, dtype=torch.float32)))
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no GPU work, not a measurement)", "n_gpus": world, "steps": args.steps, "gathered_ok": ok,
                          "launched_by": "self (torch.distributed.run re-exec)" if os.environ.get("TORCHELASTIC_RUN_ID") else "single process",
                          "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 3)}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def bench_codec(args, net, sd_cpu, step, inputs, rank, world, dev, dt):
    """--model raw_compression_tcm_final: the RAW codec's forward (models/raw2bit.py:1768-1855, likelihood path; no entropy coder)
    on 4K mosaics, packed RAW padded to a multiple of 128.  Same timing contract and JSON shape as the headline run."""
    from realcamnet_amd import ops, shard
    B, H2, W2 = args.frames, args.height, args.width
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(); shard.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize(); shard.barrier(); torch.cuda.synchronize()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, dev)
    ops.prof_enable(True)
    step()
    n_launch, conv_ms, conv_flops = ops.prof_collect()
    ops.prof_enable(False)
    if rank != 0:
        return
    total_frames = B * world
    value = total_frames * H2 * W2 / 1e6 * args.steps / elapsed
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = PEAK_TFLOPS[args.dtype]
    hp, wp = out["lsc"].shape[-2:]
    res = {"metric": f"megapixels/sec RAW mosaic {W2}x{H2} -> raw_compression_tcm_final.forward (likelihood path)", "value": round(value, 2), "unit": "MP/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
           "data": "synthetic (uniform[0,1) mosaics, seed-0 random-init weights)",
           "config": {"workload": f"cfg5 codec leg: {W2}x{H2} Bayer mosaic -> unshuffle + pad128 (packed {wp}x{hp}) -> raw_compression_tcm_final "
                                  f"(N=64, M=320, 5 slices) -> x_hat {2 * wp}x{2 * hp} + likelihoods, {B} frames/GPU, {args.dtype} storage / fp32 accumulate",
                      "frames_per_gpu": B, "global_frames": total_frames, "parallelism": f"frame-shard x{world}"},
           "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                        "traffic": None, "kernel": "conv_mfma_kernel (all instantiations)", "launches_per_step": int(n_launch),
                        "kernel_ms_per_step": round(conv_ms, 3), "flops_per_step": conv_flops,
                        "flops_note": ("necessary MACs: stride-2 convolutions (bf16: a 2x2 window over the space-to-depth map) are counted at their "
                                       "9 real (tap, phase) blocks, not the 16 executed" if args.dtype == "bf16" else
                                       "fp32 stride-2 convolutions run as a 3x3 embedding over the space-to-depth map and are counted as executed (4x)")}}
    if world == 1 and not args.no_cpu_baseline:
        import liteisp_oracle as O                      # the oracle: checker and CPU baseline only
        import raw2bit_oracle as RO
        g = torch.Generator().manual_seed(1234)
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        S = 2048                                                      # bounded sample: one 2048 x 2048 mosaic (~10 s of CPU work on 16 threads)
        with torch.no_grad():
            wm = torch.rand(1, 1, 512, 512, generator=g)              # untimed warm-up (thread pool, oneDNN primitives)
            wr, wc = O.raw_ingest(wm)
            RO.raw_compression_tcm_final(sd_cpu, [wr, wc, O.make_coord(1, 256, 256)])
            mos = torch.rand(1, 1, S, S, generator=g)
            raw, cond = O.raw_ingest(mos); coord = O.make_coord(1, S // 2, S // 2)
            t0 = time.perf_counter()
            ref = RO.raw_compression_tcm_final(sd_cpu, [raw, cond, coord])
            tc = time.perf_counter() - t0
            y = net([raw.to(dev, dt), cond.to(dev, dt), coord.to(dev, dt)])
        res["cpu_baseline"] = {"value": round(S * S / 1e6 / tc, 4), "unit": "MP/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"1 frame, packed RAW 4x{S // 2}x{S // 2} (a {S}x{S} mosaic) fp32, oracle/raw2bit_oracle.py, {tc:.1f} s"}
        res["psnr_db_vs_cpu_fp32"] = {"y (latent, before rounding)": round(O.psnr(y["para"]["y"].float().cpu(), ref["para"]["y"]), 2),
                                      "x_hat": round(O.psnr(y["x_hat"].float().cpu(), ref["x_hat"]), 2)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
