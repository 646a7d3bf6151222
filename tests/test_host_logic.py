"""CPU: host-side mirror of the reference interface -- checkpoint contract, error behaviour, sharding."""
import os
import subprocess
import sys

import pytest
import torch

import realcamnet_amd as M
from realcamnet_amd import networks as N
from realcamnet_amd import shard
from conftest import ROOT, golden_names, load_golden, sd_digest, seed0_state_dict


def test_seed0_parameters_equal_the_reference():
    from conftest import NET_NAMES
    for name in NET_NAMES:                                  # the whole ISP family: same construction order => same seed-0 parameters
        g = load_golden(f"e2e_{name}_32x32")
        sd = seed0_state_dict(name)
        assert sd_digest(sd) == g["sd_digest"], name
        if "n_tensors" in g:
            assert len(sd) == int(g["n_tensors"]), name


def test_ispunet_checkpoint_keys():
    """Row a13: strided U-Net sibling -- down-samplers are (2c, c, 2, 2) convs, up-samplers bias-free 1x1 convs."""
    sd = seed0_state_dict("ISPUNet_GFM_LSC")
    for k, shape in {"intro.weight": (32, 4, 3, 3), "down1.weight": (64, 32, 2, 2), "down3.bias": (256,),
                     "up3.0.weight": (512, 256, 1, 1), "encoder_modulation2.1.GFM_scale_conv0.weight": (128, 32),
                     "middle.1.rg.4.weight": (256, 256, 3, 3), "decoder1.1.weight": (32, 32, 3, 3), "tail.2.weight": (3, 32, 3, 3)}.items():
        assert tuple(sd[k].shape) == shape, k
    assert "up3.0.bias" not in sd


def test_checkpoint_keys_and_counts():
    sd = seed0_state_dict("LiteISPNet")
    assert len(sd) == 302
    assert sd["head.weight"].shape == (64, 4, 3, 3)
    assert sd["down1.1.rg.0.ca.conv_du.0.weight"].shape == (4, 64, 1, 1)
    assert sd["down1.3.weight"].shape == (256, 1, 2, 2)
    sd = seed0_state_dict("LiteISPNet_GFM_LSC")
    assert len(sd) == 378 and sum(v.numel() for v in sd.values()) == 14367267
    for k, shape in {"encoder_modulation1.GFM_scale_conv0.weight": (48, 32), "classifier.model.3.weight": (16,),
                     "lsc.model.6.weight": (48, 48, 1, 1), "tail.2.weight": (3, 48, 3, 3)}.items():
        assert tuple(sd[k].shape) == shape
    net = M.LiteISPNet_GFM_LSC()
    net.load_state_dict(sd, strict=True)


def test_block_state_dicts_load_strict():
    for fixture, mod in (("block_rcag_32_nb4", N.RCAGroup(32, 32, nb=4)), ("block_rcab_32", N.RCABlock(32, 32)),
                         ("block_dwt_forward", N.DWTForward(16)), ("block_dwt_inverse", N.DWTInverse(64)),
                         ("block_res_gfm_48", M.LiteISP.Res_GFM(48, 48, 32, 48, 48)),
                         ("block_lsc_48", M.LiteISP.Lens_Shading_Correction(2, 48, 48)),
                         ("block_color_condition", M.LiteISP.Color_Condition_GFM(4, 32))):
        mod.load_state_dict(load_golden(fixture)["sd"], strict=True)


def test_no_cpu_fallback():
    net = M.LiteISPNet().eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net([torch.zeros(1, 4, 16, 16)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        N.conv(16, 16, mode="C")(torch.zeros(1, 16, 8, 8))


def test_error_behaviour_matches_contract():
    with pytest.raises(NotImplementedError):
        N.conv(16, 16, mode="CBR")           # BatchNorm is not on the hot path
    with pytest.raises(NotImplementedError):
        N.conv(16, 16, mode="Z")
    with pytest.raises(AssertionError):
        N.RCABlock(16, 32)
    assert isinstance(N.seq(N.conv(4, 8, mode="C")), N.Conv2d)        # single module collapses (key 'head.weight')
    s = N.conv(8, 8, mode="CRC")
    assert list(s.state_dict().keys()) == ["0.weight", "0.bias", "2.weight", "2.bias"]
    net = M.LiteISPNet_GFM_LSC()                                     # training mode is refused
    with pytest.raises(RuntimeError, match="eval"):
        net.classifier._vec(torch.zeros(1, 4, 8, 8))


def test_frame_shard_partition():
    for n in (0, 1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            parts = [shard.frame_shard(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [e - s for s, e in parts]
            assert max(sizes) - min(sizes) <= 1
    assert shard.frame_shard(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        shard.frame_shard(8, 8, 8)


_WORKER = r'''
import os, sys, torch
sys.path.insert(0, {root!r})
from realcamnet_amd import shard
rank, world, _ = shard.init_distributed("gloo")
n = 5                                   # uneven on purpose: 3 + 2 frames
s, e = shard.frame_shard(n, rank, world)
frames = torch.arange(n, dtype=torch.float32).view(n, 1, 1, 1).expand(n, 3, 4, 6).contiguous()
local = frames[s:e] * 2.0               # the "forward": independent per frame
full = shard.gather_frames(local, n)
assert full.shape == (n, 3, 4, 6) and torch.equal(full, frames * 2.0), (rank, full[:, 0, 0, 0])
# the bench's overlapped per-step gather (equal shards, one all_gather_into_tensor per step, double-buffered)
og = shard.OverlappedGather(4)
for step in range(3):
    mine = torch.full((2, 3, 4, 6), float(10 * step + rank))
    og.submit(mine)
got = og.wait()
want = torch.cat([torch.full((2, 3, 4, 6), float(20 + r)) for r in range(world)])
assert got.shape == (4, 3, 4, 6) and torch.equal(got, want), (rank, got[:, 0, 0, 0])
t = shard.max_over_ranks(1.0 + rank)
assert t == float(world), t
shard.barrier()
print("ok", rank)
'''


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_bench_launches_itself_for_more_than_one_gpu():
    """`python bench.py --gpus 2` with no launcher in the environment (the driver's N = 1 command form at N > 1) re-executes itself under
    torch.distributed.run with one rank per GPU; --launcher-selftest swaps the GPU step for a trivial CPU step over gloo, so the whole
    launch / rendezvous / shard / overlapped gather / barrier / max-over-ranks skeleton runs here.  The torchrun form keeps working."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "3", "--frames", "2", "--launcher-selftest"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["gathered_ok"] is True and line["launched_by"].startswith("self") and "not a measurement" in line["metric"]
    # one GPU: no launcher, no process group
    r1 = subprocess.run([sys.executable, bench, "--gpus", "1", "--steps", "2", "--launcher-selftest"], capture_output=True, text=True, timeout=120, env=env)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    assert json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])["launched_by"] == "single process"
    # the driver's multi-GPU form: already under torch.distributed.run -> no second launch; a rank count that contradicts --gpus is an error
    cmd = shard.torchrun_command(bench, ["--gpus", "2", "--steps", "2", "--frames", "2", "--launcher-selftest"], 2)
    r2 = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r2.returncode == 0 and json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 2, r2.stdout + r2.stderr
    bad = subprocess.run(shard.torchrun_command(bench, ["--gpus", "3", "--launcher-selftest"], 2), capture_output=True, text=True, timeout=240, env=env)
    assert bad.returncode != 0 and "--nproc-per-node must equal --gpus" in (bad.stdout + bad.stderr)


def test_early_gate_is_taken_only_where_its_closed_form_holds():
    """ca_gate_ahead's closed form (the mean of conv2's output from conv1's channel sums and border lines) holds for a 3x3, C -> C, stride 1,
    zero padding 1, dilation 1, groups 1 convolution and a squeeze -> ReLU -> excite -> Sigmoid attention only; any other checkpoint-compatible
    configuration must take the schedule that reduces conv2's real output instead of silently computing a wrong gate."""
    import torch.nn as nn
    from realcamnet_amd import networks as N, ops, raw2bit as RB
    blk = N.RCABlock(32, 32, 3, 1, 1, True, "CRC", 16)
    c1, c2 = blk.res[0], blk.res[2]
    assert ops.gate_ahead_ok(c1, c2, blk.ca)
    rb = RB.ResidualBlockWithCA(32, 32)
    assert ops.gate_ahead_ok(rb.conv1, rb.conv2, rb.ca)
    for bad in (nn.Conv2d(32, 32, 3, 2, 1), nn.Conv2d(32, 32, 3, 1, 2, dilation=2), nn.Conv2d(32, 32, 3, 1, 1, groups=2), nn.Conv2d(32, 32, 3, 1, 1, padding_mode="reflect"),
                nn.Conv2d(32, 32, 3, 1, 0), nn.Conv2d(32, 32, 5, 1, 2), nn.Conv2d(32, 48, 3, 1, 1), nn.Conv2d(16, 32, 3, 1, 1)):
        assert not ops.gate_ahead_ok(c1, bad, blk.ca), bad
    ca = N.CALayer(32, 16)
    ca.conv_du[1] = nn.LeakyReLU(0.1)
    assert not ops.gate_ahead_ok(c1, c2, ca)
    ca = N.CALayer(32, 16)
    ca.conv_du[3] = nn.Tanh()
    assert not ops.gate_ahead_ok(c1, c2, ca)
    assert not ops.gate_ahead_ok(nn.Conv2d(32, 16, 3, 1, 1), c2, blk.ca)           # conv1 must produce conv2's C channels
    blk.ca.conv_du[1] = nn.LeakyReLU(0.1)
    assert not blk._early(None)                                                     # the block itself falls back to the staged schedule


def test_debug_knobs_are_mirrored_host_side():
    """packed_conv asked the library for the `conv32` knob on every conv of every forward; the binding now mirrors knobs and drops the
    mirrored value whenever one is set through it."""
    from realcamnet_amd import _lib
    L = _lib.load()
    assert L.rc_debug_set(b"conv32", 0) == 0 and _lib.knob(b"conv32") == 0 and _lib._knobs.get(b"conv32") == 0
    assert L.rc_debug_set(b"conv32", 4) == 0 and b"conv32" not in _lib._knobs and _lib.knob(b"conv32") == 4
    assert L.rc_debug_set(b"conv32", 0) == 0 and _lib.knob(b"conv32") == 0
    assert L.rc_debug_get(b"persist_auto") == 1                                     # default: kernel 6 for the plain / +sums forms


def test_small_map_cout_tile_policy():
    """ops.small_map_cout_tile: 16-wide cout tiles only for fp32 3x3 layers with 64 | cout and 16 | cin whose automatic 64-wide tiling gives the general kernel fewer
    than two blocks per CU (256 CUs assumed without a GPU) -- cfg2's 128-channel levels at 1080p, B = 1 -- and never for bf16, pixel-shuffle stores, 1x1 or odd widths."""
    import torch.nn as nn
    from realcamnet_amd import ops
    from realcamnet_amd._lib import RC_OUT_NHWC, RC_OUT_NCHW, RC_OUT_PIXEL_SHUFFLE2
    conv = nn.Conv2d(128, 128, 3, 1, 1)
    x = lambda b, h, w, c=128, dt=torch.float32: torch.empty(b, h, w, c, dtype=dt)
    assert ops.small_map_cout_tile(x(1, 135, 240), conv, RC_OUT_NHWC) == 16            # 17 x 8 tiles x 2 cout tiles = 272 blocks < 512
    assert ops.small_map_cout_tile(x(1, 68, 120), conv, RC_OUT_NCHW) == 16
    assert ops.small_map_cout_tile(x(1, 270, 480), conv, RC_OUT_NHWC) == 0             # 34 x 15 x 2 = 1 020 blocks: enough
    assert ops.small_map_cout_tile(x(8, 135, 240), conv, RC_OUT_NHWC) == 0             # a batch fills the chip
    assert ops.small_map_cout_tile(x(1, 135, 240, dt=torch.bfloat16), conv.to(torch.bfloat16), RC_OUT_NHWC) == 0
    assert ops.small_map_cout_tile(x(1, 135, 240), conv, RC_OUT_PIXEL_SHUFFLE2) == 0
    assert ops.small_map_cout_tile(x(1, 135, 240), nn.Conv2d(128, 128, 1), RC_OUT_NHWC) == 0
    assert ops.small_map_cout_tile(x(1, 135, 240), nn.Conv2d(128, 96, 3, 1, 1), RC_OUT_NHWC) == 0
    assert ops.small_map_cout_tile(x(1, 135, 240, c=4), nn.Conv2d(4, 64, 3, 1, 1), RC_OUT_NHWC) == 0
    old = ops.SMALL_MAP_COUT_TILE
    try:
        ops.SMALL_MAP_COUT_TILE = False
        assert ops.small_map_cout_tile(x(1, 135, 240), conv, RC_OUT_NHWC) == 0
    finally:
        ops.SMALL_MAP_COUT_TILE = old
