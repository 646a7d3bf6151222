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
