"""The drop-in boundary as PyTorch custom ops: `torch.ops.realcam.*` are registered by schema, each with a
CUDA kernel (the C-ABI launch) and a fake-tensor kernel, so the host mirror's forward() can be traced without a GPU.
CPU: schemas and FakeTensorMode traces of whole forwards.  GPU: the ops called directly give the module results."""
import pytest
import torch
from torch._subclasses.fake_tensor import FakeTensorMode

import realcamnet_amd as M
from realcamnet_amd import torch_ops


def test_every_op_is_registered_with_its_schema():
    assert len(torch_ops.SCHEMAS) >= 36
    for name, schema in torch_ops.SCHEMAS.items():
        op = getattr(torch.ops.realcam, name).default
        assert str(op._schema).replace("realcam::", "") == schema, name
    # the schemas SURVEY.md 8(b) names
    for name in ("bayer_unshuffle", "conv2d", "ca_gate", "haar_dwt", "haar_idwt", "pointwise_chain48", "color_block", "window_attention",
                 "gma_kv", "gma_apply", "raw_ingest"):
        assert name in torch_ops.SCHEMAS


def test_ops_have_no_cpu_kernel():
    x = torch.zeros(1, 8, 8, 16)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.realcam.square(x)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fake_tensor_trace_of_whole_forwards(dt):
    """FakeTensorMode on a machine without a GPU: every launch goes through a registered op with a fake kernel, so shapes and
    dtypes of a whole forward (ingest, conv stack, conditioning, GroupMix block, tail) come out without touching the library."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        with FakeTensorMode():
            with torch.device("cuda"):
                mosaic, cond, coord = torch.empty(2, 1, 172, 280), torch.empty(2, 4, 64, 64), torch.empty(2, 2, 86, 140)
                raw, coord_p = torch.empty(2, 4, 64, 96), torch.empty(2, 2, 64, 96)
                for name in ("LiteISPNet_GFM_LSC", "LiteISPNet_GFM_LSC_GMA", "ISPUNet_GFM_LSC", "LiteISPNet", "ResUNet", "ISPUNet_GFM", "ISPUNet_LSC",
                             "LiteISPNet_LSC", "LiteISPNet_GFM", "LiteISPNet_GFMresize", "ISPUNet_GFM_LFM", "ISPUNet_GFM_crop", "ISPUNet_GFM_LSC1",
                             "ISPUNet_GFM_LSC_noskip"):
                    net = getattr(M, name)().eval()
                    with torch.no_grad():
                        y = net.forward_mosaic(mosaic, cond, coord)
                        y2 = net([raw, cond, coord_p])
                    assert y.shape == (2, 3, 172, 280) and y.dtype == dt and y.device.type == "cuda", name
                    assert y2.shape == (2, 3, 128, 192), name
                blk = M.GMA_Block(80, 8).eval()
                with torch.no_grad():
                    t = blk(torch.empty(1, 24 * 40, 80), (24, 40))
                assert t.shape == (1, 960, 80)
    finally:
        torch.set_default_dtype(old)


def test_fake_tensor_trace_of_the_codecs():
    """The three codecs' forward under FakeTensorMode (no GPU, no library call): entropy models, GDN, window attention, slice loop, synthesis --
    every launch is a registered op with a fake kernel, host-side parameter packing is skipped for fake parameters."""
    import realcamnet_amd.raw2bit as RB
    import realcamnet_amd.tcm as T
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with FakeTensorMode():
            with torch.device("cuda"):
                raw = [torch.empty(2, 4, 256, 256), torch.empty(2, 4, 64, 64), torch.empty(2, 2, 256, 256)]
                for make, x, keys in ((lambda: RB.raw_compression_tcm_final(N=64), raw, {"x_hat", "likelihoods", "para", "y", "lft", "lsc"}),
                                      (lambda: RB.raw_compression_tcm(N=64), raw, {"x_hat", "likelihoods", "para"}),
                                      (lambda: T.TCM(N=64), torch.empty(2, 3, 256, 256), {"x_hat", "likelihoods", "para"})):
                    net = make().eval()
                    with torch.no_grad():
                        out = net(x)
                    assert set(out) == keys
                    assert out["x_hat"].shape == ((2, 3, 512, 512) if x is raw else (2, 3, 256, 256)) and out["x_hat"].dtype == torch.bfloat16
                    assert out["likelihoods"]["y"].shape == (2, 320, 16, 16) and out["likelihoods"]["z"].shape == (2, 192, 4, 4)
    finally:
        torch.set_default_dtype(old)


def test_fake_kernels_of_single_ops():
    with FakeTensorMode():
        with torch.device("cuda"):
            x = torch.empty(2, 16, 24, 48, dtype=torch.bfloat16)
            wp = torch.empty(1024, dtype=torch.uint8)
            out, stored, sums = torch.ops.realcam.conv2d(x, wp, None, 192, 3, 0, 0.0, None, None, None, None, None, None, False, 1, False, 0, 0,
                                                         None, None)
            assert out.shape == (2, 32, 48, 48) and stored.numel() == 0 and sums.numel() == 0     # PixelShuffle(2) folded into the store
            out, _, sums = torch.ops.realcam.conv2d(x, wp, None, 3, 3, 0, 0.0, None, None, None, None, None, None, False, 2, True, 10, 20,
                                                    torch.float32, None)
            assert out.shape == (2, 3, 10, 20) and out.dtype == torch.float32 and sums.shape[0] == 2 and sums.shape[2] == 3
            packed, cnd = torch.ops.realcam.raw_ingest(torch.empty(3, 100, 60), torch.bfloat16, 16, 64.0, 1023.0, 32, 48)
            assert packed.shape == (3, 64, 32, 4) and cnd.shape == (3, 4, 32, 48)
            assert torch.ops.realcam.haar_dwt(x, torch.empty(192, 1, 2, 2), True).shape == (2, 8, 12, 192)
            assert torch.ops.realcam.channel_concat([x, x[..., :16]]).shape == (2, 16, 24, 64)
            y = torch.ops.realcam.conv2d_fold2(torch.empty(2, 37, 71, 64, dtype=torch.bfloat16), wp, None, 96, 0, 0.0)
            assert y.shape == (2, 19, 36, 96)                    # 3x3 stride-2 convolution reading its own input: ceil(H / 2) x ceil(W / 2)


@pytest.mark.gpu
def test_ops_called_directly_equal_module_results(hip):
    """torch.ops.realcam.* are the product entry points: calling them by name reproduces what the modules compute."""
    from realcamnet_amd import networks as N, ops
    dev = "cuda"
    g = torch.Generator().manual_seed(2)
    mosaic = torch.rand(2, 1, 44, 72, generator=g).to(dev)
    a = torch.ops.realcam.bayer_unshuffle(mosaic[:, 0], torch.float32, 16)
    assert torch.equal(a, ops.bayer_unshuffle(mosaic, pad_to=16))
    c = N.conv(4, 48, mode="C").to(dev)
    wp, bp = torch.ops.realcam.conv_pack_weights(c.weight.detach(), c.bias.detach(), torch.float32, 0)
    out, stored, sums = torch.ops.realcam.conv2d(a, wp, bp, 48, 3, 1, 0.0, None, None, None, None, None, None, False, 0, True, 0, 0, None)
    want, want_sums = ops.conv2d(a, c, act="relu", want_sums=True)
    assert torch.equal(out, want) and torch.equal(sums, want_sums) and stored.numel() == 0
    with pytest.raises(Exception):
        torch.ops.realcam.square(torch.zeros(4, 4))              # CPU tensor: no backend
