"""Generate tests/golden/tcm_block_*.npz by running the IMPORTED reference's tcm.Block (build container only).

    python oracle/make_golden_tcm.py

Fixtures are data only (inputs, the block's small state_dict, expected outputs).  CompressAI is absent and stubbed by
name (oracle/_import_reference.py); tcm.Block / WMSA do not use it.  torch.set_num_threads(1); weights from
torch.manual_seed(0) + a bias/LN perturbation so that zero-initialised parameters are exercised too."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np
import torch

import _import_reference as R
import tcm_oracle as TO

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CASES = [  # name, C, head_dim, window, type, (b, h, w)
    ("tcm_block_W_ws8_c64_hd16", 64, 16, 8, "W", (2, 16, 24)),
    ("tcm_block_SW_ws8_c64_hd16", 64, 16, 8, "SW", (1, 16, 24)),
    ("tcm_block_W_ws4_c32_hd8", 32, 8, 4, "W", (2, 8, 12)),
    ("tcm_block_SW_ws4_c64_hd32", 64, 32, 4, "SW", (1, 12, 8)),
]


def main():
    torch.set_num_threads(1)
    (T,) = R.load("tcm")
    g = torch.Generator().manual_seed(1234)
    for name, c, hd, ws, typ, shape in CASES:
        torch.manual_seed(0)
        m = T.Block(c, c, hd, ws, 0.0, type=typ).eval()
        with torch.no_grad():
            for k, v in m.state_dict().items():          # exercise biases / LayerNorm affine / a livelier position bias
                if k.endswith("bias") or "ln" in k:
                    v.add_(torch.randn(v.shape, generator=g) * 0.1)
                if k.endswith("relative_position_params"):
                    v.add_(torch.randn(v.shape, generator=g) * 0.5)
            x = torch.randn(*shape, c, generator=g)
            y = m(x)
            sd = m.state_dict()
            yo = TO.block(sd, "", x, hd, ws, typ)
        assert (y - yo).abs().max() <= 1e-5 * y.abs().max(), name
        arrays = {"x": x.numpy(), "y": y.numpy(), "head_dim": np.array(hd), "window": np.array(ws), "type": np.array(typ),
                  "torch_version": np.array(torch.__version__), "reference": np.array("kepengxu/RealCamNet@2024-10-20")}
        arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, oracle max |diff| {(y - yo).abs().max().item():.2e}")


def swin():
    torch.set_num_threads(1)
    (T,) = R.load("tcm")
    g = torch.Generator().manual_seed(4321)
    torch.manual_seed(0)
    c, hd, ws = 64, 16, 8
    m = T.SwinBlock(c, c, hd, ws, 0.0).eval()
    with torch.no_grad():
        for k, v in m.state_dict().items():
            if k.endswith("bias") or "ln" in k:
                v.add_(torch.randn(v.shape, generator=g) * 0.1)
            if k.endswith("relative_position_params"):
                v.add_(torch.randn(v.shape, generator=g) * 0.5)
        x = torch.randn(1, c, 16, 24, generator=g)
        y = m(x)
        sd = m.state_dict()
        yo = TO.swin_block(sd, "", x, hd, ws)
    assert (y - yo).abs().max() <= 1e-5 * y.abs().max()
    arrays = {"x": x.numpy(), "y": y.numpy(), "head_dim": np.array(hd), "window": np.array(ws),
              "torch_version": np.array(torch.__version__), "reference": np.array("kepengxu/RealCamNet@2024-10-20")}
    arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
    path = os.path.join(OUT, "tcm_swinblock_ws8_c64_hd16.npz")
    np.savez_compressed(path, **arrays)
    print(f"tcm_swinblock_ws8_c64_hd16: {os.path.getsize(path) / 1024:.1f} KiB, oracle max |diff| {(y - yo).abs().max().item():.2e}")


class _RestatedResidualBlock(torch.nn.Module):
    """Stand-in for compressai.layers.ResidualBlock (absent here), written from its published definition; only used to let
    the reference's ConvTransBlock run.  The fixture it produces pins ConvTransBlock's OWN logic (split, double residual,
    concat order, 1x1 convs, Block); this layer itself stays unpinned."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv1 = torch.nn.Conv2d(in_ch, out_ch, 3, padding=1)
        self.leaky_relu = torch.nn.LeakyReLU(inplace=True)
        self.conv2 = torch.nn.Conv2d(out_ch, out_ch, 3, padding=1)
        self.skip = torch.nn.Conv2d(in_ch, out_ch, 1) if in_ch != out_ch else None

    def forward(self, x):
        out = self.leaky_relu(self.conv2(self.leaky_relu(self.conv1(x))))
        return out + (x if self.skip is None else self.skip(x))


def conv_trans():
    torch.set_num_threads(1)
    (T,) = R.load("tcm")
    T.ResidualBlock = _RestatedResidualBlock
    g = torch.Generator().manual_seed(777)
    for name, typ in (("tcm_convtrans_W_n64_hd16_ws8", "W"), ("tcm_convtrans_SW_n64_hd16_ws8", "SW")):
        torch.manual_seed(0)
        m = T.ConvTransBlock(64, 64, 16, 8, 0.0, type=typ).eval()
        with torch.no_grad():
            for k, v in m.state_dict().items():
                if k.endswith("bias") or "ln" in k:
                    v.add_(torch.randn(v.shape, generator=g) * 0.1)
                if k.endswith("relative_position_params"):
                    v.add_(torch.randn(v.shape, generator=g) * 0.5)
            x = torch.randn(1, 128, 16, 24, generator=g)
            y = m(x)
            sd = m.state_dict()
            yo = TO.conv_trans_block(sd, "", x, 64, 64, 16, 8, typ)
        assert (y - yo).abs().max() <= 1e-5 * y.abs().max(), name
        arrays = {"x": x.numpy(), "y": y.numpy(), "head_dim": np.array(16), "window": np.array(8), "type": np.array(typ),
                  "conv_dim": np.array(64), "trans_dim": np.array(64),
                  "torch_version": np.array(torch.__version__), "reference": np.array("kepengxu/RealCamNet@2024-10-20 (+ restated compressai ResidualBlock)")}
        arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, oracle max |diff| {(y - yo).abs().max().item():.2e}")


class _RestatedAttentionBlock(torch.nn.Module):
    """Stand-in for compressai.layers.AttentionBlock (absent here), written from its published definition; installed as the
    base class of the reference's SWAtten so that class can be constructed and run.  The fixture pins SWAtten's OWN logic
    (in_conv / out_conv, SwinBlock placement, which branch sees z, the sigmoid gate and identity); conv_a / conv_b stay
    unpinned."""

    def __init__(self, N):
        super().__init__()
        nn = torch.nn

        class ResidualUnit(nn.Module):
            def __init__(self):
                super().__init__()
                self.conv = nn.Sequential(nn.Conv2d(N, N // 2, 1), nn.ReLU(inplace=True), nn.Conv2d(N // 2, N // 2, 3, padding=1),
                                          nn.ReLU(inplace=True), nn.Conv2d(N // 2, N, 1))
                self.relu = nn.ReLU(inplace=True)

            def forward(self, x):
                identity = x
                out = self.conv(x)
                out += identity
                return self.relu(out)

        self.conv_a = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit())
        self.conv_b = nn.Sequential(ResidualUnit(), ResidualUnit(), ResidualUnit(), nn.Conv2d(N, N, 1))

    def forward(self, x):
        return self.conv_a(x) * torch.sigmoid(self.conv_b(x)) + x


class _RestatedGDN(torch.nn.Module):
    """compressai.layers.GDN, restated (see _RestatedAttentionBlock for why)."""

    class _NonNeg(torch.nn.Module):
        def __init__(self, minimum=0.0, reparam_offset=2 ** -18):
            super().__init__()
            pedestal = float(reparam_offset) ** 2
            self.register_buffer("pedestal", torch.Tensor([pedestal]))
            self.lower_bound = torch.nn.Module()
            self.lower_bound.register_buffer("bound", torch.Tensor([(float(minimum) + pedestal) ** 0.5]))

        def init(self, x):
            return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

        def forward(self, x):
            return torch.max(x, self.lower_bound.bound) ** 2 - self.pedestal

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_reparam = self._NonNeg(minimum=float(beta_min))
        self.beta = torch.nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = self._NonNeg()
        self.gamma = torch.nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))

    def forward(self, x):
        c = x.shape[1]
        norm = torch.nn.functional.conv2d(x ** 2, self.gamma_reparam(self.gamma).reshape(c, c, 1, 1), self.beta_reparam(self.beta))
        return x * (torch.sqrt(norm) if self.inverse else torch.rsqrt(norm))


def _restated_conv3x3(in_ch, out_ch, stride=1):
    return torch.nn.Conv2d(in_ch, out_ch, kernel_size=3, stride=stride, padding=1)


def _restated_subpel_conv3x3(in_ch, out_ch, r=1):
    return torch.nn.Sequential(torch.nn.Conv2d(in_ch, out_ch * r ** 2, kernel_size=3, padding=1), torch.nn.PixelShuffle(r))


class _RestatedResidualBlockWithStride(torch.nn.Module):
    def __init__(self, in_ch, out_ch, stride=2):
        super().__init__()
        self.conv1 = _restated_conv3x3(in_ch, out_ch, stride=stride)
        self.leaky_relu = torch.nn.LeakyReLU(inplace=True)
        self.conv2 = _restated_conv3x3(out_ch, out_ch)
        self.gdn = _RestatedGDN(out_ch)
        self.skip = torch.nn.Conv2d(in_ch, out_ch, kernel_size=1, stride=stride) if (stride != 1 or in_ch != out_ch) else None

    def forward(self, x):
        out = self.gdn(self.conv2(self.leaky_relu(self.conv1(x))))
        return out + (x if self.skip is None else self.skip(x))


class _RestatedResidualBlockUpsample(torch.nn.Module):
    def __init__(self, in_ch, out_ch, upsample=2):
        super().__init__()
        self.subpel_conv = _restated_subpel_conv3x3(in_ch, out_ch, upsample)
        self.leaky_relu = torch.nn.LeakyReLU(inplace=True)
        self.conv = _restated_conv3x3(out_ch, out_ch)
        self.igdn = _RestatedGDN(out_ch, inverse=True)
        self.upsample = _restated_subpel_conv3x3(in_ch, out_ch, upsample)

    def forward(self, x):
        out = self.igdn(self.conv(self.leaky_relu(self.subpel_conv(x))))
        return out + self.upsample(x)


class _Dummy(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


def transforms():
    """g_a / g_s / h_a / h_mean_s of the reference's TCM, built by ITS OWN __init__ (models/tcm.py:321-385: block order, head
    dims, window sizes, W/SW alternation, channel counts) over restated CompressAI layers; CompressionModel and the entropy
    models are inert placeholders (not exercised).  Pins the composition; the CompressAI layers themselves stay unpinned."""
    import importlib
    torch.set_num_threads(1)
    R.install_stubs()
    L = sys.modules["compressai.layers"]
    L.AttentionBlock, L.ResidualBlock = _RestatedAttentionBlock, _RestatedResidualBlock
    L.ResidualBlockWithStride, L.ResidualBlockUpsample = _RestatedResidualBlockWithStride, _RestatedResidualBlockUpsample
    L.conv3x3, L.subpel_conv3x3 = _restated_conv3x3, _restated_subpel_conv3x3
    sys.modules["compressai.models"].CompressionModel = _Dummy
    sys.modules["compressai.entropy_models"].EntropyBottleneck = _Dummy
    sys.modules["compressai.entropy_models"].GaussianConditional = _Dummy
    sys.modules.pop("models.tcm", None)
    T = importlib.import_module("models.tcm")
    g = torch.Generator().manual_seed(97531)
    torch.manual_seed(0)
    n, mm = 32, 64                                        # TCM(N=32, M=64): the default layout at half the width (head_dim 32 needs N >= 32)
    model = T.TCM(config=[2, 2, 2, 2, 2, 2], head_dim=[8, 16, 32, 32, 16, 8], N=n, M=mm, num_slices=2).eval()
    specs = TO.tcm_transform_specs(N=n)
    with torch.no_grad():
        _perturb(model, g)
        for k, v in model.state_dict().items():            # livelier GDN parameters than the identity-like init
            if k.endswith(".gamma") or k.endswith(".beta"):
                v.add_(torch.rand(v.shape, generator=g) * 0.05)
        for k, v in model.state_dict().items():            # weights made bf16-exact so the fixture can store 16 bits each
            if v.is_floating_point() and v.numel() > 64:
                v.copy_(v.bfloat16().float())
        x = torch.rand(1, 3, 128, 128, generator=g)
        y = model.g_a(x)
        xh = model.g_s(y)
        sd = model.state_dict()
        pick = lambda pre: {k[len(pre) + 1:]: v for k, v in sd.items() if k.startswith(pre + ".")}
        arrays = {"x": x.numpy(), "y": y.numpy(), "x_hat": xh.numpy(), "N": np.array(n), "M": np.array(mm),
                  "torch_version": np.array(torch.__version__),
                  "reference": np.array("kepengxu/RealCamNet@2024-10-20 TCM.__init__ (+ restated compressai layers)")}
        for name, inp in (("g_a", x), ("g_s", y)):
            yo = TO.run_transform(pick(name), "", specs[name], inp)
            want = y if name == "g_a" else xh
            assert (want - yo).abs().max() <= 1e-4 * want.abs().max(), (name, (want - yo).abs().max())
            for k, v in pick(name).items():
                if v.is_floating_point() and v.numel() > 64:
                    arrays[f"sd16.{name}.{k}"] = (v.contiguous().view(torch.int32) >> 16).to(torch.int16).numpy().view(np.uint16)
                else:
                    arrays[f"sd.{name}.{k}"] = v.numpy()
    path = os.path.join(OUT, "tcm_transforms_n32_m64.npz")
    np.savez_compressed(path, **arrays)
    print(f"tcm_transforms: {os.path.getsize(path) / 1024:.1f} KiB; y {tuple(y.shape)}, x_hat {tuple(xh.shape)}")


class _RestatedEntropyBottleneck(torch.nn.Module):
    """compressai.entropy_models.EntropyBottleneck, eval-mode likelihood path, restated (see _RestatedAttentionBlock)."""

    def __init__(self, channels, tail_mass=1e-9, init_scale=10, filters=(3, 3, 3, 3)):
        super().__init__()
        import math
        self.channels, self.filters = int(channels), tuple(filters)
        f = (1,) + self.filters + (1,)
        scale = float(init_scale) ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = math.log(math.expm1(1 / scale / f[i + 1]))
            self.register_parameter(f"_matrix{i}", torch.nn.Parameter(torch.full((channels, f[i + 1], f[i]), init)))
            self.register_parameter(f"_bias{i}", torch.nn.Parameter(torch.empty(channels, f[i + 1], 1).uniform_(-0.5, 0.5)))
            if i < len(self.filters):
                self.register_parameter(f"_factor{i}", torch.nn.Parameter(torch.zeros(channels, f[i + 1], 1)))
        self.quantiles = torch.nn.Parameter(torch.tensor([-float(init_scale), 0.0, float(init_scale)]).repeat(channels, 1, 1))
        target = math.log(2 / float(tail_mass) - 1)
        self.register_buffer("target", torch.tensor([-target, 0.0, target]))

    def _get_medians(self):
        return self.quantiles[:, :, 1:2]

    def _logits_cumulative(self, inputs):
        logits = inputs
        for i in range(len(self.filters) + 1):
            logits = torch.matmul(torch.nn.functional.softplus(getattr(self, f"_matrix{i}")), logits)
            logits = logits + getattr(self, f"_bias{i}")
            if i < len(self.filters):
                logits = logits + torch.tanh(getattr(self, f"_factor{i}")) * torch.tanh(logits)
        return logits

    def forward(self, x):
        perm = (1, 0, 2, 3)
        values = x.permute(*perm).contiguous()
        shape = values.size()
        values = values.reshape(x.size(1), 1, -1)
        med = self._get_medians()
        outputs = torch.round(values - med) + med
        lower, upper = self._logits_cumulative(outputs - 0.5), self._logits_cumulative(outputs + 0.5)
        sign = -torch.sign(lower + upper)
        likelihood = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower)).clamp_min(1e-9)
        return outputs.reshape(shape).permute(*perm).contiguous(), likelihood.reshape(shape).permute(*perm).contiguous()


class _RestatedGaussianConditional(torch.nn.Module):
    def __init__(self, scale_table=None, scale_bound=0.11):
        super().__init__()
        self.scale_bound = float(scale_bound)

    def forward(self, inputs, scales, means=None):
        outputs = torch.round(inputs - means) + means
        values = torch.abs(outputs - means)
        s = scales.clamp_min(self.scale_bound)
        phi = lambda t: 0.5 * torch.erfc(-(2 ** -0.5) * t)
        return outputs, (phi((0.5 - values) / s) - phi((-0.5 - values) / s)).clamp_min(1e-9)


def forward_fixture():
    """TCM.forward of the reference (models/tcm.py:437-486) at N=32 (M is 320 in the slice loop regardless), eval mode, over
    restated CompressAI classes; parameters come from oracle/det_fill.py (a function of key and shape), so the fixture stores
    only the input and the outputs.  Pins the forward's composition: slice order, supports, which tensors feed which module."""
    import importlib
    from det_fill import det_fill_
    torch.set_num_threads(8)
    R.install_stubs()
    L = sys.modules["compressai.layers"]
    L.AttentionBlock, L.ResidualBlock = _RestatedAttentionBlock, _RestatedResidualBlock
    L.ResidualBlockWithStride, L.ResidualBlockUpsample = _RestatedResidualBlockWithStride, _RestatedResidualBlockUpsample
    L.conv3x3, L.subpel_conv3x3 = _restated_conv3x3, _restated_subpel_conv3x3
    sys.modules["compressai.models"].CompressionModel = _Dummy
    sys.modules["compressai.entropy_models"].EntropyBottleneck = _RestatedEntropyBottleneck
    sys.modules["compressai.entropy_models"].GaussianConditional = _RestatedGaussianConditional
    sys.modules.pop("models.tcm", None)
    T = importlib.import_module("models.tcm")
    torch.manual_seed(0)
    n, slices = 32, 5
    model = T.TCM(N=n, M=320, num_slices=slices).eval()
    sd = model.state_dict()
    det_fill_(sd)
    g = torch.Generator().manual_seed(8642)
    x = torch.rand(1, 3, 256, 256, generator=g)           # latent 16x16: SWAtten's SwinBlock needs a map larger than its 8x8 window
    with torch.no_grad():
        out = model(x)
        ours = TO.tcm_forward(sd, x, N=n, num_slices=slices)
    flat = lambda o: {"x_hat": o["x_hat"], "lik_y": o["likelihoods"]["y"], "lik_z": o["likelihoods"]["z"], "means": o["para"]["means"],
                      "scales": o["para"]["scales"], "y": o["para"]["y"]}
    a, b = flat(out), flat(ours)
    for k in a:
        assert (a[k] - b[k]).abs().max() <= 1e-4 * max(a[k].abs().max().item(), 1e-6), (k, (a[k] - b[k]).abs().max())
    arrays = {"x": x.numpy(), "N": np.array(n), "num_slices": np.array(slices), "n_keys": np.array(len(sd)),
              "torch_version": np.array(torch.__version__),
              "reference": np.array("kepengxu/RealCamNet@2024-10-20 TCM.forward (+ restated compressai classes; det_fill parameters)")}
    arrays.update({"out." + k: v.numpy() for k, v in a.items()})
    path = os.path.join(OUT, "tcm_forward_n32.npz")
    np.savez_compressed(path, **arrays)
    print(f"tcm_forward: {os.path.getsize(path) / 1024:.1f} KiB;", {k: (tuple(v.shape), float(v.abs().mean())) for k, v in a.items()})


def _perturb(m, g):
    for k, v in m.state_dict().items():
        if k.endswith("bias") or "ln" in k:
            v.add_(torch.randn(v.shape, generator=g) * 0.1)
        if k.endswith("relative_position_params"):
            v.add_(torch.randn(v.shape, generator=g) * 0.5)


def swatten():
    import importlib
    torch.set_num_threads(1)
    R.install_stubs()
    sys.modules["compressai.layers"].AttentionBlock = _RestatedAttentionBlock     # before models.tcm creates `class SWAtten(AttentionBlock)`
    sys.modules.pop("models.tcm", None)
    T = importlib.import_module("models.tcm")
    g = torch.Generator().manual_seed(2468)
    torch.manual_seed(0)
    cin, cout, inter, hd, ws = 96, 96, 64, 16, 4
    m = T.SWAtten(cin, cout, hd, ws, 0, inter_dim=inter).eval()
    with torch.no_grad():
        _perturb(m, g)
        x = torch.randn(1, cin, 8, 12, generator=g)
        y = m(x)
        sd = m.state_dict()
        yo = TO.swatten(sd, "", x, hd, ws)
    assert (y - yo).abs().max() <= 1e-5 * y.abs().max()
    arrays = {"x": x.numpy(), "y": y.numpy(), "head_dim": np.array(hd), "window": np.array(ws), "inter_dim": np.array(inter),
              "torch_version": np.array(torch.__version__),
              "reference": np.array("kepengxu/RealCamNet@2024-10-20 (+ restated compressai AttentionBlock as base class)")}
    arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
    path = os.path.join(OUT, "tcm_swatten_c96_i64_hd16_ws4.npz")
    np.savez_compressed(path, **arrays)
    print(f"tcm_swatten: {os.path.getsize(path) / 1024:.1f} KiB, oracle max |diff| {(y - yo).abs().max().item():.2e}")


def slice_transforms():
    """cc_mean_transforms[1]-shaped stack, built exactly as TCM.__init__ builds it (models/tcm.py:398-405) from the reference's
    own `conv` helper and nn.GELU; TCM itself cannot be constructed here (CompressionModel is a CompressAI class).  Channel
    counts are scaled down (40 -> 28 -> 16 -> 8 instead of 384 -> 224 -> 128 -> 64) to keep the fixture small."""
    torch.set_num_threads(1)
    (T,) = R.load("tcm")
    nn = torch.nn
    g = torch.Generator().manual_seed(1357)
    torch.manual_seed(0)
    m = nn.Sequential(T.conv(40, 28, stride=1, kernel_size=3), nn.GELU(), T.conv(28, 16, stride=1, kernel_size=3), nn.GELU(),
                      T.conv(16, 8, stride=1, kernel_size=3)).eval()
    with torch.no_grad():
        x = torch.randn(2, 40, 9, 13, generator=g)
        y = m(x)
        sd = m.state_dict()
        yo = TO.slice_transform(sd, "", x)
    assert torch.equal(y, yo)
    arrays = {"x": x.numpy(), "y": y.numpy(), "torch_version": np.array(torch.__version__), "reference": np.array("kepengxu/RealCamNet@2024-10-20")}
    arrays.update({"sd." + k: v.numpy() for k, v in sd.items()})
    path = os.path.join(OUT, "tcm_slice_transform_40_28_16_8.npz")
    np.savez_compressed(path, **arrays)
    print(f"tcm_slice_transform: {os.path.getsize(path) / 1024:.1f} KiB, bitwise equal to the oracle")


if __name__ == "__main__":
    if "--swatten" in sys.argv:
        swatten(); sys.exit(0)
    if "--slice" in sys.argv:
        slice_transforms(); sys.exit(0)
    if "--transforms" in sys.argv:
        transforms(); sys.exit(0)
    if "--forward" in sys.argv:
        forward_fixture(); sys.exit(0)
    if "--swin" in sys.argv:
        swin(); sys.exit(0)
    if "--convtrans" in sys.argv:
        conv_trans(); sys.exit(0)
    main()
