"""Winograd F(2x2, 3x3) form of the 3x3 stride-1 convolutions (rc_conv_desc.algo = 1, csrc/wino.hip; upstream layer: networks.conv mode 'C',
models/networks.py:146-160 = nn.Conv2d).

CPU: the packed U = G g G^T buffer follows the documented fragment order and, run through the Winograd algebra in torch, reproduces F.conv2d.
GPU (-m gpu): the kernel against F.conv2d -- bit for bit on small-integer data (every U, V, product and partial sum is exactly representable), within
the fp32 block tolerance of tests/test_gpu_parity.py (2e-5 * max|ref|) on real-valued data; every epilogue form; ragged / odd / tiny images; the channel
sums against the sums of the stored map; frame i of a batch == frame i alone."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from realcamnet_amd import _lib
from realcamnet_amd import networks as N
from realcamnet_amd import ops

DEV = "cuda"
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def _unpack_u(buf: np.ndarray, cin: int, cout: int) -> torch.Tensor:
    """Packed buffer -> U[xi][cout][cin], following the layout comment of rc_wino_pack_weights:
    [cout group][stage][ks][cw][q][lane][e], xi = 4 q + e, cout = cg * 16 NCW + 16 cw + (lane & 15), cin = 8 stage + 2 (lane >> 4) + ks."""
    ncw = 4 if cout % 64 == 0 else 3 if cout % 48 == 0 else 2 if cout % 32 == 0 else 1
    n_cg, n_st = cout // (16 * ncw), cin // 8
    a = torch.from_numpy(buf.view(np.float32).copy()).reshape(n_cg, n_st, 2, ncw, 4, 64, 4)
    U = torch.empty(16, cout, cin)
    for cg in range(n_cg):
        for s in range(n_st):
            for ks in range(2):
                for cw in range(ncw):
                    for lane in range(64):
                        co, ci = cg * 16 * ncw + 16 * cw + (lane & 15), 8 * s + 2 * (lane >> 4) + ks
                        U[:, co, ci] = a[cg, s, ks, cw, :, lane, :].reshape(16)
    return U


@pytest.mark.parametrize("cin,cout", [(8, 16), (16, 48), (64, 64), (24, 128)])
def test_packed_u_follows_the_documented_order_and_reproduces_conv2d(cin, cout):
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin + cout)
    w = torch.randn(cout, cin, 3, 3, generator=g)
    n = lib.rc_wino_packed_bytes(cin, cout, _lib.RC_F32)
    assert n == 16 * cin * cout * 4
    dst = np.empty(n, dtype=np.uint8)
    wh = np.ascontiguousarray(w.numpy())
    assert lib.rc_wino_pack_weights(wh.ctypes.data, cin, cout, _lib.RC_F32, dst.ctypes.data) == 0
    U = _unpack_u(dst, cin, cout)
    ref_u = torch.einsum("ij,ocjk,lk->iloc", G, w.double(), G).reshape(16, cout, cin).float()
    assert torch.equal(U, ref_u)
    # the algebra the kernel runs, in torch: V = B^T d B per 4x4 patch, M = U V per xi, Y = A^T M A
    x = torch.randn(1, cin, 6, 10, generator=g)
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2).double()                       # B,C,Th,Tw,4,4
    V = torch.einsum("ij,bcyxjk,lk->ilbcyx", BT, d, BT).reshape(16, 1, cin, 3, 5)
    M = torch.einsum("xoc,xbcyt->xboyt", U.double(), V).reshape(4, 4, 1, cout, 3, 5)
    Y = torch.einsum("ij,jkboyt,lk->boyitl", AT, M, AT).reshape(1, cout, 6, 10)
    assert (Y - F.conv2d(x.double(), w.double(), padding=1)).abs().max() < 1e-5


def test_wino_pack_rejects_shapes_it_cannot_run():
    lib = _lib.load()
    assert lib.rc_wino_packed_bytes(4, 64, _lib.RC_F32) == 0        # cin % 8
    assert lib.rc_wino_packed_bytes(64, 3, _lib.RC_F32) == 0        # cout % 16
    assert lib.rc_wino_packed_bytes(64, 64, _lib.RC_BF16) == 0      # fp32 form only
    d = _lib.ConvDesc()
    d.batch, d.height, d.width, d.cin, d.cout, d.ksize, d.dtype, d.algo = 1, 8, 8, 64, 64, 1, _lib.RC_F32, 1
    assert lib.rc_conv_sum_slots(C.byref(d)) == -1                 # not a 3x3 layer
    d.ksize = 3
    d.out_dtype = _lib.RC_F32
    assert lib.rc_conv_sum_slots(C.byref(d)) == 2 * 1                # ceil(8 / 4) x ceil(8 / 32) regions
    d.algo = 2
    assert lib.rc_conv2d(C.byref(d), None) != 0


def _int_conv(cin, cout, seed):
    g = torch.Generator().manual_seed(seed)
    c = N.Conv2d(cin, cout, 3, 1, 1)
    with torch.no_grad():
        c.weight.copy_(torch.randint(-2, 3, c.weight.shape, generator=g).float() / 2)
        c.bias.copy_(torch.randint(-2, 3, c.bias.shape, generator=g).float())
    return c, g


def _wino(xn, c, **kw):
    """ops.conv2d with the Winograd form forced for every shape the LIBRARY runs (ops.winograd_ok's policy only routes cout % 64 == 0 there)."""
    real = ops.winograd_ok
    ops.winograd_ok = lambda x, mod, **k: mod.weight.shape[0] % 16 == 0 and mod.weight.shape[1] % 8 == 0 and k.get("act") != "gelu"
    try:
        return ops.conv2d(xn, c, **kw)
    finally:
        ops.winograd_ok = real


SHAPES = [(64, 64, 8, 32, 1), (64, 64, 16, 64, 2), (8, 16, 5, 7, 2), (16, 32, 9, 33, 1), (48, 48, 21, 70, 2), (24, 128, 3, 3, 1), (128, 64, 1, 1, 3), (64, 192, 17, 31, 1),
          (256, 64, 12, 40, 1)]


@pytest.fixture(params=[0, 1, 2], ids=["nnt_auto", "nnt1", "nnt2"])
def nnt(request, hip):
    """The kernel's two item sizes (4 x 16 / 4 x 32 pixels; rc_debug_set("wino_nnt")): forced in turn, and the automatic choice."""
    assert hip.rc_debug_set(b"wino_nnt", request.param) == 0
    yield request.param
    hip.rc_debug_set(b"wino_nnt", 0)


def _regions(h, w, nnt):
    return ((h + 3) // 4) * ((w + 16 * nnt - 1) // (16 * nnt))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_winograd_conv_exact_on_small_integer_data(hip, nnt, shape):
    """Multiples of 1/2 in, multiples of 1/8 in U: every intermediate is exact in fp32, so the Winograd kernel equals F.conv2d bit for bit --
    ragged, odd, tiny and multi-region images, every cout-tile count, several cout groups and stages."""
    cin, cout, h, w, b = shape
    c, g = _int_conv(cin, cout, cin * 1000 + cout + h)
    x = torch.randint(-2, 3, (b, cin, h, w), generator=g).float() / 2
    ref = F.conv2d(x, c.weight.detach(), c.bias.detach(), padding=1).permute(0, 2, 3, 1).contiguous()
    c = c.to(DEV)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = _wino(xn, c)
    assert torch.equal(y.cpu(), ref)
    # and equals the implicit GEMM on the same data
    ops.WINOGRAD = False
    try:
        y0 = ops.conv2d(xn, c)
    finally:
        ops.WINOGRAD = True
    assert torch.equal(y0, y)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["relu", "leaky", "residual", "relu_post", "scale_residual", "film_leaky", "relu_sums", "leaky_sums", "nobias"])
@pytest.mark.parametrize("shape", [(64, 64, 19, 45, 2), (16, 48, 8, 32, 1)])
def test_winograd_epilogues_exact_on_small_integer_data(hip, nnt, form, shape):
    cin, cout, h, w, b = shape
    c, g = _int_conv(cin, cout, 77 + cin + len(form))
    if form == "nobias":
        c.bias = None
    x = torch.randint(-2, 3, (b, cin, h, w), generator=g).float() / 2
    res = torch.randint(-4, 5, (b, cout, h, w), generator=g).float() / 2
    scale = torch.randint(0, 5, (b, cout), generator=g).float() / 4
    fs, ft = torch.randint(-2, 3, (b, cout), generator=g).float() / 2, torch.randint(-2, 3, (b, cout), generator=g).float()
    v = F.conv2d(x, c.weight.detach(), c.bias.detach() if c.bias is not None else None, padding=1)
    kw = {}
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)
    if form in ("relu", "relu_sums"):
        v, kw = torch.relu(v), dict(act="relu")
    elif form in ("leaky", "leaky_sums"):
        v, kw = F.leaky_relu(v, 0.25), dict(act="leaky", slope=0.25)
    elif form == "residual":
        v, kw = v + res, dict(residual=nhwc(res))
    elif form == "relu_post":
        v, kw = torch.relu(v + res), dict(act="relu_post", residual=nhwc(res))
    elif form == "scale_residual":
        v, kw = v * scale[:, :, None, None] + res, dict(out_scale=scale.to(DEV), residual=nhwc(res))
    elif form == "film_leaky":
        v = F.leaky_relu(v * fs[:, :, None, None] + ft[:, :, None, None] + v, 0.5)
        kw = dict(act="leaky", slope=0.5, film=(fs.to(DEV), ft.to(DEV)))
    sums = form.endswith("_sums")
    c = c.to(DEV)
    xn = nhwc(x)
    out = _wino(xn, c, want_sums=sums, **kw)
    if sums:
        out, s = out
        assert s.shape[0] == b and s.shape[2] == cout and s.shape[1] in ([_regions(h, w, nnt)] if nnt and cout % 64 == 0 else [_regions(h, w, 1), _regions(h, w, 2)])
        assert torch.equal(s.sum(dim=1).cpu(), v.sum(dim=(2, 3)))          # exact data: any summation order gives the same total
    assert torch.equal(out.cpu(), v.permute(0, 2, 3, 1).contiguous())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(64, 64, 40, 72, 2), (128, 128, 17, 30, 1), (48, 96, 33, 65, 1), (512, 128, 9, 15, 1)])
def test_winograd_conv_real_valued_within_fp32_block_tolerance(hip, nnt, shape):
    """Real-valued data: F(2,3) in fp32 differs from a direct fp32 convolution by rounding only -- the block tolerance of the fp32 path (2e-5 * max|ref|),
    against a float64 reference; frame i of a batch equals frame i alone bit for bit; run to run bit for bit."""
    cin, cout, h, w, b = shape
    g = torch.Generator().manual_seed(cin + cout + h)
    c = N.Conv2d(cin, cout, 3, 1, 1)
    x = torch.randn(b, cin, h, w, generator=g)
    ref = F.conv2d(x.double(), c.weight.detach().double(), c.bias.detach().double(), padding=1).permute(0, 2, 3, 1)
    c = c.to(DEV)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y, s = _wino(xn, c, act="relu", want_sums=True)
    y2, s2 = _wino(xn, c, act="relu", want_sums=True)
    assert torch.equal(y, y2) and torch.equal(s, s2)
    refr = torch.relu(ref)
    assert ((y.cpu().double() - refr).abs().max() / refr.abs().max()).item() <= 2e-5
    tot = s.sum(dim=1).cpu().double()
    assert ((tot - refr.sum(dim=(1, 2))).abs().max() / refr.sum(dim=(1, 2)).abs().max()).item() <= 1e-5
    if b > 1:
        y1, s1 = _wino(xn[1:2].contiguous(), c, act="relu", want_sums=True)
        assert torch.equal(y1[0], y[1]) and torch.equal(s1[0], s[1])


@pytest.mark.gpu
def test_winograd_is_what_the_fp32_nets_run(hip):
    """ops.conv2d routes eligible fp32 3x3 layers to algo 1 by default; bf16 layers and the shapes it does not cover stay on the implicit GEMM."""
    c = N.Conv2d(64, 64, 3, 1, 1).to(DEV)
    x32 = torch.zeros(1, 8, 32, 64, device=DEV)
    assert ops.winograd_ok(x32, c) and not ops.winograd_ok(x32.bfloat16(), c)
    assert not ops.winograd_ok(x32, N.Conv2d(4, 64, 3, 1, 1).to(DEV)) and not ops.winograd_ok(x32, N.Conv2d(64, 3, 3, 1, 1).to(DEV))
    assert not ops.winograd_ok(x32, N.Conv2d(64, 48, 3, 1, 1).to(DEV))            # policy: the 64-couts-per-block form only
    assert not ops.winograd_ok(x32, N.Conv2d(64, 64, 1, 1, 0).to(DEV)) and not ops.winograd_ok(x32, c, act="gelu")
