"""what does Winograd F(2x2,3x3) with bf16 transformed operands cost in PSNR?  (CPU only, test infrastructure.)

Emulates the bf16 path of the build on the CPU oracle (weights and every conv's input / output rounded to bf16, fp32 accumulation) and
replaces chosen 3x3 layers by F(2,3):  U = G g G^T (from the fp32 master weights, rounded to bf16), V = B^T d B (rounded to bf16),
M = sum_c U V in fp32, Y = A^T M A in fp32.  Prints the PSNR of each variant against the fp32 oracle on the same frame.

    python tools/winograd_psnr.py [--size 1080x1920] [--which multi|all|none]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import liteisp_oracle as O  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def winograd_conv(x, w, b, round_v=True, round_u=True):
    """3x3 stride-1 zero-padded conv as F(2x2,3x3); x (B,C,H,W) fp32 (already bf16-valued), w (O,C,3,3) fp32 masters."""
    B, C, H, W = x.shape
    Hp, Wp = H + (H & 1), W + (W & 1)
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                         # B,C,Th,Tw,4,4
    V = torch.einsum("ij,bcyxjk,lk->bcyxil", BT, d, BT)
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                  # O,C,4,4
    if round_v:
        V = bf(V)
    if round_u:
        U = bf(U)
    Th, Tw = V.shape[2], V.shape[3]
    Vm = V.permute(4, 5, 1, 0, 2, 3).reshape(16, C, B * Th * Tw)
    Um = U.permute(2, 3, 0, 1).reshape(16, w.shape[0], C)
    M = torch.bmm(Um, Vm).reshape(4, 4, w.shape[0], B, Th, Tw)
    Y = torch.einsum("ij,jkobyx,lk->boyixl", AT, M, AT)            # B,O,Th,2,Tw,2
    Y = Y.reshape(B, w.shape[0], 2 * Th, 2 * Tw)[:, :, :H, :W]
    return Y + b.view(1, -1, 1, 1) if b is not None else Y


def make_conv(mode, stats):
    real = O.conv.__wrapped__ if hasattr(O.conv, "__wrapped__") else O.conv

    def conv(sd, p, x, pad=None):
        w = sd[p + ".weight"]
        b = sd.get(p + ".bias")
        if mode == "fp32" or w.dim() != 4:
            return real(sd, p, x, pad)
        k = w.shape[-1]
        if pad is None:
            pad = k // 2
        cin, cout = w.shape[1], w.shape[0]
        use = False
        if k == 3 and pad == 1 and x.shape[-1] >= 8:
            if mode == "wino_multi":
                use = cin > 64 and cout >= 48                       # the multi-chunk layers (128 / 192 / 512 channels in)
            elif mode == "wino_all":
                use = cin >= 48 and cout >= 48
        xb = bf(x)
        if use:
            stats[(cin, cout)] = stats.get((cin, cout), 0) + 1
            y = winograd_conv(xb, w, b)
        else:
            y = F.conv2d(xb, bf(w), b, stride=1, padding=pad)
        return bf(y)
    return conv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="540x960", help="packed HxW (1080x1920 = a 4K mosaic)")
    ap.add_argument("--net", default="LiteISPNet_GFM_LSC")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    h, w = map(int, a.size.split("x"))
    import realcamnet_amd as M
    torch.manual_seed(0)
    net = getattr(M, a.net)().eval()
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(1234)
    mosaic = torch.rand(1, 1, 2 * h, 2 * w, generator=g)
    packed, cond = O.raw_ingest(mosaic)
    coord = O.make_coord(1, h, w)
    real = O.conv
    outs = {}
    for mode in ("fp32", "bf16", "wino_multi", "wino_all"):
        stats = {}
        O.conv = make_conv(mode, stats) if mode != "fp32" else real
        t0 = time.time()
        with torch.no_grad():
            outs[mode] = O.run_padded(a.net, sd, packed, cond, coord)
        O.conv = real
        msg = f"{mode:11s} {time.time() - t0:7.1f} s"
        if mode != "fp32":
            msg += f"  PSNR vs fp32 oracle {O.psnr(outs[mode], outs['fp32']):.2f} dB"
        if mode.startswith("wino"):
            msg += f"  (vs bf16-direct {O.psnr(outs[mode], outs['bf16']):.2f} dB)  layers {dict(sorted(stats.items()))}"
        print(msg, flush=True)


if __name__ == "__main__":
    main()
