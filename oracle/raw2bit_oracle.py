"""CPU oracle for the RAW codec's own blocks and `raw_compression_tcm[_final].forward`.
*** TEST INFRASTRUCTURE *** -- only tests/ (and the fixture generators here) may import this file.

Functional fp32 PyTorch-CPU restatement over the reference's state_dict; every function cites models/raw2bit.py.  The layers
that come from CompressAI (not in the upstream tree) are the restatements of oracle/tcm_oracle.py: parity UNPINNED for those;
what the fixtures of oracle/make_golden_raw2bit.py pin is upstream's own code in this file's functions.
"""
from __future__ import annotations

from typing import Mapping

import torch
import torch.nn.functional as F

import liteisp_oracle as LO
import tcm_oracle as TO

SD = Mapping[str, torch.Tensor]


def _conv(sd: SD, p: str, x, stride=1, pad=None):
    w = sd[p + ".weight"]
    return F.conv2d(x, w, sd.get(p + ".bias"), stride=stride, padding=(w.shape[-1] // 2 if pad is None else pad))


def residual_block_with_ca(sd: SD, p: str, x):
    """models/raw2bit.py:257-289 with CALayer :238-254 (bias-free Linear pair, reduction 8)."""
    out = F.leaky_relu(_conv(sd, p + ".conv1", x), 0.01)
    out = _conv(sd, p + ".conv2", out)
    y = out.mean(dim=(2, 3))
    y = torch.sigmoid(F.linear(F.relu(F.linear(y, sd[p + ".ca.fc.0.weight"])), sd[p + ".ca.fc.2.weight"]))
    out = out * y[:, :, None, None]
    identity = _conv(sd, p + ".skip", x) if (p + ".skip.weight") in sd else x
    return out + identity


def spatial_feature_transform(sd: SD, p: str, x, cond):
    """models/raw2bit.py:877-885 ('vanilla', residual=True): x*scale(cond) + shift(cond) + x."""
    scale = _conv(sd, p + ".cond_scale.2", F.relu(_conv(sd, p + ".cond_scale.0", cond)))
    shift = _conv(sd, p + ".cond_shift.2", F.relu(_conv(sd, p + ".cond_shift.0", cond)))
    return x * scale + shift + x


def conv_trans_block_mzj(sd: SD, p: str, x, cond, conv_dim: int, trans_dim: int, head_dim: int, ws: int, typ: str):
    """models/raw2bit.py:306-328."""
    pre = p + "." if p else ""
    t = _conv(sd, pre + "conv1_1", x)
    conv_x, trans_x = torch.split(t, (conv_dim, trans_dim), dim=1)
    conv_identity = conv_x
    conv_x = residual_block_with_ca(sd, pre + "conv_block", conv_x)
    conv_x = spatial_feature_transform(sd, pre + "spatial_transform", conv_x, cond) + conv_identity
    trans_x = TO.block(sd, pre + "trans_block", trans_x.permute(0, 2, 3, 1), head_dim, ws, typ).permute(0, 3, 1, 2)
    res = _conv(sd, pre + "conv1_2", torch.cat((conv_x, trans_x), dim=1))
    return x + res


def hybrid_condition_module(sd: SD, p: str, x):
    """models/raw2bit.py:817-858 with the HyCondMod blocks of :730-813."""
    pre = p + "." if p else ""
    cb = lambda q, t, stride=1: F.relu(_conv(sd, pre + q + ".conv", t, stride=stride))
    enc = lambda q, t: cb(q + ".conv", cb(q + ".down", t, 2))

    def dec(q, t1, t2):
        up = cb(q + ".up.1", F.interpolate(t1, scale_factor=2, mode="bilinear", align_corners=True))
        return cb(q + ".conv", torch.cat([t2, up], dim=1))

    x1 = cb("in_conv", x)
    x2 = enc("enc_1", x1); x3 = enc("enc_2", x2); x4 = enc("enc_3", x3)
    y = dec("dec_1", x4, x3); y = dec("dec_2", y, x2); y = dec("dec_3", y, x1)
    y = cb("out_conv", y)
    lr = lambda t: F.leaky_relu(t, 0.1)
    c1 = _conv(sd, pre + "CondNet1.2", lr(_conv(sd, pre + "CondNet1.0", y, stride=2)))
    c2 = _conv(sd, pre + "CondNet2.2", lr(_conv(sd, pre + "CondNet2.0", y, stride=2)), stride=2)
    c3 = _conv(sd, pre + "CondNet3.4", lr(_conv(sd, pre + "CondNet3.2", lr(_conv(sd, pre + "CondNet3.0", y, stride=2)), stride=2)), stride=2)
    return [c1, c2, c3]


def raw_compression_tcm_final(sd: SD, x, N: int = 64, num_slices: int = 5, max_support_slices: int = 5,
                              config=(2, 2, 2, 2, 2, 2, 2), head_dim=(8, 16, 32, 32, 16, 8, 8)):
    """raw_compression_tcm_final.forward, models/raw2bit.py:1768-1855 (eval)."""
    raw, cond, coord = x
    sub = lambda pre: {k[len(pre) + 1:]: v for k, v in sd.items() if k.startswith(pre + ".")}
    fea = _conv(sd, "conv_first", raw)
    vec = LO.color_condition_gfm(sd, "classifier", cond)
    lsc_fea = LO.lens_shading(sd, "lsc", coord)
    local = hybrid_condition_module(sd, "local_condition", raw)
    fea = fea * (lsc_fea + 1)
    fea = TO.residual_block_with_stride(sd, "conv_down", fea)
    for s in range(3):
        fea = LO.res_gfm(sd, f"gfm{s + 1}.0", fea, vec)
        for i in range(config[s]):
            fea = conv_trans_block_mzj(sd, f"m_down{s + 1}.{i}", fea, local[s], N, N, head_dim[s], 8, "W" if not i % 2 else "SW")
        if s < 2:
            fea = TO.residual_block_with_stride(sd, f"m_down{s + 1}_down", fea)
        else:
            fea = _conv(sd, "m_down3_down", fea, stride=2)
    y = fea
    specs = _codec_specs(N, config, head_dim)
    out = slice_loop(sd, y, specs, num_slices, max_support_slices)
    out.update({"y": y, "lft": local[2], "lsc": lsc_fea})
    return out


def _codec_specs(N, config, head_dim):
    ctb = lambda n, hd, ws: [("ctb", N, hd, ws, "W" if not i % 2 else "SW") for i in range(n)]
    return {
        "g_s": [("rbu",)] + ctb(config[3], head_dim[3], 8) + [("rbu",)] + ctb(config[4], head_dim[4], 8) + [("rbu",)] + ctb(config[5], head_dim[5], 8) +
               [("subpel",), ("rb",), ("subpel",)],
        "h_a": [("rbws",)] + ctb(config[0], 32, 4) + [("conv3x3s2",)],
        "h_s": [("rbu",)] + ctb(config[3], 32, 4) + [("subpel",)],
    }


def raw_compression_tcm(sd: SD, x, N: int = 64, num_slices: int = 5, max_support_slices: int = 5,
                        config=(2, 2, 2, 2, 2, 2, 2), head_dim=(8, 16, 32, 32, 16, 8, 8)):
    """raw_compression_tcm.forward, models/raw2bit.py:491-579 (eval): no local condition, plain ConvTransBlocks in m_down1..3
    (built at :385-400 with their stride-2 stage as the Sequential's last entry)."""
    raw, cond, coord = x
    sub = lambda pre: {k[len(pre) + 1:]: v for k, v in sd.items() if k.startswith(pre + ".")}
    fea = _conv(sd, "conv_first", raw)
    vec = LO.color_condition_gfm(sd, "classifier", cond)
    fea = fea * (LO.lens_shading(sd, "lsc", coord) + 1)
    fea = TO.residual_block_with_stride(sd, "conv_down", fea)
    for s in range(3):
        fea = LO.res_gfm(sd, f"gfm{s + 1}.0", fea, vec)
        spec = [("ctb", N, head_dim[s], 8, "W" if not i % 2 else "SW") for i in range(config[s])] + [("rbws",) if s < 2 else ("conv3x3s2",)]
        fea = TO.run_transform(sub(f"m_down{s + 1}"), "", spec, fea)
    return slice_loop(sd, fea, _codec_specs(N, config, head_dim), num_slices, max_support_slices)


def gma_block_nchw(sd: SD, p: str, x, heads: int):
    """GMA_Block (models/raw2bit.py:117-143 == models/groupmix.py:274-299) over the tokens of an NCHW map, as GMABlock (:178-184) and
    ConvGMABlock (:349-352) call it."""
    import groupmix_oracle as GO
    b, c, h, w = x.shape
    tok = GO.gma_block(sd, x.flatten(2).transpose(1, 2), (h, w), heads, p=p + ".")
    return tok.transpose(1, 2).reshape(b, c, h, w)


def gma_atten(sd: SD, p: str, x, head_dim: int):
    """GMAAtten.forward with inter_dim set, models/raw2bit.py:225-234 (conv_a / conv_b from CompressAI's AttentionBlock, restated)."""
    pre = p + "." if p else ""
    x = _conv(sd, pre + "in_conv", x)
    heads = x.shape[1] // head_dim
    z = gma_block_nchw(sd, pre + "non_local_block.block_2", gma_block_nchw(sd, pre + "non_local_block.block_1", x, heads), heads)
    a = TO.attention_branch(sd, pre + "conv_a", x, False)
    b = TO.attention_branch(sd, pre + "conv_b", z, True)
    return _conv(sd, pre + "out_conv", a * torch.sigmoid(b) + x)


def conv_gma_block(sd: SD, p: str, x, conv_dim: int, trans_dim: int, head_dim: int):
    """ConvGMABlock.forward, models/raw2bit.py:345-355."""
    pre = p + "." if p else ""
    conv_x, trans_x = torch.split(_conv(sd, pre + "conv1_1", x), (conv_dim, trans_dim), dim=1)
    conv_x = TO.residual_block(sd, pre + "conv_block", conv_x) + conv_x
    trans_x = gma_block_nchw(sd, pre + "trans_block", trans_x, trans_dim // head_dim)
    return x + _conv(sd, pre + "conv1_2", torch.cat((conv_x, trans_x), dim=1))


def rbu(sd: SD, p: str, x):
    """RBU.forward, models/raw2bit.py:3198-3206: ResidualBlockUpsample without the IGDN."""
    pre = p + "." if p else ""
    out = F.leaky_relu(TO.subpel_conv3x3(sd, pre + "subpel_conv", x), 0.01)
    return _conv(sd, pre + "conv", out) + TO.subpel_conv3x3(sd, pre + "upsample", x)


def slice_loop(sd: SD, y, specs, num_slices: int, max_support_slices: int):
    """models/raw2bit.py:1791-1846 (identical to models/tcm.py:439-486)."""
    sub = lambda pre: {k[len(pre) + 1:]: v for k, v in sd.items() if k.startswith(pre + ".")}
    z = TO.run_transform(sub("h_a"), "", specs["h_a"], y)
    _, z_lik = TO.entropy_bottleneck(sd, "entropy_bottleneck", z)
    z_offset = sd["entropy_bottleneck.quantiles"][:, :, 1:2].reshape(1, -1, 1, 1)
    z_hat = TO.ste_round(z - z_offset) + z_offset
    latent_scales = TO.run_transform(sub("h_scale_s"), "", specs["h_s"], z_hat)
    latent_means = TO.run_transform(sub("h_mean_s"), "", specs["h_s"], z_hat)
    hh, ww = y.shape[2:]
    y_hat_slices, y_lik, mu_list, scale_list = [], [], [], []
    for i, y_slice in enumerate(y.chunk(num_slices, 1)):
        support = y_hat_slices if max_support_slices < 0 else y_hat_slices[:max_support_slices]
        mean_support = TO.swatten(sd, f"atten_mean.{i}.0", torch.cat([latent_means] + support, dim=1), 16, 8)
        mu = TO.slice_transform(sd, f"cc_mean_transforms.{i}", mean_support)[:, :, :hh, :ww]
        scale_support = TO.swatten(sd, f"atten_scale.{i}.0", torch.cat([latent_scales] + support, dim=1), 16, 8)
        scale = TO.slice_transform(sd, f"cc_scale_transforms.{i}", scale_support)[:, :, :hh, :ww]
        _, lik = TO.gaussian_conditional(y_slice, scale, mu)
        y_hat_slice = TO.ste_round(y_slice - mu) + mu
        lrp = TO.slice_transform(sd, f"lrp_transforms.{i}", torch.cat([mean_support, y_hat_slice], dim=1))
        y_hat_slice = y_hat_slice + 0.5 * torch.tanh(lrp)
        y_hat_slices.append(y_hat_slice); y_lik.append(lik); mu_list.append(mu); scale_list.append(scale)
    x_hat = TO.run_transform(sub("g_s"), "", specs["g_s"], torch.cat(y_hat_slices, dim=1))
    return {"x_hat": x_hat, "likelihoods": {"y": torch.cat(y_lik, dim=1), "z": z_lik},
            "para": {"means": torch.cat(mu_list, dim=1), "scales": torch.cat(scale_list, dim=1), "y": y}}
