"""CPU oracle for the GroupMix attention block (GMA_Block).   *** TEST INFRASTRUCTURE ***

Functional fp32 PyTorch-CPU restatement over the reference state_dict; same rules as
oracle/liteisp_oracle.py (never imported by the product path).  Pinned by tests/golden/gma_*.npz,
generated from the imported reference by oracle/make_golden_gma.py.
Reference: kepengxu/RealCamNet models/groupmix.py (identical copy at models/raw2bit.py:98-142).
"""
from __future__ import annotations

from typing import Mapping, Tuple

import torch
import torch.nn.functional as F

SD = Mapping[str, torch.Tensor]
CRPE_WINDOWS = ((3, 2), (5, 3), (7, 3))     # (kernel, heads) -- models/groupmix.py:175


def _tokens_to_img(x: torch.Tensor, hw: Tuple[int, int]) -> torch.Tensor:
    b, n, c = x.shape
    return x.transpose(1, 2).reshape(b, c, hw[0], hw[1])


def _img_to_tokens(x: torch.Tensor) -> torch.Tensor:
    return x.flatten(2).transpose(1, 2)


def conv_pos_enc(sd: SD, p: str, x: torch.Tensor, hw) -> torch.Tensor:
    """Depthwise 3x3 (+bias) plus identity.  models/groupmix.py:203-217."""
    img = _tokens_to_img(x, hw)
    return _img_to_tokens(F.conv2d(img, sd[p + ".proj.weight"], sd[p + ".proj.bias"], padding=1, groups=img.shape[1]) + img)


def _bn_eval(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """SyncBatchNorm in eval = affine with running stats.  models/groupmix.py:64-76."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _sep_conv(sd: SD, p: str, x: torch.Tensor, k: int) -> torch.Tensor:
    """SeparableConv2d: depthwise kxk (no bias) then pointwise 1x1 (no bias).  models/groupmix.py:240-249."""
    y = F.conv2d(x, sd[p + ".conv1.weight"], None, padding=k // 2, groups=x.shape[1])
    return F.conv2d(y, sd[p + ".pointwise_conv.weight"], None)


def aggregator(sd: SD, p: str, qkv: torch.Tensor, hw, heads: int):
    """models/groupmix.py:82-105.  qkv (3B,N,C) -> (3,B,heads,N,Ch), local (B,N,C/5)."""
    b3, n, c = qkv.shape
    b = b3 // 3
    seg = c // 5
    img = _tokens_to_img(qkv, hw)
    g = img.split([seg] * 5, dim=1)
    loc = g[4].reshape(3, b, seg, *hw).permute(1, 0, 2, 3, 4).reshape(b, 3 * seg, *hw)
    loc = _sep_conv(sd, p + ".agg0.conv", loc, 3)
    loc = F.layer_norm(_img_to_tokens(loc), (seg,), sd[p + ".agg0.norm.weight"], sd[p + ".agg0.norm.bias"], 1e-5)
    loc = F.hardswish(loc)
    x0 = F.hardswish(_bn_eval(sd, p + ".norm0", g[0]))
    x1 = F.hardswish(_bn_eval(sd, p + ".norm1", _sep_conv(sd, p + ".agg1", g[1], 3)))
    x2 = F.hardswish(_bn_eval(sd, p + ".norm2", _sep_conv(sd, p + ".agg2", g[2], 5)))
    x3 = F.hardswish(_bn_eval(sd, p + ".norm3", _sep_conv(sd, p + ".agg3", g[3], 7)))
    x = torch.cat([x0, x1, x2, x3], dim=1)
    ct = c // 5 * 4
    x = x.reshape(3, b, heads, ct // heads, n).permute(0, 1, 2, 4, 3)
    return x, loc


def conv_rel_pos_enc(sd: SD, p: str, q: torch.Tensor, v: torch.Tensor, hw) -> torch.Tensor:
    """q * depthwise_conv(v) with per-head-group windows 3/5/7.  models/groupmix.py:138-156."""
    b, h, n, ch = q.shape
    img = v.permute(0, 1, 3, 2).reshape(b, h * ch, *hw)           # 'B h (H W) Ch -> B (h Ch) H W'
    outs, c0 = [], 0
    for i, (k, nh) in enumerate(CRPE_WINDOWS):
        cw = nh * ch
        part = img[:, c0:c0 + cw]
        outs.append(F.conv2d(part, sd[f"{p}.conv_list.{i}.weight"], sd[f"{p}.conv_list.{i}.bias"], padding=k // 2, groups=cw))
        c0 += cw
    conv_v = torch.cat(outs, dim=1).reshape(b, h, ch, n).permute(0, 1, 3, 2)
    return q * conv_v


def efficient_att(sd: SD, p: str, x: torch.Tensor, hw, heads: int) -> torch.Tensor:
    """Linear attention with multi-scale aggregators.  models/groupmix.py:177-200."""
    b, n, c = x.shape
    qkv = F.linear(x, sd[p + ".qkv.weight"]).reshape(b, n, 3, c).permute(2, 0, 1, 3).reshape(3 * b, n, c)
    qkv, loc = aggregator(sd, p + ".aggregator", qkv, hw, heads)
    q, k, v = qkv[0], qkv[1], qkv[2]
    ks = k.softmax(dim=2)
    ktv = torch.einsum("bhnk,bhnv->bhkv", ks, v)
    att = torch.einsum("bhnk,bhkv->bhnv", q, ktv)
    crpe = conv_rel_pos_enc(sd, p + ".crpe", q, v, hw)
    scale = (c // heads) ** -0.5
    y = (scale * att + crpe).transpose(1, 2).reshape(b, n, c // 5 * 4)
    y = torch.cat([y, loc], dim=-1)
    return F.linear(y, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def gma_block(sd: SD, x: torch.Tensor, hw, heads: int = 8, p: str = "") -> torch.Tensor:
    """GMA_Block.forward (models/groupmix.py:289-299): cpe -> LN -> att -> + ; LN -> MLP(GELU) -> +."""
    c = x.shape[-1]
    x = conv_pos_enc(sd, p + "cpe", x, hw)
    cur = F.layer_norm(x, (c,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    x = x + efficient_att(sd, p + "att", cur, hw, heads)
    cur = F.layer_norm(x, (c,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    cur = F.linear(F.gelu(F.linear(cur, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + cur
