"""CPU: the oracle restatement vs fixtures produced by the imported reference (oracle/make_golden.py)."""
import pytest
import torch
import torch.nn.functional as F

import liteisp_oracle as O
from conftest import net_name_of, golden_names, load_golden, sd_digest, seed0_state_dict

torch.set_num_threads(max(1, min(8, torch.get_num_threads())))

# the oracle equalled the reference bitwise when the fixtures were made (same torch, 1 thread); across
# thread counts oneDNN may reorder sums, hence a tight tolerance instead of equality.
TOL = dict(rtol=0, atol=2e-6)


def _close(a, b, scale_tol=3e-6):
    assert a.shape == b.shape
    err = (a - b).abs().max().item()
    assert err <= scale_tol * max(1.0, b.abs().max().item()), err


def test_dwt_blocks():
    g = load_golden("block_dwt_forward")
    _close(O.dwt_forward({"w.weight": g["sd"]["weight"]}, "w", g["x"]), g["y"])
    g = load_golden("block_dwt_inverse")
    _close(O.dwt_inverse({"w.weight": g["sd"]["weight"]}, "w", g["x"]), g["y"])


def test_dwt_any_channel_count_pair():
    """DWTForward_ / DWTInverse_ (upstream models/networks.py:9-47): one (4,1,2,2) tap set for any channel count."""
    g = load_golden("block_dwt_forward_anyc")
    _close(O.dwt_forward_(g["sd"]["weight"], g["x"]), g["y"])
    gi = load_golden("block_dwt_inverse_anyc")
    _close(O.dwt_inverse_(gi["sd"]["weight"], gi["x"]), gi["y"])
    assert tuple(g["sd"]["weight"].shape) == (4, 1, 2, 2)
    _close(O.dwt_inverse_(gi["sd"]["weight"], O.dwt_forward_(g["sd"]["weight"], g["x"])), g["x"])       # analysis -> synthesis = identity


def test_dwt_roundtrip_is_identity():
    g = load_golden("block_dwt_forward")
    gi = load_golden("block_dwt_inverse")
    x = g["x"]
    y = O.dwt_forward({"w.weight": g["sd"]["weight"]}, "w", x)
    back = O.dwt_inverse({"w.weight": gi["sd"]["weight"][: 4 * x.shape[1]]}, "w", y)
    _close(back, x)


def test_conv_blocks():
    g = load_golden("block_conv3x3_16_32")
    _close(O.conv({"c.weight": g["sd"]["weight"], "c.bias": g["sd"]["bias"]}, "c", g["x"]), g["y"])
    g = load_golden("block_conv_crc_48")
    sd = g["sd"]
    y = O.conv(sd, "2", torch.relu(O.conv(sd, "0", g["x"])))
    _close(y, g["y"])


def test_channel_attention_blocks():
    g = load_golden("block_calayer_32")
    _close(O.ca_layer({"ca." + k: v for k, v in g["sd"].items()}, "ca", g["x"]), g["y"])
    g = load_golden("block_rcab_32")
    _close(O.rcab({"b." + k: v for k, v in g["sd"].items()}, "b", g["x"]), g["y"])
    g = load_golden("block_rcag_32_nb4")
    _close(O.rcag({"g." + k: v for k, v in g["sd"].items()}, "g", g["x"], nb=4), g["y"])
    g = load_golden("block_rcag_48_nb2")
    _close(O.rcag({"g." + k: v for k, v in g["sd"].items()}, "g", g["x"], nb=2), g["y"])


def test_conditioning_blocks():
    g = load_golden("block_res_gfm_48")
    _close(O.res_gfm({"m." + k: v for k, v in g["sd"].items()}, "m", g["x"], g["v"]), g["y"])
    g = load_golden("block_lsc_48")
    _close(O.lens_shading({"l." + k: v for k, v in g["sd"].items()}, "l", g["x"]), g["y"])
    g = load_golden("block_color_condition")
    _close(O.color_condition_gfm({"c." + k: v for k, v in g["sd"].items()}, "c", g["x"]), g["y"])


def test_gfm_lfm_blocks():
    """Oracle restatements of the global + local modulation blocks (models/LiteISP.py:215-230, 293-321, 501-534, 601-620)."""
    pre = lambda g: {"m." + k: v for k, v in g["sd"].items()}
    g = load_golden("block_res_gfm_lfm_64")
    _close(O.res_gfm_lfm(pre(g), "m", g["x"], g["v"], g["cmap"]), g["y"])
    g = load_golden("block_sftlayer_32")
    _close(O.sft_layer(pre(g), "m", g["x"], g["cmap"]), g["y"])
    g = load_golden("block_gfmlayer_128")
    sd, c = pre(g), g["x"].shape[1]
    s, t = (O._gfm_vec(sd, "m", w, g["v"]).view(-1, c, 1, 1) for w in ("scale", "shift"))
    _close(g["x"] * s + t + g["x"], g["y"])
    g = load_golden("block_color_condition_gfm_lfm")
    vec, lfm = O.color_condition_gfm_lfm(pre(g), "m", g["x"], g["local"])
    _close(vec, g["y"]); _close(lfm, g["lfm"])
    g = load_golden("block_cb_4_16")
    vec_sd = {"m.downblocks.0." + k: v for k, v in g["sd"].items()}
    h = O.conv(vec_sd, "m.downblocks.0.conv", g["x"])
    h = F.leaky_relu(F.avg_pool2d(h, 3, stride=2, padding=1, count_include_pad=True), 0.2)
    _close(F.instance_norm(h, weight=g["sd"]["norm.weight"], bias=g["sd"]["norm.bias"], use_input_stats=True, eps=1e-5), g["y"])


def test_tail_and_padding():
    g = load_golden("block_tail_16")
    sd = g["sd"]
    y = O.conv(sd, "2", O.pixel_shuffle2(O.conv(sd, "0", g["x"])))
    _close(y, g["y"])
    g = load_golden("block_pad16")
    x = torch.zeros(*[int(v) for v in g["x_shape"]])
    yp, hw = O.pad_to_multiple(x, 16)
    assert yp.shape == g["y"].shape and tuple(int(v) for v in g["hw"]) == hw
    assert O.remove_padding(torch.zeros(1, 3, 2 * yp.shape[2], 2 * yp.shape[3]), hw).shape[-2:] == (2 * hw[0], 2 * hw[1])


def test_bayer_unshuffle_matches_pixel_unshuffle():
    x = torch.rand(2, 1, 12, 20)
    assert torch.equal(O.bayer_unshuffle(x), torch.nn.functional.pixel_unshuffle(x, 2))


@pytest.mark.parametrize("fixture", golden_names("e2e_"))
def test_end_to_end_vs_reference(fixture):
    g = load_golden(fixture)
    name = net_name_of(fixture)
    sd = seed0_state_dict(name)
    assert sd_digest(sd) == g["sd_digest"], "mirror module's seed-0 parameters differ from the reference's"
    with torch.no_grad():
        y = O.FORWARDS[name](sd, [g["raw"], g["cond"], g["coord"]])
    _close(y, g["y"], scale_tol=1e-5)
    assert O.psnr(y, g["y"]) > 120.0
