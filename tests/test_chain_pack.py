"""rc_chain_pack_weights (host function of librealcam_hip.so): the fragment order that lets a layer's MFMA accumulator fragments be
the next layer's B fragments (csrc/gma_fused.hip).  Checked on the CPU by emulating the lane-level MFMA operand rules
(v_mfma_f32_16x16x32_bf16: A lane (R = l & 15, q = l >> 4) holds k = 8 q + i; B lane (n, q) holds k = 8 q + i; D lane (n, g) holds rows
4 g + j; the K = 16 form holds k = 4 q + i) over the packed bytes, for a two-layer chain 80 -> 320 -> 80 and a 80 -> 240 layer."""
import ctypes as C

import numpy as np
import torch

from realcamnet_amd import _lib


def _bf16_bits_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def _pack(w):
    L = _lib.load()
    cout, cin = w.shape
    buf = np.zeros(L.rc_chain_packed_bytes(cin, cout), dtype=np.uint8)
    w32 = np.ascontiguousarray(w, dtype=np.float32)
    assert L.rc_chain_pack_weights(w32.ctypes.data, cin, cout, buf.ctypes.data) == 0
    return buf.view(np.uint16)


def _layer(frags, cin, cout, act):
    """act: dict step -> (64 lanes, 8) floats (B fragments of one 16-token column tile; 'tail' -> (64, 4)).  Returns per output
    tile m the D fragment (64 lanes, 4)."""
    ks, tail = cin // 32, cin % 32 != 0
    tile_elems = ks * 512 + (256 if tail else 0)
    mt = (cout + 15) // 16
    out = []
    for m in range(mt):
        tile = _bf16_bits_to_f32(frags[m * tile_elems:(m + 1) * tile_elems])
        D = np.zeros((16, 16), dtype=np.float64)                       # [row R][col n]
        for s in range(ks):
            A = tile[s * 512:(s + 1) * 512].reshape(64, 8)
            Amat = np.zeros((16, 32)); Bmat = np.zeros((32, 16))
            for l in range(64):
                Amat[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = A[l]
                Bmat[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = act[s][l]
            D += Amat @ Bmat
        if tail:
            A = tile[ks * 512:].reshape(64, 4)
            Amat = np.zeros((16, 16)); Bmat = np.zeros((16, 16))
            for l in range(64):
                Amat[l & 15, 4 * (l >> 4):4 * (l >> 4) + 4] = A[l]
                Bmat[4 * (l >> 4):4 * (l >> 4) + 4, l & 15] = act["tail"][l]
            D += Amat @ Bmat
        frag = np.zeros((64, 4))
        for l in range(64):
            frag[l] = D[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
        out.append(frag)
    return out


def _to_act(x, c):
    """x (16 tokens, c) natural order -> B fragments."""
    act = {}
    for s in range(c // 32):
        f = np.zeros((64, 8))
        for l in range(64):
            f[l] = x[l & 15, 32 * s + 8 * (l >> 4):32 * s + 8 * (l >> 4) + 8]
        act[s] = f
    if c % 32:
        f = np.zeros((64, 4))
        for l in range(64):
            f[l] = x[l & 15, 32 * (c // 32) + 4 * (l >> 4):32 * (c // 32) + 4 * (l >> 4) + 4]
        act["tail"] = f
    return act


def _pairs_to_act(tiles):
    """D fragments of output tiles -> next layer's B fragments: tiles (2p, 2p+1) -> step p; an unpaired last tile -> the K=16 tail."""
    act = {}
    for p in range(len(tiles) // 2):
        act[p] = np.concatenate([tiles[2 * p], tiles[2 * p + 1]], axis=1)
    if len(tiles) % 2:
        act["tail"] = tiles[-1]
    return act


def _act_to_natural(act, c):
    x = np.zeros((16, c))
    for s in range(c // 32):
        for l in range(64):
            x[l & 15, 32 * s + 8 * (l >> 4):32 * s + 8 * (l >> 4) + 8] = act[s][l]
    if c % 32:
        for l in range(64):
            x[l & 15, 32 * (c // 32) + 4 * (l >> 4):32 * (c // 32) + 4 * (l >> 4) + 4] = act["tail"][l]
    return x


def _bf16(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32)).bfloat16().float().numpy()


def test_two_layer_chain_through_packed_fragments():
    rng = np.random.default_rng(0)
    w1, w2 = _bf16(rng.standard_normal((320, 80)) * 0.2), _bf16(rng.standard_normal((80, 320)) * 0.1)
    x = _bf16(rng.standard_normal((16, 80)))
    h = _pairs_to_act(_layer(_pack(w1), 80, 320, _to_act(x, 80)))
    assert set(h) == set(range(10))
    y = _act_to_natural(_pairs_to_act(_layer(_pack(w2), 320, 80, h)), 80)
    want = (x.astype(np.float64) @ w1.T.astype(np.float64)) @ w2.T.astype(np.float64)
    assert np.abs(y - want).max() <= 1e-9 * max(1.0, np.abs(want).max())


def test_odd_tile_count_and_bias_order():
    rng = np.random.default_rng(1)
    w = _bf16(rng.standard_normal((240, 80)) * 0.2)                       # 15 output tiles: 7 pairs + an unpaired natural-order tile
    x = _bf16(rng.standard_normal((16, 80)))
    y = _act_to_natural(_pairs_to_act(_layer(_pack(w), 80, 240, _to_act(x, 80))), 240)
    assert np.abs(y - x.astype(np.float64) @ w.T.astype(np.float64)).max() <= 1e-9
    L = _lib.load()
    b = np.arange(80, dtype=np.float32)
    rows = L.rc_chain_packed_rows(80)
    assert rows == 80
    out = np.zeros(rows, dtype=np.float32)
    assert L.rc_chain_pack_bias(b.ctypes.data, 80, out.ctypes.data) == 0
    # lane (n, g) reads bias4(tile m) = out[16 m + 4 g .. + 4): it must be the channels of that lane's accumulator rows
    tiles = [np.repeat(out[16 * m:16 * m + 16].reshape(4, 4), 16, axis=0).reshape(4, 16, 4).transpose(1, 0, 2).reshape(64, 4)
             for m in range(5)]                                         # fragment lane l = 16 g + n  ->  rows 4 g + j
    frag_as_lane = [np.stack([out[16 * m + 4 * (l >> 4):16 * m + 4 * (l >> 4) + 4] for l in range(64)]) for m in range(5)]
    nat = _act_to_natural(_pairs_to_act(frag_as_lane), 80)
    assert np.array_equal(nat, np.tile(b, (16, 1)))
    del tiles
