"""f3: CDF tables, symbol preparation and the rANS streams of compress() / decompress().

Integer work: the bar is BIT-EXACT against the oracle (oracle/rans_oracle.c, oracle/entropy_oracle.py -- CompressAI's published
algorithms restated; the package is absent and unpinned upstream, so the wire format is parity-unpinned against CompressAI itself).
CPU: host-side functions of the library (table quantiser, the single-stream "compressai" coder) and the mirror's update() /
strict load.  GPU: symbol kernels, the chunked coder byte for byte, escapes, and compress -> decompress round trips of both codecs."""
import struct

import numpy as np
import pytest
import torch

import entropy_oracle as E
from det_fill import det_fill_
from realcamnet_amd import _lib, bitstream


def _tables(kind="gc"):
    t = E.gc_update(E.get_scale_table())
    return t, bitstream.Tables(t["_quantized_cdf"], t["_cdf_length"], t["_offset"], "cpu")


def _symbols(n, t, seed=0, escapes=True):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 64, n).astype(np.int32)
    sym = np.round(rng.standard_normal(n) * t["scale_table"].numpy()[idx] * 1.3).astype(np.int32)
    if escapes:                                                   # far outside the table's support: 4-bit bypass escapes, up to 8 nibbles
        sym[::97] += 3000; sym[5::211] -= 70000; sym[7::1009] = 2 ** 27; sym[11::1013] = -(2 ** 27)
    return sym, idx


def test_pmf_to_quantized_cdf_equals_oracle():
    L = _lib.load()
    rng = np.random.default_rng(3)
    for n in (1, 2, 5, 33, 400):
        for _ in range(6):
            pmf = rng.random(n).astype(np.float32) ** 8
            pmf[rng.integers(0, n, max(1, n // 4))] = 0.0                # zero-probability symbols must still get a frequency
            pmf[0] = max(pmf[0], 1e-3)
            pmf /= pmf.sum()
            got = np.zeros(n + 1, dtype=np.int32)
            assert L.rc_pmf_to_quantized_cdf(pmf.ctypes.data, n, 16, got.ctypes.data) == 0
            want = E.pmf_to_quantized_cdf(pmf)
            assert np.array_equal(got, want)
            assert got[0] == 0 and got[-1] == 65536 and (np.diff(got) > 0).all()


def test_host_coder_equals_oracle_stream_and_round_trips():
    t, tables = _tables()
    for n, seed in ((1, 1), (17, 2), (5000, 3), (60000, 4)):
        sym, idx = _symbols(n, t, seed)
        stream = bitstream.encode(torch.from_numpy(sym), torch.from_numpy(idx), tables, "compressai")
        assert stream == E.encode_with_indexes(sym, idx, t)              # CompressAI layout, byte for byte
        dec = bitstream.Decoder(stream, tables, "cpu", "compressai")
        cut = n // 3                                                     # decode_stream semantics: the state carries over
        got = torch.cat([dec.decode(torch.from_numpy(idx[:cut])), dec.decode(torch.from_numpy(idx[cut:]))]).numpy()
        assert np.array_equal(got, sym)
        o = E.Decoder(stream)
        assert np.array_equal(o.decode_stream(idx, t), sym)


def test_division_free_put_equals_the_dividing_put():
    """The GPU encoder's state update is ryg's reciprocal form (csrc/rans.hip put_rcp); it must equal CompressAI's dividing Rans64EncPut for
    every reachable state: checked on 2 M random (state, start, freq) incl. freq 1..3, powers of two and the two ends of the state range."""
    assert _lib.load().rc_debug_rans_rcp_selftest(2_000_000, 12345) == 0


def test_truncated_or_corrupt_streams_are_errors_not_overreads():
    """a short / corrupt string must raise, never read past the stream (host coder here; the chunked container's header is
    validated before upload and the kernel bounds every chunk: GPU test below)."""
    t, tables = _tables()
    sym, idx = _symbols(4000, t, 9)
    stream = bitstream.encode(torch.from_numpy(sym), torch.from_numpy(idx), tables, "compressai")
    for bad in (stream[:4], stream[:len(stream) // 2 // 4 * 4], stream[:-3], b""):
        with pytest.raises((ValueError, _lib.HipError)):
            bitstream.Decoder(bad, tables, "cpu", "compressai").decode(torch.from_numpy(idx))
    # chunked container: every malformed header is rejected on the host
    good = struct_pack_container(4000, 2048, [4000, 4000])
    for mutate in (lambda b: b[:10], lambda b: b[:20],                                   # truncated header / size table
                   lambda b: b[:16] + (7).to_bytes(4, "little") + b[20:],               # a chunk shorter than its flushed state
                   lambda b: b[:16] + (4002).to_bytes(4, "little") + b[20:],            # not word-sized
                   lambda b: b[:-8],                                                    # payload truncated
                   lambda b: b[:8] + (0).to_bytes(4, "little") + b[12:]):               # chunk = 0
        with pytest.raises(ValueError):
            bitstream.Decoder(mutate(good), tables, "cpu", "chunked").decode(torch.from_numpy(idx))


def struct_pack_container(n, chunk, sizes):
    import struct
    return struct.pack("<4sIII", bitstream.MAGIC, n, chunk, len(sizes)) + np.asarray(sizes, "<u4").tobytes() + bytes(sum(sizes))


def test_coder_tables_do_not_depend_on_the_order_of_cast_and_update():
    """`.to(bfloat16)` used to round GaussianConditional.scale_table (63 of 64 thresholds moved) and, after a cast, update() built
    the EntropyBottleneck tables from rounded parameters -- two processes with the same weights but another order of .to(bf16) / update() /
    load_state_dict produced streams the other could not decode.  The coder now reads fp32 masters (tcm._Fp32Masters)."""
    from realcamnet_amd import tcm
    gc = tcm.GaussianConditional(None)
    gc.update_scale_table(bitstream.get_scale_table())
    want = gc.scale_table.clone()
    gc.to(torch.bfloat16)
    assert gc.scale_table.dtype == torch.float32 and torch.equal(gc._table("cpu"), want)      # the buffer itself stays fp32
    gc2 = tcm.GaussianConditional(None).to(torch.bfloat16)
    gc2.update_scale_table(bitstream.get_scale_table())
    assert torch.equal(gc2._table("cpu"), want) and torch.equal(gc2._quantized_cdf, gc._quantized_cdf)

    torch.manual_seed(3)
    eb = tcm.EntropyBottleneck(8)
    with torch.no_grad():
        eb.quantiles.add_(torch.randn_like(eb.quantiles) * 0.37)
        for i in range(5):
            getattr(eb, f"_matrix{i}").add_(torch.randn_like(getattr(eb, f"_matrix{i}")) * 0.1)
    sd = {k: v.clone() for k, v in eb.state_dict().items() if k.split(".")[-1] not in ("_offset", "_quantized_cdf", "_cdf_length")}
    eb.update()
    cdf, med = eb._quantized_cdf.clone(), eb.quantiles[:, 0, 1].clone()
    a = tcm.EntropyBottleneck(8); a.load_state_dict(sd, strict=False); a = a.to(torch.bfloat16); a.update(force=True)       # load, cast, update
    b = tcm.EntropyBottleneck(8).to(torch.bfloat16); b.load_state_dict(sd, strict=False); b.update(force=True)             # cast, load, update
    c = tcm.EntropyBottleneck(8); c.load_state_dict(sd, strict=False); c.update(); c = c.to(torch.bfloat16)               # load, update, cast
    for m in (a, b, c):
        assert m.quantiles.dtype == torch.bfloat16
        assert torch.equal(m._quantized_cdf, cdf) and torch.equal(m._master("quantiles")[:, 0, 1], med)
    assert not torch.equal(a.quantiles[:, 0, 1].float(), med)                                   # (the live bf16 parameter IS rounded)


def test_values_written_while_bf16_survive_the_cast_back_to_fp32():
    """the saved fp32 master used to be copied over the live tensor on every cast back to fp32, silently reverting a coder parameter that was
    changed while the module was bf16 (optimizer step, p.data = ..., copy_).  A master is authoritative only while the live tensor is still its rounding."""
    from realcamnet_amd import tcm
    torch.manual_seed(5)
    eb = tcm.EntropyBottleneck(4)
    exact = eb.quantiles.detach().clone() + 0.123456789
    with torch.no_grad():
        eb.quantiles.copy_(exact)
    eb = eb.to(torch.bfloat16)
    assert torch.equal(eb._master("quantiles"), exact)                       # untouched: the master (exact fp32) is what the coder reads
    assert torch.equal(eb.float().quantiles, exact)                          # ... and what a cast back restores
    eb = eb.to(torch.bfloat16)
    with torch.no_grad():
        eb.quantiles.add_(1.0)                                               # written while bf16
    live = eb.quantiles.detach().float().clone()
    assert torch.equal(eb._master("quantiles"), live)                        # the coder sees the new values, not the stale master
    assert torch.equal(eb.float().quantiles, live) and not torch.equal(live, exact)


def test_bad_index_is_an_error():
    t, tables = _tables()
    with pytest.raises(Exception):
        bitstream.encode(torch.tensor([0], dtype=torch.int32), torch.tensor([64], dtype=torch.int32), tables, "compressai")


def _tcm(n=32):
    import realcamnet_amd.tcm as T
    m = T.TCM(N=n, M=320, num_slices=5).eval()
    det_fill_(m.state_dict())
    return m


def test_update_tables_equal_oracle_and_checkpoint_loads_strict():
    """TCM.update() (models/tcm.py:430-435): Gaussian tables over get_scale_table() and the bottleneck's factorised-density tables,
    bit-exact against the oracle; a state_dict that carries them (a reference-shaped checkpoint) loads into a fresh model with
    strict=True although its table buffers start empty (models/tcm.py:492-499)."""
    m = _tcm()
    assert m.gaussian_conditional._quantized_cdf.numel() == 0
    assert m.update() is True and m.update() is False and m.update(force=True) is True
    g = E.gc_update(E.get_scale_table())
    gc = m.gaussian_conditional
    assert torch.equal(gc._quantized_cdf, g["_quantized_cdf"]) and torch.equal(gc._cdf_length, g["_cdf_length"]) and torch.equal(gc._offset, g["_offset"])
    assert torch.equal(gc.scale_table, g["scale_table"]) and gc._quantized_cdf.shape == (64, 3133)
    sd = m.state_dict()
    e = E.eb_update({k: v for k, v in sd.items() if k.startswith("entropy_bottleneck.")}, "entropy_bottleneck")
    eb = m.entropy_bottleneck
    assert torch.equal(eb._quantized_cdf, e["_quantized_cdf"]) and torch.equal(eb._cdf_length, e["_cdf_length"]) and torch.equal(eb._offset, e["_offset"])
    for k in ("gaussian_conditional._quantized_cdf", "gaussian_conditional._offset", "gaussian_conditional._cdf_length", "gaussian_conditional.scale_table",
              "gaussian_conditional.scale_bound", "gaussian_conditional.lower_bound_scale.bound", "gaussian_conditional.likelihood_lower_bound.bound",
              "entropy_bottleneck._quantized_cdf", "entropy_bottleneck._offset", "entropy_bottleneck._cdf_length", "entropy_bottleneck.target",
              "entropy_bottleneck.likelihood_lower_bound.bound"):
        assert k in sd, k
    import realcamnet_amd.tcm as T
    fresh = T.TCM(N=32, M=320, num_slices=5).eval()
    res = fresh.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(fresh.gaussian_conditional._quantized_cdf, gc._quantized_cdf) and torch.equal(fresh.entropy_bottleneck._cdf_length, eb._cdf_length)
    import realcamnet_amd.raw2bit as RB
    r = RB.raw_compression_tcm_final(N=32).eval()
    r.update()
    r2 = RB.raw_compression_tcm_final(N=32).eval()
    assert not r2.load_state_dict(r.state_dict(), strict=True).missing_keys


def test_compress_without_tables_raises():
    m = _tcm()
    with pytest.raises(RuntimeError, match="update"):
        from realcamnet_amd.tcm import _coder_tables
        _coder_tables(m.gaussian_conditional)


# ---------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_symbol_kernels_equal_oracle(hip):
    t, _ = _tables()
    table = t["scale_table"].cuda()
    g = torch.Generator().manual_seed(5)
    for dt in (torch.float32, torch.bfloat16):
        y = (torch.randn(2, 6, 7, 64, generator=g) * 6).to(dt)
        mu = torch.randn(2, 6, 7, 64, generator=g).to(dt)
        scale = (torch.randn(2, 6, 7, 64, generator=g).abs() * 4 - 0.3).to(dt)                  # includes values under the 0.11 bound
        scale[0, 0, 0, :4] = torch.tensor([0.11, 256.0, 300.0, 0.1100001]).to(dt)
        sym, idx, y_hat = torch.ops.realcam.gc_symbols(y.cuda(), mu.cuda(), scale.cuda(), table, 0.11)
        nchw = lambda a: a.float().permute(0, 3, 1, 2)
        want_sym = E.quantize_symbols(nchw(y), nchw(mu)).reshape(2, 64, 42)
        want_idx = E.gc_build_indexes(nchw(scale), t["scale_table"]).reshape(2, 64, 42)
        assert torch.equal(sym.cpu(), want_sym) and torch.equal(idx.cpu(), want_idx)
        assert torch.equal(y_hat.cpu(), (want_sym.reshape(2, 64, 6, 7).permute(0, 2, 3, 1).float() + mu.float()).to(dt))
        _, idx2, _ = torch.ops.realcam.gc_symbols(None, None, scale.cuda(), table, 0.11)        # decoder side: indexes only
        assert torch.equal(idx2, idx)
        assert torch.equal(torch.ops.realcam.gc_dequantize(sym, mu.cuda()), y_hat)
        z = (torch.randn(2, 3, 5, 24, generator=g) * 5).to(dt)
        med = torch.randn(24, generator=g)
        s2, i2, z_hat = torch.ops.realcam.eb_symbols(z.cuda(), None, med.cuda(), 2, 3, 5, dt)
        want = torch.round(nchw(z) - med.view(1, -1, 1, 1)).int()
        assert torch.equal(s2.cpu(), want.reshape(2, 24, 15)) and torch.equal(i2.cpu()[0, :, 0], torch.arange(24, dtype=torch.int32))
        assert torch.equal(torch.ops.realcam.eb_symbols(None, s2, med.cuda(), 2, 3, 5, dt)[2], z_hat)


@pytest.mark.gpu
def test_chunked_gpu_coder_is_the_oracle_per_chunk_and_round_trips(hip):
    """Every chunk of the GPU coder's output is byte for byte the oracle's stream of that chunk's symbols; decode inverts it, escapes
    included; chunk = n reproduces the single CompressAI-layout stream."""
    import struct
    t, _ = _tables()
    tables = bitstream.Tables(t["_quantized_cdf"], t["_cdf_length"], t["_offset"], "cuda")
    for n, chunk, seed in ((1, 2048, 1), (4097, 512, 2), (70001, 2048, 3), (3000, 3000, 4)):
        sym, idx = _symbols(n, t, seed)
        s_d, i_d = torch.from_numpy(sym).cuda(), torch.from_numpy(idx).cuda()
        blob = bitstream.encode(s_d, i_d, tables, "chunked", chunk)
        magic, n_sym, ch, n_chunks = struct.unpack_from("<4sIII", blob, 0)
        assert magic == b"RCR1" and n_sym == n and ch == chunk and n_chunks == -(-n // chunk)
        sizes = np.frombuffer(blob, dtype="<u4", count=n_chunks, offset=16)
        p = 16 + 4 * n_chunks
        for c in range(n_chunks):
            want = E.encode_with_indexes(sym[c * chunk:(c + 1) * chunk], idx[c * chunk:(c + 1) * chunk], t)
            assert blob[p:p + sizes[c]] == want, (n, chunk, c)
            p += sizes[c]
        assert p == len(blob)
        got = bitstream.Decoder(blob, tables, "cuda", "chunked").decode(i_d)
        assert torch.equal(got, s_d)
    sym, idx = _symbols(3000, t, 4)
    blob = bitstream.encode(torch.from_numpy(sym).cuda(), torch.from_numpy(idx).cuda(), tables, "chunked", 3000)
    assert blob[20:] == bitstream.encode(torch.from_numpy(sym), torch.from_numpy(idx), tables, "compressai")


def _psnr(a, b):
    import liteisp_oracle as O
    return O.psnr(a.float().cpu(), b.float().cpu())


@pytest.mark.gpu
def test_chunked_decoder_flags_corrupt_payload_instead_of_overreading(hip):
    """A payload whose words were zeroed makes the decoder renormalise on every symbol: it must stop at the chunk's end with an error."""
    t, _ = _tables()
    tables = bitstream.Tables(t["_quantized_cdf"], t["_cdf_length"], t["_offset"], "cuda")
    sym, idx = _symbols(10000, t, 5)
    d_idx = torch.from_numpy(idx).cuda()
    stream = bytearray(bitstream.encode(torch.from_numpy(sym).cuda(), d_idx, tables, "chunked"))
    hdr = 16 + 4 * 5
    ok = bitstream.Decoder(bytes(stream), tables, "cuda", "chunked").decode(d_idx)
    assert np.array_equal(ok.cpu().numpy(), sym)
    stream[hdr:] = bytes(len(stream) - hdr)                          # all-zero payload: state 0 -> a word per symbol
    with pytest.raises(_lib.HipError, match="truncated or corrupt"):
        bitstream.Decoder(bytes(stream), tables, "cuda", "chunked").decode(d_idx)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fmt", ["chunked", "compressai"])
def test_tcm_compress_decompress_round_trip(hip, dt, fmt):
    """decompress(compress(x)) reproduces the encoder's reconstruction: the decoder re-derives every slice's mean / scale from the
    slices it has decoded (deterministic kernels), so the symbols it reads are the symbols that were written; x_hat equals forward()'s
    x_hat clamped to [0, 1] up to forward's ste_round arithmetic (round(v) - v + v vs round(v))."""
    m = _tcm().to("cuda", dt)
    m.update()
    x = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(8)).to("cuda", dt)
    with torch.no_grad():
        enc = m.compress(x, fmt)
        assert len(enc["strings"][0]) == 2 and len(enc["strings"][1]) == 2 and tuple(enc["shape"]) == (4, 4)
        out = m.decompress(enc["strings"], enc["shape"], fmt)["x_hat"]
        again = m.decompress(m.compress(x, fmt)["strings"], enc["shape"], fmt)["x_hat"]
        fwd = m(x)
    assert out.shape == (2, 3, 256, 256) and torch.equal(out, again)
    assert _psnr(out, fwd["x_hat"].clamp(0, 1)) >= (80.0 if dt == torch.float32 else 45.0)
    bits = 8 * sum(len(s) for lst in enc["strings"] for s in lst)
    est = float(-torch.log2(fwd["likelihoods"]["y"].double()).sum() - torch.log2(fwd["likelihoods"]["z"].double()).sum())
    # the stream costs about what forward()'s likelihoods estimate (measured 0.87 - 0.88 of it on this key-filled random net: the
    # estimate clamps tail likelihoods at 1e-9 = 30 bits, the coder's escapes are cheaper; a trained net sits closer to 1)
    assert 0.75 * est <= bits <= 1.15 * est, (bits, est)
    one = m.compress(x[1:2], fmt)                                          # frames are independent: same strings alone as in the batch
    assert one["strings"][0][0] == enc["strings"][0][1] and one["strings"][1][0] == enc["strings"][1][1]


@pytest.mark.gpu
def test_strings_do_not_depend_on_the_order_of_cast_and_update(hip):
    """update() -> .to(bf16) and .to(bf16) -> update() compress to identical strings, and each decodes the other's."""
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(8)).to("cuda", torch.bfloat16)
    a = _tcm(); a.update(); a = a.to("cuda", torch.bfloat16)
    b = _tcm().to("cuda", torch.bfloat16); b.update()
    with torch.no_grad():
        ea, eb_ = a.compress(x), b.compress(x)
        assert ea["strings"] == eb_["strings"]
        assert torch.equal(a.decompress(eb_["strings"], eb_["shape"])["x_hat"], b.decompress(ea["strings"], ea["shape"])["x_hat"])


@pytest.mark.gpu
def test_raw_codec_compress_decompress_round_trip(hip):
    import liteisp_oracle as O
    import realcamnet_amd.raw2bit as RB
    m = RB.raw_compression_tcm_final(N=32).eval()
    det_fill_(m.state_dict())
    m = m.to("cuda", torch.bfloat16)
    m.update()
    g = torch.Generator().manual_seed(12)
    x = [torch.rand(1, 4, 256, 256, generator=g).cuda(), torch.rand(1, 4, 64, 64, generator=g).cuda(), O.make_coord(1, 256, 256).cuda()]
    with torch.no_grad():
        enc = m.compress(x)
        out = m.decompress(enc["strings"], enc["shape"])["x_hat"]
        fwd = m(x)
    assert out.shape == (1, 3, 512, 512) and _psnr(out, fwd["x_hat"].clamp(0, 1)) >= 45.0


@pytest.mark.gpu
def test_chunk_length_is_a_container_field_of_the_codec(hip):
    """compress(x, chunk=...) writes the chunk length into every container's header; decompress() takes no argument and reproduces the SAME x_hat bit for
    bit for every chunk length (the symbols are the same, only their grouping into independent rANS streams changes); shorter chunks cost bytes (a 64-bit
    state flush + a 4-byte size each).  A header whose chunk field was altered no longer matches its size table and is rejected, never decoded."""
    import liteisp_oracle as O
    import realcamnet_amd.raw2bit as RB
    m = RB.raw_compression_tcm_final(N=32).eval()
    det_fill_(m.state_dict())
    m = m.to("cuda", torch.bfloat16)
    m.update()
    g = torch.Generator().manual_seed(12)
    x = [torch.rand(1, 4, 256, 256, generator=g).cuda(), torch.rand(1, 4, 64, 64, generator=g).cuda(), O.make_coord(1, 256, 256).cuda()]
    outs, sizes = {}, {}
    with torch.no_grad():
        for chunk in (2048, 512, 100, 1 << 20):
            enc = m.compress(x, chunk=chunk)
            y0 = enc["strings"][0][0]
            assert struct.unpack_from("<4sIII", y0, 0)[2] == chunk
            outs[chunk] = m.decompress(enc["strings"], enc["shape"])["x_hat"]
            sizes[chunk] = sum(len(s) for grp in enc["strings"] for s in grp)
        for chunk in (512, 100, 1 << 20):
            assert torch.equal(outs[chunk], outs[2048]), chunk
        assert sizes[100] > sizes[512] > sizes[2048] >= sizes[1 << 20]
        enc = m.compress(x, chunk=512)
        bad = bytearray(enc["strings"][0][0]); bad[8:12] = (256).to_bytes(4, "little")          # chunk 512 -> 256: the size table no longer fits
        with pytest.raises(ValueError):
            m.decompress([[bytes(bad)], enc["strings"][1]], enc["shape"])
        with pytest.raises(ValueError):
            m.compress(x, chunk=0)


def test_state_dict_contract_equals_the_reference_classes():
    """tests/golden/codec_state_keys.npz: keys, shapes and dtypes of the reference's TCM / raw_compression_tcm_final after update()
    (its own __init__ / update / load_state_dict over restated CompressAI entropy models) -- the mirror must have exactly those tensors,
    and its update() must build the same tables (SHA-256 of the int32 arrays) from the same key-filled parameters."""
    import hashlib
    import realcamnet_amd.raw2bit as RB
    import realcamnet_amd.tcm as T
    from conftest import load_golden
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "codec_state_keys.npz"))
    sha = lambda t: hashlib.sha256(np.ascontiguousarray(t.cpu().numpy()).tobytes()).hexdigest()
    for pre, build in (("tcm", lambda: T.TCM(N=32, M=320, num_slices=5)), ("raw", lambda: RB.raw_compression_tcm_final(N=32, M=320, num_slices=5))):
        m = build().eval()
        det_fill_(m.state_dict())
        m.update()
        sd = m.state_dict()
        want = {k: (s, d) for k, s, d in zip(g[pre + ".keys"].tolist(), g[pre + ".shapes"].tolist(), g[pre + ".dtypes"].tolist())}
        assert set(sd) == set(want), (sorted(set(sd) ^ set(want))[:10])
        for k, v in sd.items():
            assert "x".join(map(str, v.shape)) == want[k][0] and str(v.dtype) == want[k][1], k
        assert sha(sd["gaussian_conditional._quantized_cdf"]) == str(g[pre + ".sha_gc_cdf"])
        assert sha(sd["entropy_bottleneck._quantized_cdf"]) == str(g[pre + ".sha_eb_cdf"])
        assert sha(sd["gaussian_conditional._offset"]) == str(g[pre + ".sha_gc_offset"])
        assert sha(sd["entropy_bottleneck._cdf_length"]) == str(g[pre + ".sha_eb_length"])


@pytest.mark.gpu
def test_compress_as_a_hip_graph_gives_the_same_strings(hip):
    """compress(graph=True) replays the analysis transform + slice loop + chunk coder as ONE HIP graph captured per input shape; the strings are
    byte-identical to the eager path's (same kernels, same order), a second input goes through the same capture, decompress() of them equals forward()'s x_hat up to
    the clamp, and update() drops the capture (it holds the old tables' addresses)."""
    import realcamnet_amd as M
    from realcamnet_amd import ops
    torch.manual_seed(0)
    net = M.raw2bit.raw_compression_tcm_final().eval().to("cuda", torch.bfloat16)
    net.update()
    g = torch.Generator().manual_seed(3)
    for rep in range(2):
        raw = torch.rand(1, 4, 256, 384, generator=g).to("cuda", torch.bfloat16)
        cond = torch.rand(1, 4, 64, 64, generator=g).to("cuda", torch.bfloat16)
        coord = ops.make_coord(1, 256, 384, device="cuda", dtype=torch.bfloat16)
        with torch.no_grad():
            eager = net.compress([raw, cond, coord])
            graphed = net.compress([raw, cond, coord], graph=True)
            assert graphed["strings"] == eager["strings"] and graphed["shape"] == eager["shape"]
            x_hat = net.decompress(graphed["strings"], graphed["shape"])["x_hat"]
            ref = net([raw, cond, coord])["x_hat"].clamp(0, 1)
        assert torch.equal(x_hat, ref)
    assert len(net.__dict__["_graphs"]) == 1
    net.update(force=True)
    assert "_graphs" not in net.__dict__
