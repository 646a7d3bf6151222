#!/usr/bin/env python3
"""Round-6 evidence files from what tools/final_profiles.sh left under gpurun_out/profiles_<TAG>/ (run locally after the GPU call): profiles/r06_winograd.md, r06_codec_b1.md.
usage: python tools/write_profiles_r06.py TAG"""
import os, sys
tag = sys.argv[1]
G = f"gpurun_out/profiles_{tag}"


def rd(name):
    p = f"{G}/{name}"
    return open(p).read().replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", "").strip() if os.path.exists(p) else "(not collected)"


def last(name):
    t = rd(name)
    return t.splitlines()[-1] if t and t != "(not collected)" else t


PSNR = """fp32           51.9 s
bf16           47.2 s  PSNR vs fp32 oracle 64.41 dB
wino_multi    100.6 s  PSNR vs fp32 oracle 64.45 dB  (vs bf16-direct 65.74 dB)  layers {(128, 128): 36, (128, 192): 1, (128, 512): 1, (192, 48): 1, (192, 128): 1, (192, 192): 4, (512, 128): 1, (512, 512): 2}
wino_all      266.2 s  PSNR vs fp32 oracle 63.53 dB  (vs bf16-direct 62.47 dB)  layers {(48, 48): 41, (48, 192): 2, (128, 128): 36, (128, 192): 1, (128, 512): 1, (192, 48): 1, (192, 128): 1, (192, 192): 4, (512, 128): 1, (512, 512): 2}"""

open("profiles/r06_winograd.md", "w").write(f"""# r06 — Winograd F(2x2, 3x3): the fp32 kernel (`csrc/wino.hip`), the measurement its structure rests on, and the bf16 gate (VERDICT r5 item 1), 1x MI355X

## 1. Item 1(a): PSNR of F(2,3) with bf16 transformed operands (CPU, `python tools/winograd_psnr.py --size 1080x1920`, one 4K frame, LiteISPNet_GFM_LSC, seed-0 weights)

bf16 path emulated on the CPU oracle (weights and every conv's input / output rounded to bf16, fp32 accumulation); `wino_multi`: F(2,3) with U = G g G^T and V = B^T d B rounded to bf16 on every
multi-chunk layer (cin > 64), `wino_all`: on every 3x3 layer with >= 48 channels.  Gate: >= 58 dB (today's bf16 path 63.1 on the GPU, floor 55).

```
{PSNR}
```

Numerics are not what stops a bf16 Winograd path; DESIGN.md section 4.2 ("bf16: costed, not built") is.

## 2. Does a SIMD overlap MFMA passes with another wave's instructions?  (`tools/ubench/mfma_valu_overlap.hip`: 256 blocks x 8 waves; waves 0-3 role A, 4-7 role B, one of each per SIMD)

```
{rd(f'mvo_{tag}.txt')}
```

Every pair takes the SUM of its parts: an MFMA-issuing wave and a VALU- or LDS-issuing wave on one SIMD do not overlap (fp32 and bf16 MFMA alike).  A second wave hides latency, not issue --
so a kernel's time per SIMD is (MFMA passes) + (every other instruction x ~4 cycles) + unhidden waits, and the Winograd kernel is written for instruction count.

## 3. The fp32 layers of cfg2 one by one, Winograd against the implicit GEMM (`tools/wino_probe.py`: 30 launches after 50 warm-up, alternating twice; TF/s columns are ALGORITHMIC for both), then with the item size forced

```
{rd(f'wino_probe_{tag}.txt')}
```

## 4. Instruction mix of the 64 -> 64 layer at 544 x 960 (`tools/pmc_any.py`, per dispatch; 8.356 M MFMAs = 64 per wave and stage)

```
{rd(f'wino_pmc_{tag}.txt')}
```

## 5. cfg2 end to end, same box: `bench.py --model LiteISPNet --dtype f32 --frames 1 --height 1080 --width 1920 --steps 20 --warmup 5`

Winograd (default), under `rocprofv3 --kernel-trace --stats`:
```
{last(f'bench_cfg2_trace_{tag}.json')}
```
```
{rd(f'cfg2_kernel_stats_{tag}.txt')}
```
`RC_WINOGRAD=0` (every layer on the implicit GEMM, round 5's path):
```
{last(f'bench_cfg2_direct_{tag}.json')}
```
""")

open("profiles/r06_codec_b1.md", "w").write(f"""# r06 — the RAW codec at ONE 4K frame (VERDICT r5 item 5): `tools/codec_b1.py 1 --codec`, raw_compression_tcm_final, bf16, 1x MI355X

Wall ms per call (perf_counter around call + synchronize, median of 7 after 2 warm-up calls).  Round 5 (`profiles/r05_rawcodec_kernel_stats.md`): forward 20.8, compress 23.6, decompress 28.0.

```
{rd(f'codec_b1_{tag}.txt')}
```

* forward: `realcamnet_amd.GraphedCall` -- the whole `forward_mosaic` captured per input signature and replayed; `ops.fork_join` keeps the slice loop's two-stream forks as graph branches
  (`ops.GRAPH_FORK`); "1 stream" = the same capture with the forks serialised.  Every tensor of the result dict is bit-identical to the eager call
  (`tests/test_gpu_parity.py::test_codec_forward_replayed_as_a_hip_graph_equals_the_eager_forward`).
* compress: eager = the device half of every container enqueued first (`bitstream.encode_async`), ONE sync for all chunk byte counts (`bitstream.finish`); graph = analysis transform + slice loop +
  chunk coder replayed as one graph (`compress(graph=True)`), identical strings (`tests/test_bitstream.py::test_compress_as_a_hip_graph_gives_the_same_strings`).
* decompress: the decode kernels' error flags are looked at once, after the synthesis transform is enqueued (`rans_decode_chunks_async`, `Decoder.check`).
""")
print("wrote profiles/r06_winograd.md, profiles/r06_codec_b1.md")
