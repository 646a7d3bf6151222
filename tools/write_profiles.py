#!/usr/bin/env python3
"""Turn the outputs of tools/final_profiles.sh TAG (under gpurun_out/) into the committed evidence under profiles/.
usage: python tools/write_profiles.py TAG [ROUND]"""
import json, os, shutil, subprocess, sys
tag = sys.argv[1]
RND = sys.argv[2] if len(sys.argv) > 2 else "r03"
OUT = sys.argv[3] if len(sys.argv) > 3 else "profiles"      # the GPU box writes under gpurun_out/ (the only directory that travels back)
os.makedirs(OUT, exist_ok=True)
G = "gpurun_out"
last = lambda p: open(p).read().strip().splitlines()[-1]
shutil.copy(f"{G}/pmc_{tag}.json", f"{OUT}/{RND}_pmc_bench.json")
trace, default = last(f"{G}/bench_trace_{tag}.json"), last(f"{G}/bench_default_{tag}.json")
summ = subprocess.run([sys.executable, "tools/rocpd_summary.py", f"{G}/prof_{tag}/trace_results.db", "--last-forwards", "4"], capture_output=True, text=True).stdout
open(f"{OUT}/{RND}_bench_kernel_stats.md", "w").write(f"""# {RND} — kernel trace of `python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-codec-leg` (cfg3, 1x MI355X)

Command: `rocprofv3 --kernel-trace --stats -d gpurun_out/prof_{tag} -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline`
(5 forwards in the trace: 1 warm-up + 3 timed + 1 HIP-event profiling pass; the table counts the LAST FOUR -- steady state, the first forward's weight
packing is out). Summarised from the rocpd database (kept: `gpurun_out/keep_{tag}/bench_trace_results.db`) with `tools/rocpd_summary.py --last-forwards 4`. The conv kernels (`conv_mfma_*`) sum to the `kernel_ms_per_step` that `bench.py` measures live with
HIP events on the launch stream.

```
{trace}
```

{summ}
""")
open(f"{OUT}/{RND}_bench_default.md", "w").write(f"""# {RND} — default `python bench.py` (steps 5, warm-up 2, CPU baseline leg on), 1x MI355X

```json
{default}
```

Other rows of BASELINE.md section 4, same box, same build (`tools/final_profiles.sh`):

```
no attention block (LiteISPNet_GFM_LSC, 4K, B=8, bf16):
{last(f'{G}/bench_nogma_{tag}.json')}
cfg2 (LiteISPNet, 1080p, B=1, fp32):
{last(f'{G}/bench_cfg2_{tag}.json')}
ISPUNet_GFM_LSC (row a13; 4K, B=8, bf16):
{last(f'{G}/bench_ispunet_{tag}.json')}
driver launch form, `python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 ... bench.py --gpus 1` (RCCL, world size 1):
{last(f'{G}/bench_torchrun_{tag}.json')}
```

Matrix-pipe calibration (`tools/mfma_peak.py`, `rc_debug_mfma_peak`: nothing but independent `v_mfma_f32_16x16x32_bf16`):

```
{open(f'{G}/mfma_peak_{tag}.txt').read().strip()}
```
""")
d = json.load(open(f"{OUT}/{RND}_pmc_bench.json"))
dj = json.loads(default)
rows = sorted(d["kernels"].items(), key=lambda kv: -(kv[1]["fetch_bytes_per_dispatch"] + kv[1]["write_bytes_per_dispatch"]) * kv[1]["dispatches"])
tb = "\n".join(f"| `{k[:90]}` | {e['dispatches']} | {e['fetch_bytes_per_dispatch'] / 1e9:.3f} | {e['write_bytes_per_dispatch'] / 1e9:.3f} | "
               f"{(e['fetch_bytes_per_dispatch'] + e['write_bytes_per_dispatch']) * e['dispatches'] / 3 / 1e9:.1f} |" for k, e in rows[:24])
tot = (d["all_kernels_total_bytes"]["fetch"] + d["all_kernels_total_bytes"]["write"]) / 3 / 1e9
rate = tot / dj["ms_per_step"]
open(f"{OUT}/{RND}_pmc_bench.md", "w").write(f"""# {RND} — HBM traffic of the bench command from PMC counters (cfg3, 1x MI355X)

`tools/pmc_bench.sh`: two passes of `rocprofv3 --pmc <counter> --kernel-trace -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-codec-leg`
(FETCH_SIZE and WRITE_SIZE do not fit one pass; no other trace domains), 3 forwards per pass, summarised per kernel by
`tools/pmc_bench_summary.py` into `profiles/{RND}_pmc_bench.json` (which `bench.py` reads for `roofline.traffic` and
`hbm_whole_step`). Units: FETCH_SIZE KiB x 1024 x 2 (gfx950: a 16-byte-per-lane streaming read is tallied at half, MI355X guide
section HBM), WRITE_SIZE KiB x 1024 (calibrated earlier on the 48->48 layer: equals the output bytes exactly).

**Whole step: {tot:.0f} GB of HBM traffic per forward** ({d['all_kernels_total_bytes']['fetch'] / 3 / 1e9:.0f} GB read + {d['all_kernels_total_bytes']['write'] / 3 / 1e9:.0f} GB written) = {rate:.2f} TB/s at {dj['ms_per_step']:.1f} ms per step:
{rate / 5.9 * 100:.0f} % of this part's measured copy rate (5.9 TB/s, 1.6 GB -> 1.6 GB one-shot float4 copy, `tools/hbm_probe.py`), {rate / 8 * 100:.0f} % of the 8 TB/s spec.
Conv kernels: {d['conv_kernels_all']['dispatches']} dispatches, {d['conv_kernels_all']['hbm_bytes_per_dispatch'] / 1e9:.2f} GB per dispatch on average.

| kernel | dispatches (3 forwards) | fetch GB / dispatch | write GB / dispatch | GB per forward |
|---|---|---|---|---|
{tb}
""")
codec = last(f"{G}/bench_codec_{tag}.json")
csumm = subprocess.run([sys.executable, "tools/rocpd_summary.py", f"{G}/prof_codec_{tag}/trace_results.db", "--last-forwards", "2"], capture_output=True, text=True).stdout
open(f"{OUT}/{RND}_rawcodec_kernel_stats.md", "w").write(f"""# {RND} — RAW codec leg: `python bench.py --model raw_compression_tcm_final --frames 8` (cfg5 shape on one GPU, bf16, 1x MI355X)

Default run (steps 5, warm-up 2, CPU baseline leg on):

```json
{codec}
```

Kernel trace: `rocprofv3 --kernel-trace --stats -- python bench.py --model raw_compression_tcm_final --frames 8 --steps 2 --warmup 1 --no-cpu-baseline`
(3 forwards of 8 frames in the trace; the table counts the LAST TWO -- steady state, weight packing and its copies are out), summarised with
`tools/rocpd_summary.py --last-forwards 2`:

{csumm}

Bitstream legs (`tools/codec_stream_bench.py`: compress / decompress of one 4K mosaic's packed RAW, both stream formats; the forward's time beside them):

```
{open(f'{G}/codec_stream_{tag}.txt').read().strip()}
```
""")
open(f"{OUT}/{RND}_gma_stages.md", "w").write(f"""# {RND} — the GroupMix block launch by launch at the cfg3 size (8 x 544 x 960 tokens, dim 80, bf16), `tools/gma_stage_bench.py`, HIP events

```
{open(f'{G}/gma_stages_{tag}.txt').read().strip()}
```
""")
def rd(name):
    p = f"{G}/{name}_{tag}.txt"
    return open(p).read().replace("/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", "").strip() if os.path.exists(p) else "(not collected)"


open(f"{OUT}/{RND}_tail_fold_rcag.md", "w").write(f"""# {RND} — the folded tail against the two launches it replaces, and the early-gate RCAGroup (`tools/tail_fold_probe.py`, `tools/rcag_probe.py`), 1x MI355X

```
{rd('tail_fold')}
```

`RCAGroup shipped` = the early-gate schedule (default); `proxy kernels` = the same schedule timed with round 3's kernels before the new epilogues existed
(conv + sums, conv + residual) -- the estimate the work was started on.  Round 3's schedule (gate folded into the next conv's staging) measured 9.16 / 2.36 ms
at the two sizes on the same tool.
""")
open(f"{OUT}/{RND}_power_and_autonomous_kernels.md", "w").write(f"""# {RND} — power / clock per kernel variant and the wave-autonomous kernels at the final sources (`tools/power_probe.py`, `tools/auto_probe.py 48`, `tools/auto_probe.py 64`), 1x MI355X

`rocm-smi` sampled in the middle of a ~1.2 s back-to-back run of each variant (ms per launch | shader clock | socket power).  The analysis is in
`profiles/r05_power_wall.md`; this is the same probe on the round's final build.

```
{rd('power_probe')}
```

Kernel 6 (48 channels; `persist_auto` 2 = every eligible form, the default 1 keeps the residual forms on kernel 2) and kernel 7 (64 channels) against the kernels
they replace, bit-equality of every operand form first:

```
{rd('auto_probe')}
```

`LiteISPNet`, 4K, 8 frames, bf16 (the 64-channel trunk; round 4: 58.6 ms):
{last(f'{G}/bench_liteisp_bf16_{tag}.json') if os.path.exists(f'{G}/bench_liteisp_bf16_{tag}.json') else '(not collected)'}
""")
open(f"{OUT}/{RND}_thin_stage_kernel.md", "w").write(f"""# {RND} — kernel 4b (thin stages of the multi-chunk conv kernel, DESIGN 4.12) against kernel 4 at the final sources, 1x MI355X

In order: the codec bench line (`bench.py --model raw_compression_tcm_final --frames 8 --steps 5 --warmup 2`) with `RC_DEBUG=thin=2 / 1 / 0` twice (2 = default: kernel 4b
in both of its forms; 1 = only the LDS-DMA form for thin stages; 0 = kernel 4 everywhere; one process each, same box); the cfg3 headline (`bench.py --steps 10 --warmup 3`)
the same way; cfg3's multi-chunk layers one by one (`tools/wsm_probe.py`) under the three settings; the per-launch table of one codec forward at 8 frames
(`tools/codec_conv_breakdown.py 8`), then the rows kernel 4b's DMA form serves re-measured with kernel 4; the back-to-back stress loops (`tools/thin_stress.py`) and the
per-launch A/B inside the small RAW codec's compress / decompress / forward after the TCM round trips (`tools/thin_ab.py --after-tcm`: the order that exposed the stale tile).

```
{rd('thin')}
```
""")
mf = f"{G}/pmc_mfma_{tag}.md"
if os.path.exists(mf):
    open(f"{OUT}/{RND}_pmc_mfma_lds.md", "w").write(f"""# {RND} — matrix-pipe and LDS counters of the bench command (cfg3), `tools/pmc_mfma.sh`

One `rocprofv3 --pmc <counter> --kernel-trace` pass per counter over `python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-codec-leg` (3 forwards; never combined
with other trace domains).  `SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)` = fraction of the CUs' busy time in which a SIMD's matrix pipe is busy.

{open(mf).read().strip()}
""")
cmp_db = f"{G}/prof_cmp_{tag}/trace_results.db"
if os.path.exists(cmp_db):
    cs = subprocess.run([sys.executable, "tools/rocpd_summary.py", cmp_db], capture_output=True, text=True).stdout
    ans = "\n".join(l for l in cs.splitlines() if "ans::" in l or l.startswith("| kernel") or l.startswith("|---"))
    open(f"{OUT}/{RND}_codec_stream.md", "w").write(f"""# {RND} — entropy-coding kernels inside compress() / decompress() of one 4K frame (`tools/compress_trace.py` under rocprofv3 --kernel-trace)

```
{rd('compress')}
```

{ans}

Round 2's one-kernel encoder (idx -> sizes/offsets -> cdf as three dependent global loads per symbol at an 8 KB lane stride, then a 64-bit
division): `rc::ans::encode_chunks_kernel` 1 547 us per call, 6 calls per compress = 9.3 of its 30 ms.  Round 3: `prepare_kernel` (all symbols in
parallel: CDF row, escape, start / freq, reciprocal of freq; operations written transposed) + `encode_serial_kernel` (one lane per chunk, 16 symbols'
operations loaded per batch, division-free state update bit-identical to the dividing one).
""")
c2, ng, iu = (json.loads(last(f"{G}/bench_{n}_{tag}.json")) for n in ("cfg2", "nogma", "ispunet"))
print(json.dumps({"codec_leg": dj.get("codec_leg"), "cfg3": [dj["value"], dj["ms_per_step"], dj["roofline"]["achieved"], dj["roofline"]["frac"], dj.get("hbm_whole_step"), dj["cpu_baseline"]["value"], dj["psnr_db_vs_cpu_fp32"]],
                  "cfg2": [c2["value"], c2["ms_per_step"], c2["roofline"]["achieved"], c2["roofline"]["frac"], c2["cpu_baseline"]["value"], c2["psnr_db_vs_cpu_fp32"]],
                  "nogma": [ng["value"], ng["ms_per_step"], ng["roofline"]["achieved"], ng["roofline"]["frac"]],
                  "ispunet": [iu["value"], iu["ms_per_step"], iu["roofline"]["achieved"]]}, indent=1))
