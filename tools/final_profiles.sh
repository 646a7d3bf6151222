#!/bin/bash
# Round-end evidence: PMC traffic passes, kernel-trace of the bench command, default bench line (with CPU baseline),
# the other table rows, MFMA calibration.  usage: tools/final_profiles.sh TAG [ROUND]   (outputs under gpurun_out/)
TAG=${1:-fin}
ROUND=${2:-r06}
mkdir -p gpurun_out gpurun_out/profiles_$TAG
bash tools/pmc_bench.sh $TAG > gpurun_out/pmc_${TAG}.log 2>&1

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp gpurun_out/pmc_${TAG}.json profiles/${ROUND}_pmc_bench.json   # so this run's bench line carries the traffic
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-codec-leg > gpurun_out/bench_trace_$TAG.json 2> gpurun_out/bench_trace_$TAG.err
timeout 900 python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err
timeout 300 python bench.py --model LiteISPNet_GFM_LSC --no-cpu-baseline > gpurun_out/bench_nogma_$TAG.json 2>/dev/null
timeout 300 python bench.py --model LiteISPNet --dtype f32 --frames 1 --height 1080 --width 1920 --steps 20 --warmup 5 > gpurun_out/bench_cfg2_$TAG.json 2>/dev/null
timeout 300 python bench.py --model ISPUNet_GFM_LSC --no-cpu-baseline > gpurun_out/bench_ispunet_$TAG.json 2>/dev/null
timeout 600 python bench.py --model raw_compression_tcm_final --frames 8 > gpurun_out/bench_codec_$TAG.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_codec_$TAG -o trace -- python bench.py --model raw_compression_tcm_final --frames 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_codec_trace_$TAG.json 2> gpurun_out/bench_codec_trace_$TAG.err
timeout 300 python tools/gma_stage_bench.py > gpurun_out/gma_stages_$TAG.txt 2>&1
timeout 300 python tools/codec_stream_bench.py > gpurun_out/codec_stream_$TAG.txt 2>&1
timeout 120 python tools/mfma_peak.py > gpurun_out/mfma_peak_$TAG.txt 2>/dev/null
# round 5 evidence: power / clock per kernel variant (the conv kernels sit at the board's power cap), kernels 6 / 7 against the kernels they replace
timeout 200 python tools/power_probe.py > gpurun_out/power_probe_$TAG.txt 2>&1
( timeout 150 python tools/auto_probe.py 48; timeout 150 python tools/auto_probe.py 64 ) > gpurun_out/auto_probe_$TAG.txt 2>&1
timeout 300 python bench.py --model LiteISPNet --no-cpu-baseline > gpurun_out/bench_liteisp_bf16_$TAG.json 2>/dev/null
# round 4 evidence: the folded tail against the two launches it replaces, the early-gate RCAGroup, the codec's per-launch breakdown
( timeout 200 python tools/tail_fold_probe.py; timeout 200 python tools/rcag_probe.py ) > gpurun_out/tail_fold_$TAG.txt 2>&1
timeout 200 python tools/codec_conv_breakdown.py > gpurun_out/codec_probes_$TAG.txt 2>&1
# round 5, later: kernel 4b (thin stages) against kernel 4 in one process each, same box; the per-launch table at the bench's 8 frames; cfg3's multi-chunk layers
( for t in 2 1 0 2 1 0; do echo "codec thin=$t"; RC_DEBUG=thin=$t timeout 200 python bench.py --model raw_compression_tcm_final --frames 8 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-260; done;
  for t in 2 1 0 2 1 0; do echo "cfg3 thin=$t"; RC_DEBUG=thin=$t timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-codec-leg 2>/dev/null | cut -c1-200; done;
  for t in 2 1 0; do echo "multi-chunk layers of cfg3, thin=$t"; RC_DEBUG=thin=$t timeout 200 python tools/wsm_probe.py; done;
  TOP=60 timeout 200 python tools/codec_conv_breakdown.py 8; RC_DEBUG=thin=0 TOP=60 timeout 200 python tools/codec_conv_breakdown.py 8 | grep -i "fold2\|cout=12 \|launches";
  timeout 200 python tools/thin_stress.py; timeout 200 python tools/thin_stress.py 192 192 3 2 200 300; timeout 300 python tools/thin_ab.py --after-tcm | tail -8 ) > gpurun_out/thin_$TAG.txt 2>&1
bash tools/pmc_mfma.sh $TAG > gpurun_out/pmc_mfma_${TAG}.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cmp_$TAG -o trace -- python tools/compress_trace.py > gpurun_out/compress_$TAG.txt 2>&1
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun_$TAG.json 2>/dev/null

# round 6: the fp32 Winograd kernel (cfg2) -- kernel trace of the cfg2 command, layer-by-layer A/B against the implicit GEMM, both item sizes, the MFMA / VALU overlap
# microbenchmark its structure rests on; the codec at one frame (eager / HIP-graph replay / compress / decompress)
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg2_$TAG -o trace -- python bench.py --model LiteISPNet --dtype f32 --frames 1 --height 1080 --width 1920 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_cfg2_trace_$TAG.json 2> gpurun_out/bench_cfg2_trace_$TAG.err
python tools/rocpd_summary.py gpurun_out/prof_cfg2_$TAG/trace_results.db > gpurun_out/cfg2_kernel_stats_$TAG.txt 2>&1
RC_WINOGRAD=0 timeout 300 python bench.py --model LiteISPNet --dtype f32 --frames 1 --height 1080 --width 1920 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_cfg2_direct_$TAG.json 2>/dev/null
( timeout 300 python tools/wino_probe.py; for v in 2 1; do echo "== RC_DEBUG=wino_nnt=$v"; RC_DEBUG=wino_nnt=$v timeout 300 python tools/wino_probe.py; done ) > gpurun_out/wino_probe_$TAG.txt 2>&1
( cd tools/ubench && hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o /tmp/mvo 2>/dev/null && /tmp/mvo ) > gpurun_out/mvo_$TAG.txt 2>&1
timeout 300 python tools/pmc_any.py $TAG "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_VALU_MFMA_BUSY_CYCLES;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY;GRBM_GUI_ACTIVE" -- python tools/wino_one.py 64 64 544 960 10 > gpurun_out/wino_pmc_$TAG.txt 2>&1
timeout 400 python tools/codec_b1.py 1 --codec > gpurun_out/codec_b1_$TAG.txt 2>&1
# round 6, second session: fillers in the SAME wave's stream (tools/ubench/mfma_valu_samewave.hip); the GroupMix front in one launch (rc_gma_qkv_aggregate): counters, A/B of the whole bench
( cd tools/ubench && hipcc --offload-arch=gfx950 -O3 mfma_valu_samewave.hip -o /tmp/mvs 2>/dev/null && /tmp/mvs ) > gpurun_out/mvs_$TAG.txt 2>&1
timeout 400 python tools/pmc_any.py ${TAG}qa "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_INSTS_SMEM;SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES;GRBM_GUI_ACTIVE" -- python tools/dbg/qa_pmc_run.py > gpurun_out/qa_pmc_$TAG.txt 2>&1
( for i in 1 2 3; do for v in 0 1; do echo -n "RC_GMA_FRONT=$v: "; RC_GMA_FRONT=$v timeout 200 python bench.py --no-cpu-baseline --no-codec-leg --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'MP/s')"; done; done; timeout 200 python tools/dbg/qa_stress.py 40 ) > gpurun_out/gma_front_ab_$TAG.txt 2>&1
# round 6, third session: conv -> Haar DWT in one launch (RC_OUT_NHWC_DWT) alone and in the whole bench, the front end (lens-shading head against the colour-prior chain), the codec's
# small dependent launches, the 4 -> 128 head with its weights resident
( for f in none dwt conv+dwt none dwt conv+dwt; do timeout 100 python tools/conv_one.py 48 48 1088 1920 8 30 $f 2>/dev/null; done
  for i in 1 2 3; do for v in 0 1; do echo -n "RC_FUSE_DWT=$v: "; RC_FUSE_DWT=$v timeout 200 python bench.py --no-cpu-baseline --no-codec-leg --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms', d['value'], 'MP/s')"; done; done
  timeout 200 python tools/front_probe.py 2>&1 | grep -v amdgpu.ids
  timeout 200 python tools/lsc_bench.py 2>&1 | grep -v amdgpu.ids
  timeout 200 python tools/small_launch_probe.py 8 2>&1 | grep -v amdgpu.ids; timeout 200 python tools/small_launch_probe.py 1 2>&1 | grep -v amdgpu.ids
  for fl in 128 0 128 0; do echo -n "conv_flags=$fl (128: weights re-staged per cout tile) "; RC_DEBUG=conv_flags=$fl timeout 100 python tools/conv_one.py 4 128 1152 1920 8 20 none 2>/dev/null; done ) > gpurun_out/session3_$TAG.txt 2>&1
cp gpurun_out/session3_$TAG.txt gpurun_out/profiles_$TAG/ 2>/dev/null
cp gpurun_out/mvs_$TAG.txt gpurun_out/qa_pmc_$TAG.txt gpurun_out/gma_front_ab_$TAG.txt gpurun_out/gma_stages_$TAG.txt gpurun_out/profiles_$TAG/ 2>/dev/null
cp gpurun_out/cfg2_kernel_stats_$TAG.txt gpurun_out/wino_probe_$TAG.txt gpurun_out/mvo_$TAG.txt gpurun_out/wino_pmc_$TAG.txt gpurun_out/codec_b1_$TAG.txt gpurun_out/bench_cfg2_direct_$TAG.json gpurun_out/bench_cfg2_trace_$TAG.json gpurun_out/profiles_$TAG/ 2>/dev/null

tail -c 400 gpurun_out/bench_default_$TAG.json; echo; cut -c1-160 gpurun_out/bench_nogma_$TAG.json gpurun_out/bench_cfg2_$TAG.json gpurun_out/bench_ispunet_$TAG.json gpurun_out/bench_codec_$TAG.json gpurun_out/bench_torchrun_$TAG.json

# summaries are written ON the box (gpurun_out/ is capped at 64 MiB on the way back): keep them, the bench trace database and the small logs
cp gpurun_out/tail_fold_$TAG.txt gpurun_out/codec_probes_$TAG.txt gpurun_out/power_probe_$TAG.txt gpurun_out/auto_probe_$TAG.txt gpurun_out/thin_$TAG.txt gpurun_out/profiles_$TAG/ 2>/dev/null
python tools/write_profiles.py $TAG $ROUND gpurun_out/profiles_$TAG > gpurun_out/profiles_$TAG/summary.json 2> gpurun_out/write_profiles_$TAG.err
mkdir -p gpurun_out/keep_$TAG && cp gpurun_out/prof_$TAG/trace_results.db gpurun_out/keep_$TAG/bench_trace_results.db 2>/dev/null
cp gpurun_out/prof_codec_$TAG/trace_results.db gpurun_out/keep_$TAG/codec_trace_results.db 2>/dev/null
rm -rf gpurun_out/prof_* gpurun_out/pmc_${TAG}_* gpurun_out/pmcm_${TAG}_* gpurun_out/pmca_${TAG}_*
du -sh gpurun_out
