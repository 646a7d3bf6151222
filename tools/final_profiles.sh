#!/bin/bash
# Round-end evidence: PMC traffic passes, kernel-trace of the bench command, default bench line (with CPU baseline),
# the other table rows, MFMA calibration.  usage: tools/final_profiles.sh TAG [ROUND]   (outputs under gpurun_out/)
TAG=${1:-fin}
ROUND=${2:-r02}
mkdir -p gpurun_out
bash tools/pmc_bench.sh $TAG > gpurun_out/pmc_${TAG}.log 2>&1

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp gpurun_out/pmc_${TAG}.json profiles/${ROUND}_pmc_bench.json   # so this run's bench line carries the traffic
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_trace_$TAG.json 2> gpurun_out/bench_trace_$TAG.err
timeout 900 python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err
timeout 300 python bench.py --model LiteISPNet_GFM_LSC --no-cpu-baseline > gpurun_out/bench_nogma_$TAG.json 2>/dev/null
timeout 300 python bench.py --model LiteISPNet --dtype f32 --frames 1 --height 1080 --width 1920 --steps 20 --warmup 5 > gpurun_out/bench_cfg2_$TAG.json 2>/dev/null
timeout 300 python bench.py --model ISPUNet_GFM_LSC --no-cpu-baseline > gpurun_out/bench_ispunet_$TAG.json 2>/dev/null
timeout 600 python bench.py --model raw_compression_tcm_final --frames 4 > gpurun_out/bench_codec_$TAG.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_codec_$TAG -o trace -- python bench.py --model raw_compression_tcm_final --frames 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_codec_trace_$TAG.json 2> gpurun_out/bench_codec_trace_$TAG.err
timeout 300 python tools/gma_stage_bench.py > gpurun_out/gma_stages_$TAG.txt 2>&1
timeout 300 python tools/codec_stream_bench.py > gpurun_out/codec_stream_$TAG.txt 2>&1
timeout 120 python tools/mfma_peak.py > gpurun_out/mfma_peak_$TAG.txt 2>/dev/null
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun_$TAG.json 2>/dev/null

tail -c 400 gpurun_out/bench_default_$TAG.json; echo; cut -c1-160 gpurun_out/bench_nogma_$TAG.json gpurun_out/bench_cfg2_$TAG.json gpurun_out/bench_ispunet_$TAG.json gpurun_out/bench_codec_$TAG.json gpurun_out/bench_torchrun_$TAG.json
