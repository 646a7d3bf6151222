cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_codec_q -o trace -- python bench.py --model raw_compression_tcm_final --frames 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/codec_q.json 2> gpurun_out/codec_q.err
python tools/rocpd_summary.py gpurun_out/prof_codec_q/trace_results.db --last-forwards 2 > gpurun_out/codec_q.md
rm -rf gpurun_out/prof_codec_q
head -70 gpurun_out/codec_q.md | cut -c1-160
