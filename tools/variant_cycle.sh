timeout 600 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -2
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_raw -o trace -- python tools/tcm_bench.py --model raw --frames 2 --steps 2 --warmup 1 > gpurun_out/raw_trace.json 2>gpurun_out/raw_trace.err; tail -1 gpurun_out/raw_trace.json
