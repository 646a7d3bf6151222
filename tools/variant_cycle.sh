timeout 600 python -m pytest tests -m gpu -q -x --timeout 600 -k "gma or GMA or groupmix or e2e" 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_gp -o trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/rocpd_summary.py gpurun_out/prof_gp/trace_results.db 2>/dev/null | grep "gma_pointwise\|dwconv2d_kernel<bf16,7>"
