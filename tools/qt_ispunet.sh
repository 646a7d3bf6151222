cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/qt_isp -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --model ISPUNet_GFM_LSC > gpurun_out/qt_isp.json 2> gpurun_out/qt_isp.err
python tools/rocpd_summary.py gpurun_out/qt_isp/trace_results.db --last-forwards 4 > gpurun_out/qt_isp.md
rm -rf gpurun_out/qt_isp
head -30 gpurun_out/qt_isp.md | cut -c1-150
