#!/bin/bash
# Steady-state kernel trace of `bench.py --model $1` (bf16, 4K, 8 frames).  usage: tools/qt_model.sh MODEL
M=${1:-LiteISPNet}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/qt_$M -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --model $M > gpurun_out/qt_$M.json 2> gpurun_out/qt_$M.err
python tools/rocpd_summary.py gpurun_out/qt_$M/trace_results.db --last-forwards 4 > gpurun_out/qt_$M.md
rm -rf gpurun_out/qt_$M
cut -c1-200 gpurun_out/qt_$M.json; head -22 gpurun_out/qt_$M.md | cut -c1-150
