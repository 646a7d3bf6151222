#!/bin/bash
# Kernel trace of the default bench command, summarised per kernel.   usage: tools/quick_trace.sh TAG [bench args...]
TAG=${1:-q}; shift
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/qt_$TAG -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/qt_$TAG.json 2> gpurun_out/qt_$TAG.err
cut -c1-200 gpurun_out/qt_$TAG.json
python tools/rocpd_summary.py gpurun_out/qt_$TAG/trace_results.db > gpurun_out/qt_$TAG.md
head -45 gpurun_out/qt_$TAG.md | cut -c1-150
