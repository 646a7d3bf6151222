#!/bin/bash
# One GPU iteration: parity tests, bench (no CPU baseline), kernel-trace profile.  usage: tools/gpu_cycle.sh TAG [pytest-args]
TAG=${1:-x}; shift
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q -x --timeout 300 "$@" 2>&1 | tail -15 > gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -o trace -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -2 gpurun_out/bench_$TAG.err
python -c "
import json;d=json.load(open('gpurun_out/bench_$TAG.json'));print(d['value'],d['unit'],d['ms_per_step'],'ms/step', d['roofline']['achieved'],'TF/s conv', d['roofline']['kernel_ms_per_step'])"
