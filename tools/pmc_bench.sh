#!/bin/bash
# HBM traffic of the bench command from PMC counters: one rocprofv3 pass per counter (FETCH_SIZE and WRITE_SIZE do not
# fit one pass; --pmc is never combined with other trace domains).  usage: tools/pmc_bench.sh TAG
TAG=${1:-x}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d gpurun_out/pmc_${TAG}_$C -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-codec-leg > gpurun_out/pmc_${TAG}_$C.json 2> gpurun_out/pmc_${TAG}_$C.err
  tail -1 gpurun_out/pmc_${TAG}_$C.err
done
python tools/pmc_bench_summary.py gpurun_out/pmc_${TAG}_FETCH_SIZE/pmc_results.db gpurun_out/pmc_${TAG}_WRITE_SIZE/pmc_results.db > gpurun_out/pmc_${TAG}.json
cat gpurun_out/pmc_${TAG}.json | head -c 1500
