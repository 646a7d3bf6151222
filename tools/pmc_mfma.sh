#!/bin/bash
# Matrix-pipe and LDS counters of the bench command, one rocprofv3 --pmc pass per counter group (never combined with other trace domains).
# usage: tools/pmc_mfma.sh TAG   -> gpurun_out/pmc_mfma_TAG.md
TAG=${1:-x}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d gpurun_out/pmcm_${TAG}_$C -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-codec-leg > /dev/null 2> gpurun_out/pmcm_${TAG}_$C.err
  tail -1 gpurun_out/pmcm_${TAG}_$C.err
done
python tools/pmc_mfma_summary.py $TAG > gpurun_out/pmc_mfma_$TAG.md
head -30 gpurun_out/pmc_mfma_$TAG.md
