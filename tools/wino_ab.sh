# A/B of library variants on one box: tools/wino_ab.sh <lib> <lib> ...  ("-" = the in-tree library); alternates twice
mkdir -p gpurun_out/ab
for rep in 1 2; do
for L in "$@"; do
  if [ "$L" = "-" ]; then unset RC_HIP_LIB; else export RC_HIP_LIB=$PWD/$L; fi
  echo "== $L"; python tools/wino_probe.py 2>&1 | grep -v amdgpu.ids | head -${LINES_:-3}
done; done | tee gpurun_out/ab/ab.log
