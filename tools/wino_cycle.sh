mkdir -p gpurun_out/n2
for v in 2 1 2 1; do echo "== wino_nnt=$v"; RC_DEBUG="wino_nnt=$v" python tools/wino_probe.py 2>&1 | grep -v amdgpu.ids | grep -v "48 ->"; done | tee gpurun_out/n2/probe.log
