mkdir -p gpurun_out/n1
timeout 900 python -m pytest tests/test_winograd.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4 | tee gpurun_out/n1/pytest.log
python tools/wino_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/n1/probe.log
timeout 600 python bench.py --model LiteISPNet --dtype f32 --frames 1 --height 1080 --width 1920 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/n1/cfg2.log
