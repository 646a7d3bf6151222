mkdir -p gpurun_out/w10
timeout 1500 python -m pytest tests/test_winograd.py tests/test_gpu_parity.py tests/test_full_size.py -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/w10/pytest.log
timeout 600 python bench.py --model LiteISPNet --dtype f32 --frames 1 --height 1080 --width 1920 --steps 20 --warmup 5 2>&1 | tail -1 | tee gpurun_out/w10/cfg2.log
RC_WINOGRAD=0 timeout 600 python bench.py --model LiteISPNet --dtype f32 --frames 1 --height 1080 --width 1920 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/w10/cfg2_direct.log
