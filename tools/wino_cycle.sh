mkdir -p gpurun_out/g3
timeout 1200 python -m pytest tests/test_bitstream.py tests/test_gpu_parity.py -x -q -m gpu -k "graph" 2>&1 | grep -v amdgpu | tail -12 | tee gpurun_out/g3/pytest.log
python tools/codec_b1.py 1 --codec 2>&1 | grep -v amdgpu.ids | tee gpurun_out/g3/b1.log
