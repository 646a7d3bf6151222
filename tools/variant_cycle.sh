timeout 120 python tools/conv_probe.py 2>&1 | grep "^conv"
timeout 120 python tools/conv_probe.py --sums 2>&1 | grep "^conv"
timeout 120 python tools/conv_probe.py --gated 2>&1 | grep "^conv"
timeout 120 python tools/conv_probe.py --cin 192 --cout 192 --h 272 --w 480 2>&1 | grep "^conv"
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -3
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
timeout 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"ms_per_step": [0-9.]*'
