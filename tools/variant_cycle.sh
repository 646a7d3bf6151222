timeout 120 python tools/conv_probe.py --act none 2>&1 | grep "^conv"
timeout 120 python tools/conv_probe.py --sums --act none 2>&1 | grep "^conv"
timeout 120 python tools/conv_probe.py --sums --act none --flags 16 2>&1 | grep "^conv"
timeout 120 python tools/conv_probe.py --sums --act none --flags 17 2>&1 | grep "^conv"
