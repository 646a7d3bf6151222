#!/bin/bash
# A/B of round 4's two schedule changes on the same box, same build: default | tail as two launches | round 3's staged CALayer gate | both off (= round 3's schedule)
for F in "" "--layer-by-layer-tail" "--staged-gate" "--layer-by-layer-tail --staged-gate"; do
  python bench.py --no-cpu-baseline --no-codec-leg $F 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-45s %8.2f MP/s  %7.3f ms/step  conv kernels %7.3f ms  flops/step %.2f T  frac %.4f' % ('$F' or '(default: folded tail, early gate)', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['flops_per_step']/1e12, d['roofline']['frac']))"
done
