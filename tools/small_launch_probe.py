#!/usr/bin/env python3
"""The codec's small dependent launches in isolation: one CompressAI residual unit (1x1 N -> N/2, ReLU; 3x3, ReLU; 1x1 N/2 -> N, + x, ReLU) at the slice loop's map
(B x 72 x 120, N = 128), launch by launch and as a chain, back to back on one stream (HIP events over n launches).  small_launch_probe.py [B] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import tcm as T, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev, dt = torch.device("cuda:0"), torch.bfloat16
torch.manual_seed(0)
N_ = 128
u = T._ResidualUnit(N_).to(dev, dt).eval()
att = T.AttentionBlock(N_).to(dev, dt).eval()
a = torch.randn(B, 72, 120, N_, device=dev).to(dt)


def timed(fn, n=n, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    t1 = u.conv[0]._nhwc(a, act="relu")
    t2 = u.conv[2]._nhwc(t1, act="relu")
    print(f"B = {B}, 72 x 120, N = {N_}: us per launch, {n} launches back to back")
    print(f"  1x1 {N_} -> {N_ // 2} + ReLU                 {timed(lambda: u.conv[0]._nhwc(a, act='relu')):7.1f}")
    print(f"  3x3 {N_ // 2} -> {N_ // 2} + ReLU                  {timed(lambda: u.conv[2]._nhwc(t1, act='relu')):7.1f}")
    print(f"  1x1 {N_ // 2} -> {N_} + x + ReLU             {timed(lambda: u.conv[4]._nhwc(t2, act='relu_post', residual=a)):7.1f}")
    print(f"  1x1 {N_ // 2} -> {N_} (plain)                {timed(lambda: u.conv[4]._nhwc(t2)):7.1f}")
    print(f"  residual unit (3 launches)           {timed(lambda: u._nhwc(a)):7.1f}")
    print(f"  AttentionBlock (19 launches + gate)  {timed(lambda: att._nhwc(a)):7.1f}")
    print(f"  sigmoid_gate_add                     {timed(lambda: ops.sigmoid_gate_add(a, a, a)):7.1f}")
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): u._nhwc(a)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            x = a
            for _ in range(10): x = u._nhwc(x)
    print(f"  residual unit in a graph (10 units)  {timed(lambda: g.replay(), n=50) / 10:7.1f}  per unit")
