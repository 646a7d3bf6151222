#!/usr/bin/env python3
"""Sustained MFMA rate of this GPU (pure v_mfma_f32_16x16x32_bf16 issue on every SIMD) -- calibration for MFMA utilisation."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from realcamnet_amd import ops
torch.zeros(1, device="cuda")
for wps in (1, 2):
    tf, ticks = C.c_double(), C.c_double()
    for iters in (20000, 200000):
        ops.check(ops.lib().rc_debug_mfma_peak(wps, iters, C.byref(tf), C.byref(ticks)), "mfma_peak")
        print(f"{wps} wave(s)/SIMD, {iters} x16 MFMAs/wave: {tf.value:7.1f} TF/s   {ticks.value:5.2f} s_memtime ticks per MFMA per SIMD")
for wps in (1, 2):
    tf, ticks = C.c_double(), C.c_double()
    for iters in (20000, 200000):
        ops.check(ops.lib().rc_debug_mfma_peak32(wps, iters, C.byref(tf), C.byref(ticks)), "mfma_peak32")
        print(f"32x32x16: {wps} wave(s)/SIMD, {iters} x8 MFMAs/wave: {tf.value:7.1f} TF/s   {ticks.value:5.2f} s_memtime ticks per MFMA per SIMD")
