#!/usr/bin/env python3
"""Run-to-run stress of the unshipped gma_in + ConvPosEnc fusion (tools/ubench/gi_experiment.hip) built in two MFMA orders: gi_stress.py LIB [runs]
Counts the launches whose output differs from the first launch's, bit for bit, at the cfg3 size (8 x 544 x 960, 192 -> 80)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import torch
lib = C.CDLL(os.path.abspath(sys.argv[1]))
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 40
P, I = C.c_void_p, C.c_int
lib.rc_chain_packed_bytes.restype = C.c_size_t
lib.rc_chain_packed_bytes.argtypes = [I, I]
lib.rc_chain_pack_weights_natural.argtypes = [P, I, I, P]
lib.rc_dw_toeplitz_pack.argtypes = [P, I, I, P]
lib.rc_gi_exp.argtypes = [P, P, P, P, P, P, I, I, I, P]
lib.rc_last_error.restype = C.c_char_p
rng = np.random.default_rng(0)
w_in = (rng.standard_normal((80, 192)) * 0.07).astype(np.float32)
b_in = (rng.standard_normal(80) * 0.1).astype(np.float32)
taps = (rng.standard_normal((9, 80)) * 0.2).astype(np.float32)
b_cpe = (rng.standard_normal(80) * 0.1).astype(np.float32)
wp = np.empty(lib.rc_chain_packed_bytes(192, 80), np.uint8)
assert lib.rc_chain_pack_weights_natural(w_in.ctypes.data, 192, 80, wp.ctypes.data) == 0, lib.rc_last_error()
tp = np.empty(3 * 80 * 1024, np.uint8)
assert lib.rc_dw_toeplitz_pack(taps.ctypes.data, 3, 80, tp.ctypes.data) == 0, lib.rc_last_error()
dev = "cuda"
B, H, W = 8, 544, 960
g = torch.Generator(device=dev).manual_seed(3)
d1 = torch.randn(B, H, W, 192, generator=g, device=dev).to(torch.bfloat16)
d_w, d_t = torch.from_numpy(wp).to(dev), torch.from_numpy(tp).to(dev)
d_bi, d_bc = torch.from_numpy(b_in).to(dev), torch.from_numpy(b_cpe).to(dev)
x = torch.empty(B, H, W, 80, device=dev, dtype=torch.bfloat16)
stream = torch.cuda.current_stream().cuda_stream


def run():
    rc = lib.rc_gi_exp(d1.data_ptr(), d_w.data_ptr(), d_bi.data_ptr(), d_t.data_ptr(), d_bc.data_ptr(), x.data_ptr(), B, H, W, stream)
    assert rc == 0, lib.rc_last_error()
    torch.cuda.synchronize()
    return x.clone()


ref = run()
bad_runs, bad_elems = 0, 0
for _ in range(runs):
    y = run()
    n = int((y.view(torch.int16) != ref.view(torch.int16)).sum())
    bad_runs += n > 0; bad_elems += n
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): lib.rc_gi_exp(d1.data_ptr(), d_w.data_ptr(), d_bi.data_ptr(), d_t.data_ptr(), d_bc.data_ptr(), x.data_ptr(), B, H, W, stream)
e1.record(); torch.cuda.synchronize()
print(f"{os.path.basename(sys.argv[1])}: {bad_runs} of {runs} launches differ from the first ({bad_elems} elements in all); {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch; finite: {bool(torch.isfinite(ref.float()).all())}")
