#!/usr/bin/env python
"""Per-chunk timeline of the conv kernel's workgroup 0 (s_memtime stamps; GPU box only):
   python tools/conv_timeline.py [--cin 512 --cout 512 --size 32]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_helpers as H
from disentangledcolorization_amd import _ffi

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=512); ap.add_argument("--cout", type=int, default=512)
ap.add_argument("--size", type=int, default=32); ap.add_argument("--n", type=int, default=64)
ap.add_argument("--stride", type=int, default=1); ap.add_argument("--s2d", type=int, default=0)
a = ap.parse_args()
L = _ffi.lib()
src = torch.randn(2, a.n, a.size, a.size, a.cin, device="cuda").half()
w = torch.randn(a.cout, a.cin, 3, 3) * 0.05
packed = H.pack_conv(w, bool(a.s2d))
so = (a.size - 1) // a.stride + 1
out = torch.empty(2, a.n, so, so, a.cout, device="cuda", dtype=torch.float16)
bias = torch.zeros(a.cout, device="cuda")
d = _ffi.ConvDesc(a.n, a.size, a.size, a.cin, 0, 0, 0, a.cout, a.stride, _ffi.ACT_RELU, 0.0, 0, a.s2d)
run = lambda: _ffi.check(L.disco_op_conv3x3(C.byref(d), _ffi.ptr(src), None, _ffi.ptr(packed), _ffi.ptr(bias), None, None, None, _ffi.ptr(out), H.stream()))
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print("avg launch %.3f ms" % (e0.elapsed_time(e1) / 10))
buf = torch.zeros(16 * 64 * 4, dtype=torch.int64, device="cuda")
L.disco_op_conv3x3_set_probe(_ffi.ptr(buf)); run(); torch.cuda.synchronize(); L.disco_op_conv3x3_set_probe(None)
t = buf.cpu().view(16, 64, 4)
for wv in (0, 3, 4, 7):
    x = t[wv]
    nz = int((x[:, 3] != 0).sum())
    x = x[:nz].double()
    wait = (x[:, 1] - x[:, 0]); bar = (x[:, 2] - x[:, 1]); comp = (x[:, 3] - x[:, 2])
    period = (x[1:, 0] - x[:-1, 0])
    print(f"wave {wv}: chunks {nz}  dma-wait {wait[1:].mean():7.0f}  barrier {bar[1:].mean():7.0f}  compute {comp[1:].mean():7.0f}  period {period.mean():7.0f} ticks  (first 6 periods {period[:6].tolist()})")
