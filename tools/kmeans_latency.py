#!/usr/bin/env python
"""Latency of the k-means + anchors launch on one image's tokens (GPU box): per number of Lloyd passes, DISCO_KMEANS_V1=0/1.
    python tools/kmeans_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import gpu_helpers as H
from disentangledcolorization_amd import _ffi

def run(x, k, iters=50):
    n, l, d = x.shape
    xd = x.to(H.DEV).contiguous(); sizes = torch.rand(n, l, device=H.DEV)
    idx = torch.as_tensor(np.stack([np.random.RandomState(i).choice(l, k, replace=False) for i in range(n)]).astype(np.int32)).to(H.DEV)
    assign = torch.empty(n, l, dtype=torch.int32, device=H.DEV); anchor = torch.empty(n, k, dtype=torch.int32, device=H.DEV)
    mask = torch.empty(n, l, device=H.DEV); info = torch.empty(n, 2, dtype=torch.int32, device=H.DEV)
    wsb = _ffi.lib().disco_op_kmeans_workspace_bytes(n, l)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=H.DEV)
    def f():
        _ffi.check(_ffi.lib().disco_op_kmeans_anchors_ws(_ffi.ptr(xd), _ffi.ptr(sizes), _ffi.ptr(idx), None, 0, _ffi.ptr(assign), _ffi.ptr(anchor),
                                                         _ffi.ptr(mask), _ffi.ptr(info), n, l, k, d, 0, _ffi.ptr(ws), wsb, H.stream()))
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3, info.cpu()[:, 0].tolist()

g = torch.Generator().manual_seed(0)
for l, k, spread in [(256, 8, 0.0), (256, 8, 0.7), (256, 8, 3.0), (256, 16, 0.7), (256, 32, 0.7), (128, 8, 0.7), (1024, 8, 0.7), (1024, 8, 3.0), (1536, 8, 0.7), (1536, 8, 3.0),
                     (1536, 16, 3.0), (4096, 8, 3.0), (16384, 8, 3.0)]:
    c = torch.randn(1, k, 64, generator=g) * 2
    which = torch.randint(0, k, (1, l), generator=g)
    x = torch.gather(c, 1, which[..., None].expand(-1, -1, 64)) + torch.randn(1, l, 64, generator=g) * spread
    us, passes = run(x, k)
    print("L=%4d K=%2d spread %.2f: %6.1f us, passes %s  -> %.1f us/pass" % (l, k, spread, us, passes, us / max(1, passes[0])))
