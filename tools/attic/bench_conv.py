#!/usr/bin/env python
"""Micro-benchmark of the MFMA conv kernel on the layer shapes of the DISCO forward (GPU box only).

    python tools/bench_conv.py [--n 64] [--iters 10]
Prints ms and algorithmic / executed TFLOP/s per shape.  Set DISCO_CONV_V1=1 for the first-generation kernel.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C  # noqa: E402

import gpu_helpers as H  # noqa: E402
from disentangledcolorization_amd import _ffi  # noqa: E402

SHAPES = [  # name, cin0, cin1, cout, h_in (logical), stride, up0
    ("512->512 @32", 512, 0, 512, 32, 1, 0),
    ("256->256 @64", 256, 0, 256, 64, 1, 0),
    ("128->128 @128", 128, 0, 128, 128, 1, 0),
    ("64->64 @256", 64, 0, 64, 256, 1, 0),
    ("up 512->256 @64", 512, 0, 256, 64, 1, 1),
    ("up 128->64 @256", 128, 0, 64, 256, 1, 1),
    ("cat 64+64->64 @256", 64, 64, 64, 256, 1, 1),
    ("s2 64->128 @256", 64, 0, 128, 256, 2, 0),
    ("s2 256->512 @64", 256, 0, 512, 64, 2, 0),
    ("64->2(32) @256", 64, 0, 2, 256, 1, 0),
    ("16->16 @256", 16, 0, 16, 256, 1, 0),
    ("cat 16+16->16 @256", 16, 16, 16, 256, 1, 0),      # SpixelNet's conv0_1 ...
    ("16->9(32) @256", 16, 0, 9, 256, 1, 0),             # ... and pred_mask0 (here with an activation-tensor output)
]


def bench_mx(L, args, name, c0, c1, co, hin, stride, up0):
    n = args.n
    hs = hin // 2 if up0 else hin
    z = 0.0 if args.zeros else 1.0
    x0 = H.to_act_mx(z * torch.relu(torch.randn(n, c0, hs, hs, device="cuda")), sexp=2)
    x1 = H.to_act_mx(z * torch.relu(torch.randn(n, c1, hin, hin, device="cuda")), sexp=2) if c1 else None
    w = torch.randn(co, c0 + c1, 3, 3) * 0.05 * z
    xq = args.mx in (3, 4)
    q6 = args.mx == 6
    planes = _ffi.PLANE_QL if xq else (_ffi.PLANE_Q6 if q6 else _ffi.PLANE_Q)
    if xq or q6:
        x0 = H.to_act_mx(z * torch.relu(torch.randn(n, c0, hs, hs, device="cuda")), planes=planes, sexp=2)
        x1 = H.to_act_mx(z * torch.relu(torch.randn(n, c1, hin, hin, device="cuda")), planes=planes, sexp=2) if c1 else None
    packed, wexp = H.pack_conv_mx(w, 2 if q6 else int(xq))
    ho = (hin - 1) // stride + 1
    out = H.MxAct(n, co, ho, ho, planes, 0)
    bias = torch.zeros(co, device="cuda")
    d = _ffi.ConvMxDesc(n, hin, hin, c0, c1, up0, 0, x0.sexp, x1.sexp if x1 else 0, co, stride, _ffi.ACT_RELU, 0.0, planes, 0, 0, 0, 0, int(xq), int(q6), 0)

    def run():
        _ffi.check(L.disco_op_conv3x3_mx(C.byref(d), _ffi.ptr(x0.buf), _ffi.ptr(x1.buf) if x1 else None, _ffi.ptr(packed), _ffi.ptr(wexp),
                                        _ffi.ptr(bias), None, None, None, _ffi.ptr(out.buf), None, None, H.stream()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    fl = 2.0 * 9 * (c0 + c1) * co * ho * ho * n
    print(f"{name:22s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF alg  {2 * fl / ms / 1e9:8.1f} TF-equivalent pipe units   [{'x2q' if xq else ('mx6' if q6 else 'mx8')}]", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--prec", type=int, default=0)
    ap.add_argument("--only", default="", help="substring filter on the shape name")
    ap.add_argument("--zeros", type=int, default=0, help="1: all-zero activations and weights (how much of the time is the power-managed clock?)")
    ap.add_argument("--mx", type=int, default=0, help="1: the fp16 + fp8-correction kernel (conv_mx.hip); 2: both, side by side; 3: its f16x2 + fp8 arithmetic alone; 4: that and f16x3 side by side; 6: its f16 + fp6x2 arithmetic alone")
    args = ap.parse_args()
    L = _ffi.lib()
    for name, c0, c1, co, hin, stride, up0 in SHAPES:
        if args.only and args.only not in name:
            continue
        n = args.n
        if args.mx and (c0 % 32 or c1 % 32 or co < 32):
            continue
        if args.mx in (3, 4) and (c0 % 64 or c1):
            continue
        if args.mx:
            bench_mx(L, args, name, c0, c1, co, hin, stride, up0)
            if args.mx in (1, 3, 6):
                continue
        hs = hin // 2 if up0 else hin
        z = 0.0 if args.zeros else 1.0       # --zeros: the same launch on all-zero operands (how much of the time is the power-managed clock?)
        src0 = (z * torch.randn(2, n, hs, hs, c0, device="cuda")).half()
        src1 = (z * torch.randn(2, n, hin, hin, c1, device="cuda")).half() if c1 else None
        w = torch.randn(co, c0 + c1, 3, 3) * 0.05 * z
        packed = H.pack_conv(w)
        ho = (hin - 1) // stride + 1
        out = torch.empty(2, n, ho, ho, co, device="cuda", dtype=torch.float16)
        bias = torch.zeros(co, device="cuda")
        d = _ffi.ConvDesc(n, hin, hin, c0, c1, up0, 0, co, stride, _ffi.ACT_RELU, 0.0, args.prec)

        def run():
            _ffi.check(L.disco_op_conv3x3(C.byref(d), _ffi.ptr(src0), _ffi.ptr(src1), _ffi.ptr(packed), _ffi.ptr(bias), None,
                                          None, None, _ffi.ptr(out), H.stream()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        fl = 2.0 * 9 * (c0 + c1) * co * ho * ho * n
        mult = 3 if args.prec == 0 else 1
        print(f"{name:22s} {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF alg  {mult * fl / ms / 1e9:8.1f} TF mfma", flush=True)


if __name__ == "__main__":
    main()
