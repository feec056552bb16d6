#!/usr/bin/env python
"""Do the kernels give bit-identical results when they share the GPU with other work?  (GPU box only.)

Every op below runs REPS times on one stream while a second stream keeps the GPU busy with large convolutions (persistent
workgroups on every CU), and each result is compared bit for bit with the same op run alone.  Staggered micro-batches
(runner.py) put kernels of different stages next to each other on a CU; a missing barrier or a hazard that lock-step execution
hides shows up here.

    python tools/concurrency_probe.py [--reps 40]
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as H  # noqa: E402
from disentangledcolorization_amd import _ffi, synth  # noqa: E402

L = _ffi.lib()


def sp(stream):
    return C.c_void_p(stream.cuda_stream)


def make_conv(n, cin, cout, hw, stride=1, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = H.to_act(torch.randn(n, cin, hw, hw, generator=g))
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    packed = H.pack_conv(w)
    bias = torch.randn(cout, generator=g).to(H.DEV)
    ho = (hw - 1) // stride + 1
    d = _ffi.ConvDesc(n, hw, hw, cin, 0, 0, 0, cout, stride, _ffi.ACT_RELU, 0.0, _ffi.PREC_F16X3, 0, 0, 0)
    def run(stream):
        out = torch.empty(2, n, ho, ho, cout, device=H.DEV, dtype=torch.float16)
        _ffi.check(L.disco_op_conv3x3(C.byref(d), _ffi.ptr(x), None, _ffi.ptr(packed), _ffi.ptr(bias), None, None, None, _ffi.ptr(out), sp(stream)))
        return out
    return run


def make_conv_mx(n, cin, cout, hw, q6, seed=0):
    g = torch.Generator().manual_seed(seed)
    planes = _ffi.PLANE_Q6 if q6 else _ffi.PLANE_Q
    x = H.to_act_mx(torch.relu(torch.randn(n, cin, hw, hw, generator=g)), planes=planes)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    packed, wexp = H.pack_conv_mx(w, 2 if q6 else 0)
    bias = torch.randn(cout, generator=g).to(H.DEV)
    d = _ffi.ConvMxDesc(n, hw, hw, cin, 0, 0, 0, x.sexp, 0, cout, 1, _ffi.ACT_RELU, 0.0, planes, 3, 0, 0, 0, 0, int(q6), 0)
    def run(stream):
        out = H.MxAct(n, cout, hw, hw, planes, 3)
        _ffi.check(L.disco_op_conv3x3_mx(C.byref(d), _ffi.ptr(x.buf), None, _ffi.ptr(packed), _ffi.ptr(wexp), _ffi.ptr(bias), None, None, None,
                                        _ffi.ptr(out.buf), None, None, sp(stream)))
        return out.buf
    return run


def make_encoder(n, l, seed=0):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_ops import _encoder_weights
    sd = synth.synth_state_dict(130)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, l, 64, generator=g).to(H.DEV)
    pos = torch.randn(l, 64, generator=g).to(H.DEV)
    wts = _encoder_weights(sd, "wildpath").to(H.DEV)
    def run(stream):
        out = torch.empty_like(x)
        ws = torch.empty(n * l * 704 * 4, device=H.DEV, dtype=torch.uint8)
        _ffi.check(L.disco_op_encoder_stack(_ffi.ptr(x), _ffi.ptr(pos), _ffi.ptr(wts), _ffi.ptr(out), n, l, _ffi.ptr(ws), ws.numel(), sp(stream)))
        return out
    return run


def make_pool(n, c, hw, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, hw, hw, generator=g).to(H.DEV)
    p = torch.softmax(torch.randn(n, 9, hw, hw, generator=g), 1).to(H.DEV)
    h = hw // 16
    def run(stream):
        pooled = torch.empty(n, c, h, h, device=H.DEV); conf = torch.empty(n, 1, h, h, device=H.DEV)
        ws = torch.empty(n * h * h * 9 * (c + 2) * 4, dtype=torch.uint8, device=H.DEV)
        _ffi.check(L.disco_op_poolfeat(_ffi.ptr(x), _ffi.ptr(p), _ffi.ptr(pooled), _ffi.ptr(conf), None, n, c, hw, hw, 16, _ffi.ptr(ws), ws.numel(), sp(stream)))
        return torch.cat([pooled.flatten(), conf.flatten()])
    return run


def make_pool_act(n, hw, seed=0):
    """The forward's own pooling launch (disco_op_poolfeat_act): 64 act channels hi + lo and two fp32 channels -> tokens."""
    g = torch.Generator().manual_seed(seed)
    feat = H.to_act(torch.relu(torch.randn(n, 64, hw, hw, generator=g)) * 3)
    ab = torch.randn(n, 2, hw, hw, generator=g).to(H.DEV)
    p = torch.softmax(torch.randn(n, 9, hw, hw, generator=g) * 2, 1).to(H.DEV)
    h = hw // 16
    def run(stream):
        tok = torch.empty(n, h * h, 64, device=H.DEV); pooled2 = torch.empty(n, 2, h, h, device=H.DEV)
        ws = torch.empty(n * h * h * 9 * 68 * 4, dtype=torch.uint8, device=H.DEV)
        _ffi.check(L.disco_op_poolfeat_act(_ffi.ptr(feat), _ffi.ptr(ab), _ffi.ptr(p), _ffi.ptr(tok), _ffi.ptr(pooled2), n, hw, hw, _ffi.ptr(ws), ws.numel(), sp(stream)))
        return torch.cat([tok.flatten(), pooled2.flatten()])
    return run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--only", default="")
    ap.add_argument("--bg-streams", type=int, default=1, help="number of background streams (each loops the background ops)")
    ap.add_argument("--bg-set", default="", help="with --bg mix: comma-separated indices of the background ops to keep (0 conv f16x3 512ch, 1 conv mx6 256ch, "
                                                  "2 conv f16x3 64ch @256, 3 conv f16x3 stride 2, 4 encoder stack, 5 pooling (act path), 6 conv mx6 64ch @256)")
    ap.add_argument("--bg", default="conv", help="conv: two large convolutions; mix: convolutions of several shapes / arithmetics, encoder stacks and poolings, as concurrent forwards would run them")
    args = ap.parse_args()
    ops = {
        "conv f16x3 256->256 @64 n=8": make_conv(8, 256, 256, 64),
        "conv f16x3 512->512 @32 n=8": make_conv(8, 512, 512, 32),
        "conv f16x3 128->256 s2 @128 n=8": make_conv(8, 128, 256, 128, 2),
        "conv f16x3 64->64 @256 n=4": make_conv(4, 64, 64, 256),
        "conv mx6 256->256 @64 n=8": make_conv_mx(8, 256, 256, 64, True),
        "conv mx8 256->256 @64 n=8": make_conv_mx(8, 256, 256, 64, False),
        "encoder stack n=8 l=256": make_encoder(8, 256),
        "poolfeat n=8 c=64 256^2": make_pool(8, 64, 256),
        "poolfeat (forward's act path) n=8 256^2": make_pool_act(8, 256),
    }
    if args.bg == "mix":
        bg_ops = [make_conv(8, 512, 512, 32, seed=5), make_conv_mx(8, 256, 256, 64, True, seed=6), make_conv(8, 64, 64, 256, seed=7), make_conv(8, 128, 256, 128, 2, seed=8),
                  make_encoder(8, 256, seed=9), make_pool_act(8, 256, seed=10), make_conv_mx(8, 64, 64, 256, True, seed=11)]
        if args.bg_set:
            bg_ops = [bg_ops[int(i)] for i in args.bg_set.split(",")]
    else:
        bg_ops = [make_conv(64, 512, 512, 32, seed=5), make_conv_mx(64, 256, 256, 64, True, seed=6)]
    s_bgs = [torch.cuda.Stream() for _ in range(args.bg_streams)]
    s_t = torch.cuda.Stream()
    bad_total = 0
    for name, op in ops.items():
        if args.only and args.only not in name:
            continue
        with torch.cuda.stream(s_t):
            ref = op(s_t).clone()
        torch.cuda.synchronize()
        keep = []
        for i in range(6 * args.reps):
            for j, s_bg in enumerate(s_bgs):
                with torch.cuda.stream(s_bg):
                    keep.append(bg_ops[(i + j) % len(bg_ops)](s_bg))
            if len(keep) > 8 * len(s_bgs):
                del keep[:len(s_bgs)]
        outs = []
        with torch.cuda.stream(s_t):
            for _ in range(args.reps):
                outs.append(op(s_t))
        torch.cuda.synchronize()
        bad = [i for i, o in enumerate(outs) if not torch.equal(o, ref)]
        bad_total += len(bad)
        detail = ""
        if bad:
            o = outs[bad[0]]
            diff = (o.view(torch.uint8) != ref.view(torch.uint8)) if o.dtype != torch.uint8 else (o != ref)
            detail = "  first bad rep %d: %d of %d bytes differ" % (bad[0], int(diff.sum()), diff.numel())
        print(f"{name:36s} {len(bad):3d} / {args.reps} runs differ from the run alone{detail}", flush=True)
    print("TOTAL mismatching runs:", bad_total)
    return 1 if bad_total else 0


if __name__ == "__main__":
    sys.exit(main())
