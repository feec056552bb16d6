#!/usr/bin/env python
"""Run-to-run determinism of conv3x3_mx_kernel per arithmetic and layer shape (GPU box): every launch is repeated and the raw
output buffers are compared byte for byte; differing bytes are attributed to their plane and lane pattern.
    python tools/conv_determinism_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_helpers as H  # noqa: E402
from disentangledcolorization_amd import _ffi  # noqa: E402

SHAPES = [(64, 64, 64, 64, 1, 8), (128, 128, 32, 32, 1, 8), (512, 512, 32, 32, 1, 8), (64, 128, 64, 64, 2, 8), (256, 256, 16, 16, 1, 64), (64, 64, 256, 256, 1, 4)]
for x2q in (True, False):
    for cin, cout, h, w, stride, n in SHAPES:
        g = torch.Generator().manual_seed(cin + cout + h)
        x = torch.relu(torch.randn(n, cin, h, w, generator=g))
        wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout, generator=g) * 0.1
        src = H.to_act_mx(x, _ffi.PLANE_QL if x2q else _ffi.PLANE_Q)
        planes = _ffi.PLANE_QL if x2q else _ffi.PLANE_Q
        packed = H.pack_conv_mx(wt, x2q)
        outs = []
        for rep in range(6):
            o, sat = H.conv3x3_mx(src, wt, b, stride=stride, act=_ffi.ACT_LRELU, slope=0.2, out_planes=planes, out_sexp=3, packed=packed, x2q=x2q)
            outs.append(o.buf.clone())
        bad = [int((o != outs[0]).sum()) for o in outs[1:]]
        msg = "x2q" if x2q else "mx8"
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        hi_bytes = n * cout * ho * wo * 2
        detail = ""
        if any(bad):
            d = (outs[[i for i, v in enumerate(bad) if v][0] + 1] != outs[0]).nonzero().flatten()
            in_hi = int((d < hi_bytes).sum())
            detail = " first diffs at bytes %s; %d in the hi plane, %d in the q planes; byte %% 64 histogram: %s" % (
                d[:8].tolist(), in_hi, len(d) - in_hi, torch.bincount((d % 64), minlength=64).tolist())
        print("%s %d->%d @%dx%d s%d n=%d sat=%d: differing bytes per repeat %s%s" % (msg, cin, cout, h, w, stride, n, sat, bad, detail), flush=True)

# the forward's special layer kinds: sub-pixel up-conv (4 live taps per phase: masked instantiation, depth-to-space epilogue) with a
# residual that carries a lo plane (repnet.conv8up.1), and lo-only outputs (conv3short8 / conv10_2.1)
import numpy as np  # noqa: E402
for x2q in (True, False):
    planes = _ffi.PLANE_QL if x2q else _ffi.PLANE_Q
    for cin, C4, h, w, n, with_res, out_planes in [(512, 256, 32, 32, 8, True, planes), (128, 64, 64, 64, 8, False, planes), (256, 256, 64, 64, 4, False, _ffi.PLANE_LO),
                                                   (64, 64, 128, 128, 4, False, _ffi.PLANE_LO)]:
        g = torch.Generator().manual_seed(cin + C4 + h)
        x = torch.relu(torch.randn(n, cin, h, w, generator=g))
        d2s = out_planes == planes
        if d2s:
            w3 = torch.randn(C4, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
            # upsample + 3x3 as a 4-phase conv on the low-res grid: phase (py, px) uses a 2x2 block of (pre-summed) taps, the rest are zero
            wt = torch.zeros(4 * C4, cin, 3, 3)
            for py in range(2):
                for px in range(2):
                    for ky in range(3):
                        for kx in range(3):
                            dy, dx = (py + ky - 1) >> 1, (px + kx - 1) >> 1
                            wt[(py * 2 + px) * C4:(py * 2 + px + 1) * C4, :, dy + 1, dx + 1] += w3[:, :, ky, kx]
            b = torch.randn(C4, generator=g) * 0.1
        else:
            wt = torch.randn(C4, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
            b = torch.randn(C4, generator=g) * 0.1
        src = H.to_act_mx(x, planes)
        res = H.to_act_mx(torch.randn(n, C4, 2 * h, 2 * w, generator=g), _ffi.PLANE_LO) if with_res else None
        packed = H.pack_conv_mx(wt, x2q)
        outs = []
        for rep in range(6):
            o, sat = H.conv3x3_mx(src, wt, b, act=_ffi.ACT_RELU, out_planes=out_planes, out_sexp=3, packed=packed, x2q=x2q, d2s=d2s, tapmask=d2s, res=res)
            outs.append(o.buf.clone())
        bad = [int((o != outs[0]).sum()) for o in outs[1:]]
        detail = ""
        if any(bad):
            d = (outs[[i for i, v in enumerate(bad) if v][0] + 1] != outs[0]).nonzero().flatten()
            co, ho, wo = (C4, 2 * h, 2 * w) if d2s else (C4, h, w)
            hi_bytes = n * co * ho * wo * 2
            detail = " first diffs at bytes %s; %d in the hi plane, %d beyond; byte %% 64 histogram: %s" % (
                d[:8].tolist(), int((d < hi_bytes).sum()), int((d >= hi_bytes).sum()), torch.bincount((d % 64), minlength=64).tolist())
        print("%s %s %d->%d @%dx%d n=%d res=%s out_planes=%d sat=%d: differing bytes per repeat %s%s" % (
            "x2q" if x2q else "mx8", "up-conv (masked, d2s)" if d2s else "plain", cin, C4, h, w, n, with_res, out_planes, sat, bad, detail), flush=True)
