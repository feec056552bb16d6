"""Helpers for the -m gpu parity tests: thin wrappers over the C ABI (through ctypes) so that the
tests read like calls of the reference's operators."""
import ctypes as C

import numpy as np
import torch

from disentangledcolorization_amd import _ffi

DEV = "cuda:0"


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_act(x, c_pad=None):
    """fp32 NCHW (cpu or cuda) -> act tensor (2,N,H,W,c_pad) fp16 on the GPU (hi plane, lo plane)."""
    x = x.to(DEV).float().contiguous()
    n, c, h, w = x.shape
    c_pad = c_pad or c
    out = torch.empty(2, n, h, w, c_pad, device=DEV, dtype=torch.float16)
    _ffi.check(_ffi.lib().disco_op_nchw_to_act(_ffi.ptr(x), _ffi.ptr(out), n, c, h, w, c_pad, stream()))
    return out


def from_act(a, c=None):
    """act tensor (2,N,H,W,c_pad) -> fp32 NCHW on the GPU."""
    _, n, h, w, c_pad = a.shape
    c = c or c_pad
    out = torch.empty(n, c, h, w, device=DEV, dtype=torch.float32)
    _ffi.check(_ffi.lib().disco_op_act_to_nchw(_ffi.ptr(a), _ffi.ptr(out), n, c, h, w, c_pad, stream()))
    return out


def to_act_scaled(x, sexp):
    """As to_act, every plane storing x 2^sexp (exact: a power of two)."""
    return to_act(x.float() * 2.0 ** sexp)


def from_act_scaled(a, sexp, c=None):
    return from_act(a, c) * 2.0 ** -sexp


def pack_conv(w):
    """(Cout,Cin,3,3) fp32 cpu -> packed device buffer."""
    w = w.detach().cpu().float().contiguous()
    co, ci = w.shape[:2]
    nbytes = C.c_size_t()
    pack = _ffi.lib().disco_op_conv3x3_pack
    _ffi.check(pack(None, co, ci, None, C.byref(nbytes)))
    buf = torch.empty(nbytes.value, device=DEV, dtype=torch.uint8)
    _ffi.check(pack(_ffi.ptr(w), co, ci, _ffi.ptr(buf), C.byref(nbytes)))
    return buf


def conv3x3(src0, w, bias=None, *, src1=None, up0=False, up1=False, stride=1, act=_ffi.ACT_NONE, slope=0.0,
            bn_scale=None, bn_shift=None, res=None, precision=_ffi.PREC_F16X3, sexp_in=0, sexp_out=0, sexp_res=0):
    """HIP conv on act tensors; returns the act output.  src*: (2,N,h,w,C) fp16.  sexp_*: the scale exponents the buffers carry
    (to_act_scaled / from_act_scaled below convert with one)."""
    n = src0.shape[1]
    h_in = src0.shape[2] * (2 if up0 else 1)
    w_in = src0.shape[3] * (2 if up0 else 1)
    c0 = src0.shape[4]
    c1 = src1.shape[4] if src1 is not None else 0
    co = w.shape[0]
    packed = pack_conv(w)
    d = _ffi.ConvDesc(n, h_in, w_in, c0, c1, int(up0), int(up1), co, stride, act, slope, precision, sexp_in, sexp_out, sexp_res)
    ho, wo = (h_in - 1) // stride + 1, (w_in - 1) // stride + 1
    out = torch.empty(2, n, ho, wo, co, device=DEV, dtype=torch.float16)
    dv = lambda t: None if t is None else t.to(DEV).float().contiguous()
    bias, bn_scale, bn_shift = dv(bias), dv(bn_scale), dv(bn_shift)
    _ffi.check(_ffi.lib().disco_op_conv3x3(C.byref(d), _ffi.ptr(src0), _ffi.ptr(src1), _ffi.ptr(packed), _ffi.ptr(bias),
                                          _ffi.ptr(bn_scale), _ffi.ptr(bn_shift), _ffi.ptr(res), _ffi.ptr(out), stream()))
    torch.cuda.synchronize()
    return out


# ---- the fp16 + fp8-correction conv (csrc/conv_mx.hip) ------------------------------------------------------------------
def sexp_for(x, target_bits=5):
    """Power-of-two scale exponent that maps max|x| into [2^(target_bits-1), 2^target_bits) (fp8 e4m3 tops out at 448)."""
    import math
    m = float(x.abs().max())
    return 0 if m == 0 else target_bits - 1 - math.floor(math.log2(m))


def act_bytes(n, c_pad, h, w, planes):
    nb = C.c_size_t()
    _ffi.check(_ffi.lib().disco_op_act_bytes(n, c_pad, h, w, planes, C.byref(nb)))
    return nb.value


class MxAct:
    """A flat activation buffer of the mx path: hi plane [+ lo plane] [+ fp8 q planes] (include/disco_hip.h)."""

    def __init__(self, n, c, h, w, planes, sexp=0, c_pad=None):
        self.n, self.c, self.h, self.w, self.planes, self.sexp = n, c, h, w, planes, sexp
        self.c_pad = c_pad or c
        self.buf = torch.zeros(act_bytes(n, self.c_pad, h, w, planes), device=DEV, dtype=torch.uint8)

    def read(self, which=0):
        out = torch.empty(self.n, self.c, self.h, self.w, device=DEV, dtype=torch.float32)
        _ffi.check(_ffi.lib().disco_op_act_mx_to_nchw(_ffi.ptr(self.buf), _ffi.ptr(out), self.n, self.c, self.h, self.w, self.c_pad,
                                                     self.planes, self.sexp, which, stream()))
        torch.cuda.synchronize()
        return out


def to_act_mx(x, planes=_ffi.PLANE_Q, sexp=None, c_pad=None):
    x = x.to(DEV).float().contiguous()
    n, c, h, w = x.shape
    a = MxAct(n, c, h, w, planes, sexp_for(x) if sexp is None else sexp, c_pad)
    _ffi.check(_ffi.lib().disco_op_nchw_to_act_mx(_ffi.ptr(x), _ffi.ptr(a.buf), n, c, h, w, a.c_pad, planes, a.sexp, stream()))
    return a


def pack_conv_mx(w, x2q=False):
    w = w.detach().cpu().float().contiguous()
    co, ci = w.shape[:2]
    nbytes = C.c_size_t()
    _ffi.check(_ffi.lib().disco_op_conv3x3_mx_pack(None, co, ci, int(x2q), None, None, C.byref(nbytes)))        # x2q: the pack variant 0 / 1 / 2
    buf = torch.empty(nbytes.value, device=DEV, dtype=torch.uint8)
    wexp = torch.empty((co + 31) // 32 * 32, device=DEV, dtype=torch.int32)
    _ffi.check(_ffi.lib().disco_op_conv3x3_mx_pack(_ffi.ptr(w), co, ci, int(x2q), _ffi.ptr(buf), _ffi.ptr(wexp), C.byref(nbytes)))
    return buf, wexp


def conv3x3_mx(src0, w, bias=None, *, src1=None, up0=False, up1=False, stride=1, act=_ffi.ACT_NONE, slope=0.0, bn_scale=None,
               bn_shift=None, res=None, out_planes=_ffi.PLANE_LO, out_sexp=0, out_f32=False, packed=None, x2q=False, d2s=False, tapmask=False, q6=False):
    """src*: MxAct with q planes (x2q: one source with al8-only planes, PLANE_QL); res: MxAct (hi [+ lo]).
    Returns (MxAct | fp32 NCHW tensor, saturation count)."""
    h_in, w_in = src0.h * (2 if up0 else 1), src0.w * (2 if up0 else 1)
    co = w.shape[0]
    buf, wexp = packed or pack_conv_mx(w, 2 if q6 else int(x2q))
    ho, wo = (h_in - 1) // stride + 1, (w_in - 1) // stride + 1
    d = _ffi.ConvMxDesc(src0.n, h_in, w_in, src0.c_pad, src1.c_pad if src1 is not None else 0, int(up0), int(up1), src0.sexp,
                        src1.sexp if src1 is not None else 0, co, stride, act, slope, out_planes, out_sexp, int(out_f32),
                        res.planes if res is not None else 0, res.sexp if res is not None else 0, int(x2q), int(q6), int(d2s))
    out = torch.empty(src0.n, co, ho, wo, device=DEV, dtype=torch.float32) if out_f32 else (
        MxAct(src0.n, co // 4, 2 * ho, 2 * wo, out_planes, out_sexp) if d2s else MxAct(src0.n, co, ho, wo, out_planes, out_sexp))
    sat = torch.zeros(1, device=DEV, dtype=torch.int32)
    mask = None
    if tapmask:         # the forward's masked instantiations: taps without any non-zero weight are skipped per 32-cout block
        mask = torch.zeros((co + 31) // 32, device=DEV, dtype=torch.int32)
        _ffi.check(_ffi.lib().disco_op_conv3x3_tapmask(_ffi.ptr(w.detach().cpu().float().contiguous()), co, w.shape[1], _ffi.ptr(mask)))
    dv = lambda t: None if t is None else t.to(DEV).float().contiguous()
    bias, bn_scale, bn_shift = dv(bias), dv(bn_scale), dv(bn_shift)
    _ffi.check(_ffi.lib().disco_op_conv3x3_mx(C.byref(d), _ffi.ptr(src0.buf), _ffi.ptr(src1.buf) if src1 is not None else None,
                                             _ffi.ptr(buf), _ffi.ptr(wexp), _ffi.ptr(bias), _ffi.ptr(bn_scale), _ffi.ptr(bn_shift),
                                             _ffi.ptr(res.buf) if res is not None else None,
                                             _ffi.ptr(out) if out_f32 else _ffi.ptr(out.buf), _ffi.ptr(sat), _ffi.ptr(mask), stream()))
    torch.cuda.synchronize()
    return out, int(sat.item())


def to_gray_tail(g, sexp):
    """(n,1,h,w) fp32 -> the 16-channel fp16 tail source (x_hi, x_lo, x_hi, 0...) of a two-source f16+fp8x2 layer (disco_op_gray_tail)."""
    g = g.to(DEV).float().contiguous()
    n, _, h, w = g.shape
    a = MxAct(n, 16, h, w, 0, sexp)
    _ffi.check(_ffi.lib().disco_op_gray_tail(_ffi.ptr(g), _ffi.ptr(a.buf), n, h, w, sexp, stream()))
    return a


def max_err(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()
