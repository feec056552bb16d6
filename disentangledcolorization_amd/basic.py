"""HIP-backed mirror of the `models/basic.py` helpers the reference's inference scripts call around the forward
(SURVEY §8f rows 1-2, 4): same names, argument meaning and error behaviour, CUDA/HIP tensors in and out.

    tensor2array            basic.py:10-12      (inference.py:119,124)
    poolfeat                basic.py:274-324    (spixelseg/inference.py:91,107)
    get_spixel_size         basic.py:327-335
    upfeat                  basic.py:338-376    (inference.py:115,129; spixelseg/inference.py:108)
    ColorLabel.decode_ind2ab  basic.py:196-218  (inference.py:114, integer T)
    rgb2lab / lab2rgb       basic.py:395-475
    mark_color_hints        basic.py:95-117     (inference.py:130)
    fetch_data_from_rgb8    main/colorizer/inference.py:23-42 after the image decode (pad-to-16 branch)
    normLabs_to_rgb8        utils/util.py:91-106 + batch_depadding (inference.py:44-49) before the image encode

No CPU fallback: CPU tensors raise DiscoError.
"""
import ctypes as C

import numpy as np
import torch

from . import _ffi
from .gamut import gamut_points


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if not t.is_cuda:
            raise _ffi.DiscoError("HIP path only: expected CUDA/HIP tensors (no CPU fallback)")


def tensor2array(tensors):
    arrays = tensors.detach().to("cpu").numpy()
    return np.transpose(arrays, (0, 2, 3, 1))


def _pool(input, prob, sp, want_sizes):
    _need_cuda(input, prob)
    x, p = input.contiguous().float(), prob.contiguous().float()
    n, c, hh, ww = x.shape
    if p.shape != (n, 9, hh, ww):
        raise ValueError("prob must be (N,9,H,W) matching input")
    if hh % sp or ww % sp:
        raise ValueError("H and W must be multiples of the superpixel size")
    h, w = hh // sp, ww // sp
    with torch.cuda.device(x.device):
        pooled = torch.empty(n, c, h, w, device=x.device)
        conf = torch.empty(n, 1, h, w, device=x.device)
        sizes = torch.empty(n, 1, h, w, device=x.device) if want_sizes else None
        ws = torch.empty(n * h * w * 9 * (c + 2) * 4, dtype=torch.uint8, device=x.device)
        _ffi.check(_ffi.lib().disco_op_poolfeat(_ffi.ptr(x), _ffi.ptr(p), _ffi.ptr(pooled), _ffi.ptr(conf), _ffi.ptr(sizes),
                                                n, c, hh, ww, sp, _ffi.ptr(ws), ws.numel(), _stream()))
    return pooled, conf, sizes


def poolfeat(input, prob, sp_h=2, sp_w=2, need_entry_prob=False):
    if sp_h != sp_w:
        raise NotImplementedError("square superpixels only (the reference always passes sp_h == sp_w)")
    pooled, conf, _ = _pool(input, prob, sp_h, False)
    return (pooled, conf) if need_entry_prob else pooled


def get_spixel_size(affinity_map, sp_h=2, sp_w=2, elem_thres=25):
    if sp_h != sp_w:
        raise NotImplementedError("square superpixels only")
    ones = torch.ones(affinity_map.shape[0], 1, *affinity_map.shape[2:], device=affinity_map.device)
    return _pool(ones, affinity_map, sp_h, True)[2]


def upfeat(input, prob, up_h=2, up_w=2):
    if up_h != up_w:
        raise NotImplementedError("square superpixels only")
    _need_cuda(input, prob)
    x, p = input.contiguous().float(), prob.contiguous().float()
    n, c, h, w = x.shape
    if p.shape != (n, 9, h * up_h, w * up_w):
        raise ValueError("prob must be (N,9,h*up,w*up)")
    with torch.cuda.device(x.device):
        out = torch.empty(n, c, h * up_h, w * up_w, device=x.device)
        _ffi.check(_ffi.lib().disco_op_upfeat(_ffi.ptr(x), _ffi.ptr(p), _ffi.ptr(out), n, c, h, w, up_h, _stream()))
    return out


class ColorLabel:
    """313-bin gamut labels; what inference needs: q_to_ab and decode_ind2ab (integer T = ranked bin, else annealed mean)."""

    def __init__(self, lambda_=0.5, device="cuda"):
        self.q_to_ab = torch.from_numpy(gamut_points()).to(device)

    def decode_ind2ab(self, batch_q, T=0.38):
        _need_cuda(batch_q)
        q = batch_q.contiguous().float()
        n, c, h, w = q.shape
        if c != 313:
            raise ValueError("expected 313 colour bins")
        with torch.cuda.device(q.device):
            ab = torch.empty(n, 2, h, w, device=q.device)
            if T % 1 == 0:      # the T-th most probable bin (basic.py:199-209)
                _ffi.check(_ffi.lib().disco_op_decode_ind2ab(_ffi.ptr(q), _ffi.ptr(ab), n, h * w, int(T), _stream()))
            else:               # annealed mean (basic.py:210-217)
                _ffi.check(_ffi.lib().disco_op_decode_annealed(_ffi.ptr(q), _ffi.ptr(ab), n, h * w, float(T), _stream()))
        return ab.type(batch_q.dtype)


def _color(fn_name, x):
    _need_cuda(x)
    t = x.contiguous().float()
    n, c, h, w = t.shape
    if c != 3:
        raise ValueError("expected (N,3,H,W)")
    with torch.cuda.device(t.device):
        out = torch.empty_like(t)
        _ffi.check(getattr(_ffi.lib(), fn_name)(_ffi.ptr(t), _ffi.ptr(out), n, h, w, _stream()))
    return out


def rgb2lab(rgb, l_mean=50, l_norm=50, ab_norm=110):
    if (l_mean, l_norm, ab_norm) != (50, 50, 110):
        raise NotImplementedError("only the reference's default normalisation (50, 50, 110)")
    return _color("disco_op_rgb2lab", rgb)


def lab2rgb(lab_rs, l_mean=50, l_norm=50, ab_norm=110):
    if (l_mean, l_norm, ab_norm) != (50, 50, 110):
        raise NotImplementedError("only the reference's default normalisation (50, 50, 110)")
    return _color("disco_op_lab2rgb", lab_rs)


def mark_color_hints(input_grays, target_ABs, gate_maps, kernel_size=3, base_ABs=None):
    """models/basic.py:95-117 as one fused kernel (the reference unfolds two dilations and runs six torch.where)."""
    _need_cuda(input_grays, target_ABs, gate_maps)
    g, t, m = (x.contiguous().float() for x in (input_grays, target_ABs, gate_maps))
    n, _, h, w = g.shape
    if g.shape[1] != 1 or t.shape != (n, 2, h, w) or m.shape != (n, 1, h, w):
        raise ValueError("expected gray (N,1,H,W), target_ABs (N,2,H,W), gate_maps (N,1,H,W)")
    b = None
    if base_ABs is not None:
        _need_cuda(base_ABs)
        b = base_ABs.contiguous().float()
        if b.shape != (n, 2, h, w):
            raise ValueError("base_ABs must be (N,2,H,W)")
    with torch.cuda.device(g.device):
        out = torch.empty(n, 3, h, w, device=g.device)
        _ffi.check(_ffi.lib().disco_op_mark_color_hints(_ffi.ptr(g), _ffi.ptr(t), _ffi.ptr(m), _ffi.ptr(b), _ffi.ptr(out), n, h, w,
                                                        int(kernel_size), _stream()))
    return out


def fetch_data_from_rgb8(rgb_img, org_size=True, device="cuda", psize=256, return_resized=False):
    """`fetch_data` (main/colorizer/inference.py:23-42; same default: org_size=True) minus the file decode: uint8 RGB (H,W,3) array
    or tensor -> (gray (1,1,Hp,Wp), ab (1,2,Hp,Wp), rgb (1,3,Hp,Wp), (H,W)) on the device; (H,W) is the ORIGINAL size in both
    branches, as in the reference (whose batch_depadding, :138-139, ignores it unless --no_resize).
    org_size=False (what inference.py's CLI passes unless --no_resize, :32-33, :101): cv2.resize to psize x psize (the reference
    hard-codes 256), INTER_LINEAR - the published 8-bit algorithm of opencv-python 4.6 (fixed-point coefficients; the area path
    for exact 2x downscales), fused with /255 + RGB->Lab + split in one kernel.
    org_size=True (--no_resize, :28-31): the reference's pad-to-16 quirk (both dims get `16 - dim % 16` when either is not
    a multiple of 16), one kernel: pad + /255 + RGB->Lab + split."""
    src = torch.as_tensor(rgb_img)
    if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
        raise ValueError("expected a uint8 RGB image (H,W,3)")
    src = src.to(device).contiguous()
    _need_cuda(src)
    H, W = int(src.shape[0]), int(src.shape[1])
    if not org_size:
        P = int(psize)
        with torch.cuda.device(src.device):
            gray = torch.empty(1, 1, P, P, device=src.device)
            ab = torch.empty(1, 2, P, P, device=src.device)
            rgb = torch.empty(1, 3, P, P, device=src.device)
            rs = torch.empty(P, P, 3, device=src.device, dtype=torch.uint8) if return_resized else None
            _ffi.check(_ffi.lib().disco_op_rgb8_resize_to_lab(_ffi.ptr(src), _ffi.ptr(rs), _ffi.ptr(gray), _ffi.ptr(ab), _ffi.ptr(rgb), 1, H, W,
                                                             P, P, _stream()))
        return (gray, ab, rgb, (H, W), rs) if return_resized else (gray, ab, rgb, (H, W))
    Hp, Wp = (H + 16 - H % 16, W + 16 - W % 16) if (H % 16 or W % 16) else (H, W)
    with torch.cuda.device(src.device):
        gray = torch.empty(1, 1, Hp, Wp, device=src.device)
        ab = torch.empty(1, 2, Hp, Wp, device=src.device)
        rgb = torch.empty(1, 3, Hp, Wp, device=src.device)
        _ffi.check(_ffi.lib().disco_op_rgb8_to_lab(_ffi.ptr(src), _ffi.ptr(gray), _ffi.ptr(ab), _ffi.ptr(rgb), 1, H, W, Hp, Wp, _stream()))
    return gray, ab, rgb, (H, W)


def normLabs_to_rgb8(lab_batch, H=None, W=None):
    """`save_normLabs_from_batch` (utils/util.py:91-106) minus the file encode, with `batch_depadding` folded in:
    normalised Lab (N,3,Hp,Wp) device tensor -> uint8 RGB (N,H,W,3) device tensor of the top-left H x W crop."""
    _need_cuda(lab_batch)
    lab = lab_batch.contiguous().float()
    n, c, hp, wp = lab.shape
    if c != 3:
        raise ValueError("expected (N,3,H,W)")
    H, W = (hp if H is None else int(H)), (wp if W is None else int(W))
    with torch.cuda.device(lab.device):
        out = torch.empty(n, H, W, 3, device=lab.device, dtype=torch.uint8)
        _ffi.check(_ffi.lib().disco_op_lab_to_rgb8(_ffi.ptr(lab), _ffi.ptr(out), n, hp, wp, H, W, _stream()))
    return out
