"""Drop-ins for the three conv networks of `models/network.py` as modules of their own, backed by libdisco_hip.so.

    SpixelNet(inChannel=1, outChannel=9, batchNorm=True)                 network.py:260-313   gray -> (N,9,H,W) softmax
    ColorProbNet(inChannel=1, outChannel=64)                             network.py:147-236   gray -> (N,64,H,W) features
    HourGlass2(inChannel=65, outChannel=2, resNum=3, normLayer=nn.BatchNorm2d)  network.py:125-144   (N,65,H,W) -> (N,2,H,W), no tanh

in the configurations `models/model.py:15,41,44` constructs them with (anything else raises NotImplementedError).  Each holds the
reference's `state_dict` of that network (keys without the colorizer's `segnet.net.` / `repnet.` / `enhanceNet.` prefix, strict
`load_state_dict`) and runs on a stand-alone context of the C ABI (`disco_options.network` = 1 / 2 / 3, `disco_forward_segnet` /
`_repnet` / `_enhance`, include/disco_hip.h): the same kernels, arithmetic and calibration as inside `model.AnchorColorProb`.
Inference only; CUDA/HIP tensors only (no CPU fallback: without the library this module raises).

HourGlass2 has no input of its own to measure activation ranges on, so its first forward calibrates the context on that batch
(`calibrate(x)` does it explicitly); the first `range_checks` forwards read the fp8 clamp counter like the colorizer's do.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _ffi
from .layout import state_dict_spec
from .model import _Node, _PARAM_KINDS, _PRECISIONS, default_precision


class _SubNet(nn.Module):
    _PREFIX = ""        # the network's keys inside the colorizer's state_dict
    _WHICH = 0          # disco_options.network
    _IN_CH = 1
    _OUT_CH = 0
    _ENTRY = ""

    def __init__(self, precision=None):
        super().__init__()
        self.precision = _PRECISIONS[precision or default_precision()]
        for key, shape, dt, kind in state_dict_spec():
            if not key.startswith(self._PREFIX):
                continue
            parts = key[len(self._PREFIX):].split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            t = torch.zeros(shape, dtype=getattr(torch, dt))
            if kind in _PARAM_KINDS:
                node.register_parameter(parts[-1], nn.Parameter(t, requires_grad=False))
            else:
                node.register_buffer(parts[-1], t)
        self._ctx, self._ctx_device, self._workspace = None, None, None

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the MI355X hot path is inference only")
        return super().train(False)

    def _drop_ctx(self):
        if getattr(self, "_ctx", None) is not None:
            _ffi.lib().disco_destroy(self._ctx)
        self._ctx = None

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        self._drop_ctx()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._drop_ctx()
        return out

    def __del__(self):
        try:
            self._drop_ctx()
        except Exception:
            pass

    def _context(self, dev):
        L = _ffi.lib()
        if self._ctx is None or self._ctx_device != dev:
            self._drop_ctx()
            opt = _ffi.Options(16, 1, 0, self.precision, self._WHICH)
            ctx = C.c_void_p()
            _ffi.check(L.disco_create(dev.index if dev.index is not None else torch.cuda.current_device(), C.byref(opt), C.byref(ctx)))
            try:
                for key, t in self.state_dict().items():
                    shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                    full = (self._PREFIX + key).encode()
                    if t.dtype == torch.float32:
                        h = t.detach().to("cpu").contiguous()
                        _ffi.check(L.disco_load_tensor(ctx, full, C.c_void_p(h.data_ptr()), shape, t.dim()))
                    else:
                        _ffi.check(L.disco_load_tensor(ctx, full, None, shape, t.dim()))
                _ffi.check(L.disco_finalize(ctx))
            except Exception:
                L.disco_destroy(ctx)
                raise
            self._ctx, self._ctx_device = ctx, dev
            self._fresh_context()
        return self._ctx

    def _fresh_context(self):
        pass

    def _check_input(self, x):
        if not x.is_cuda:
            raise _ffi.DiscoError("%s needs CUDA/HIP tensors: the HIP path has no CPU fallback" % type(self).__name__)
        if x.dim() != 4 or x.shape[1] != self._IN_CH or x.shape[2] % 16 or x.shape[3] % 16 or x.shape[0] < 1:
            raise ValueError("expected (N,%d,H,W) with H, W multiples of 16, got %s" % (self._IN_CH, tuple(x.shape)))
        return x.contiguous().float()

    def _run(self, x):
        dev = x.device
        n, _, H, W = x.shape
        L = _ffi.lib()
        need = C.c_size_t()
        _ffi.check(L.disco_workspace_bytes(self._ctx, n, H, W, 0, C.byref(need)))
        if self._workspace is None or self._workspace.numel() < need.value or self._workspace.device != dev:
            self._workspace = torch.empty(need.value, device=dev, dtype=torch.uint8)
        out = torch.empty(n, self._OUT_CH, H, W, device=dev, dtype=torch.float32)
        _ffi.check(getattr(L, self._ENTRY)(self._ctx, n, H, W, x.data_ptr(), out.data_ptr(), self._workspace.data_ptr(),
                                            self._workspace.numel(), torch.cuda.current_stream().cuda_stream))
        return out

    @torch.no_grad()
    def forward(self, x):
        x = self._check_input(x)
        with torch.cuda.device(x.device):
            self._context(x.device)
            return self._run(x)


class SpixelNet(_SubNet):
    """`models/network.py::SpixelNet` (:260-313): gray (N,1,H,W) -> softmax over the 9 neighbour slots (N,9,H,W).  94 tensors."""
    _PREFIX, _WHICH, _IN_CH, _OUT_CH, _ENTRY = "segnet.net.", 1, 1, 9, "disco_forward_segnet"

    def __init__(self, inChannel=1, outChannel=9, batchNorm=True, precision=None):
        if inChannel != 1 or outChannel != 9 or not batchNorm:
            raise NotImplementedError("SpixelNet(inChannel=1, outChannel=9, batchNorm=True) only (models/model.py:15)")
        super().__init__(precision)


class ColorProbNet(_SubNet):
    """`models/network.py::ColorProbNet` (:147-236): gray (N,1,H,W) in [-1,1] -> 64 feature channels at full resolution."""
    _PREFIX, _WHICH, _IN_CH, _OUT_CH, _ENTRY = "repnet.", 2, 1, 64, "disco_forward_repnet"

    def __init__(self, inChannel=1, outChannel=64, with_SA=False, precision=None):
        if inChannel != 1 or outChannel != 64 or with_SA:
            raise NotImplementedError("ColorProbNet(inChannel=1, outChannel=64) only (models/model.py:41)")
        super().__init__(precision)


class HourGlass2(_SubNet):
    """`models/network.py::HourGlass2` (:125-144): x (N,65,H,W) = cat(gray, 64 features) (model.py:196) -> (N,2,H,W); the colorizer
    applies tanh to it (model.py:197), this module does not.  The first forward calibrates the context on its batch."""
    _PREFIX, _WHICH, _IN_CH, _OUT_CH, _ENTRY = "enhanceNet.", 3, 65, 2, "disco_forward_enhance"

    def __init__(self, inChannel=65, outChannel=2, resNum=3, normLayer=nn.BatchNorm2d, precision=None):
        if inChannel != 65 or outChannel != 2 or resNum != 3 or normLayer is not nn.BatchNorm2d:
            raise NotImplementedError("HourGlass2(inChannel=64+1, outChannel=2, resNum=3, normLayer=nn.BatchNorm2d) only (models/model.py:44)")
        super().__init__(precision)
        self.range_checks = 3
        self._calibrated = False
        self._checks_left = 0

    def _fresh_context(self):
        self._calibrated = False
        self._checks_left = self.range_checks

    @torch.no_grad()
    def calibrate(self, x):
        """Measure (or widen) the activation ranges on a batch of this network's input, at most 64 images of it.  Blocking."""
        x = self._check_input(x)[:64].contiguous()
        with torch.cuda.device(x.device):
            self._context(x.device)
            torch.cuda.current_stream().synchronize()
            _ffi.check(_ffi.lib().disco_calibrate(self._ctx, x.data_ptr(), x.shape[0], x.shape[2], x.shape[3]))
            self._calibrated = True

    def enhance_arithmetic(self):
        """(precision name, channel disparity) the context runs on, as `model.AnchorColorProb.enhance_arithmetic` reports it."""
        if self._ctx is None:
            return None, 0.0
        prec, d, b = C.c_int(), C.c_float(), C.c_float()
        _ffi.check(_ffi.lib().disco_enhance_arithmetic(self._ctx, C.byref(prec), C.byref(d), C.byref(b)))
        return {v: k for k, v in _PRECISIONS.items()}.get(prec.value, str(prec.value)), float(d.value)

    @torch.no_grad()
    def forward(self, x):
        x = self._check_input(x)
        with torch.cuda.device(x.device):
            self._context(x.device)
            if not self._calibrated:
                self.calibrate(x)
            out = self._run(x)
            if self._checks_left > 0:
                self._checks_left -= 1
                cnt = C.c_uint64()
                _ffi.check(_ffi.lib().disco_saturation_count(self._ctx, _ffi.current_stream(), C.byref(cnt)))
                if cnt.value:
                    import warnings
                    warnings.warn("%d fp8 activation values were clamped: this input is outside the ranges the context was calibrated on; "
                                  "re-calibrating on this batch and running it again" % cnt.value)
                    self.calibrate(x)
                    out = self._run(x)
            return out
