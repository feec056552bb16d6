"""Drop-in for the reference's `models/model.py::AnchorColorProb` backed by libdisco_hip.so.

Same constructor, `forward(input_grays, input_colors, test_mode, sampled_T)` signature / 6-tuple of
fp32 NCHW outputs and the same 461-tensor `state_dict` (strict `load_state_dict`), so it slots in
under main/colorizer/inference.py:71-74,81,85,89,108-109.  Host code here is
plumbing only: tensor allocation, checkpoint hand-over to the C ABI and the host-side random draws
the reference makes (NumPy k-means initialisation, Python `random` hints, torch CPU fallback rows).
All arithmetic runs in hand-written HIP kernels; without the library this module raises.

Supported: every configuration main/colorizer/inference.py can produce (inference.py:71-74,156-165) -
enhanced=True, use_dense_pos=True, sp_size=16 (--psize; 8 and 32 run too, on the general pooling kernels), d_model=64, clustering or random hints, --diverse, --spix_pos,
--hint2regress - plus the validation forward of train_colorizer.py:206 (model.eval(), test_mode=False).
use_mask=True (model.py:38,121-125; no caller of the reference enables it) is supported with the semantics of torch >= 1.9, where the
float key_padding_mask the reference builds is ADDED to the attention scores (+1.0 at superpixels below 25 pixels; the pinned torch 1.8
rejects a float mask) - pinned on the live reference, tests/golden/fwd_usemask_*.npz; not with sampled_T > 0 (the reference fails there).
Not supported (NotImplementedError): test_mode=False together with hint2regress (model.py:178 reads an undefined name there),
training (set_train / gradients).
"""
import ctypes as C
import os
import random

import numpy as np
import torch
import torch.nn as nn

from . import _ffi
from .layout import state_dict_spec

_PARAM_KINDS = {"conv_w", "sn_w", "deconv_w", "bias", "bn_w", "bn_b", "lin_w", "lin_b", "ln_w", "ln_b"}
KMEANS_ITERS = 20  # clusterkit.py:43 iter_limit -> at most (K-1)*20 empty-cluster draws per image
# The conv kernel addresses an activation tensor (fp16 hi + lo planes, 64 channels at full resolution) through one
# 32-bit buffer descriptor, so a native call takes at most this many bytes per tensor; larger batches are split.
MAX_ACT_BYTES = (1 << 32) - (1 << 20)


# conv arithmetic per stack (include/disco_hip.h DISCO_PREC_*): "mx6" (default) = f16x3 on SpixelNet + ColorProbNet (the anchor-deciding
# stacks), f16+fp6x2 on HourGlass2; "mx8" = f16+fp8x2 on HourGlass2 (round 2's default); "x2q" = additionally the ColorProbNet on f16x2+fp8; "f16x3" everywhere; "mx8all":
# measurements only
_PRECISIONS = {"f16x3": _ffi.PREC_F16X3, "mx6": _ffi.PREC_MX6, "mx8": _ffi.PREC_MX8, "mx8all": _ffi.PREC_MX8_ALL, "x2q": _ffi.PREC_X2Q}
DEFAULT_PRECISION = "mx6"


def default_precision():
    """The precision a model gets when the caller names none: DEFAULT_PRECISION, or $DISCO_PRECISION (A/B runs of whole test suites)."""
    p = os.environ.get("DISCO_PRECISION", DEFAULT_PRECISION)
    if p not in _PRECISIONS:
        raise ValueError("DISCO_PRECISION=%r: choose from %s" % (p, sorted(_PRECISIONS)))
    return p


class _Node(nn.Module):
    """Bare container so that dotted checkpoint keys map onto a module tree."""


class SpixelSeg(nn.Module):
    """Drop-in for `models/model.py::SpixelSeg` (model.py:12-29): the superpixel network alone, as used by
    main/spixelseg/inference.py:45-89.  state_dict keys `net.*` (94 tensors); forward(gray) -> (N,9,H,W) affinity."""

    def __init__(self, inChannel=1, outChannel=9, batchNorm=True, precision=None):
        super().__init__()
        if inChannel != 1 or outChannel != 9 or not batchNorm:
            raise NotImplementedError("SpixelSeg(inChannel=1, outChannel=9, batchNorm=True) only")
        self.precision = _PRECISIONS[precision or default_precision()]
        for key, shape, dt, kind in state_dict_spec():
            if not key.startswith("segnet."):
                continue
            parts = key[len("segnet."):].split(".")
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

    def get_trainable_params(self, lr=1.0):
        raise NotImplementedError("training is outside the MI355X hot path")

    def _drop_ctx(self):
        if getattr(self, "_ctx", None) is not None and self.__dict__.get("_dp_origin") is None:     # (a replica shares its origin's context)
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

    def _replicate_for_data_parallel(self):
        """main/spixelseg/inference.py:50-51 wraps the model in nn.DataParallel on a multi-GPU host and calls it with batch 1: the one replica
        (on the module's own device) forwards on its origin's native context; see AnchorColorProb._replicate_for_data_parallel."""
        replica = super()._replicate_for_data_parallel()
        replica.__dict__["_dp_origin"] = self.__dict__.get("_dp_origin") or self
        return replica

    @torch.no_grad()
    def forward(self, input_grays):
        origin = self.__dict__.get("_dp_origin")
        if origin is not None:
            home = next(origin.parameters()).device
            if input_grays.device != home:
                raise NotImplementedError("nn.DataParallel scattered a batch onto %s, but the native context of SpixelSeg lives on %s: "
                                          "one process per GPU for batches over several GPUs (INTEGRATION.md section 5)" % (input_grays.device, home))
            return origin.forward(input_grays)
        if not input_grays.is_cuda:
            raise _ffi.DiscoError("SpixelSeg needs CUDA/HIP tensors: the HIP path has no CPU fallback")
        dev = input_grays.device
        gray = input_grays.contiguous().float()
        n, c, H, W = gray.shape
        if c != 1 or H % 16 or W % 16:
            raise ValueError("expected gray (N,1,H,W) with H, W multiples of 16")
        L = _ffi.lib()
        with torch.cuda.device(dev):
            if self._ctx is None or self._ctx_device != dev:
                self._drop_ctx()
                opt = _ffi.Options(16, 1, 0, self.precision, 1)
                ctx = C.c_void_p()
                _ffi.check(L.disco_create(dev.index if dev.index is not None else torch.cuda.current_device(),
                                          C.byref(opt), C.byref(ctx)))
                try:
                    for key, t in self.state_dict().items():
                        shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                        full = ("segnet." + key).encode()
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
            need = C.c_size_t()
            _ffi.check(L.disco_workspace_bytes(self._ctx, n, H, W, 0, C.byref(need)))
            if self._workspace is None or self._workspace.numel() < need.value or self._workspace.device != dev:
                self._workspace = torch.empty(need.value, device=dev, dtype=torch.uint8)
            aff = torch.empty(n, 9, H, W, device=dev, dtype=torch.float32)
            _ffi.check(L.disco_forward_segnet(self._ctx, n, H, W, gray.data_ptr(), aff.data_ptr(), self._workspace.data_ptr(),
                                              self._workspace.numel(), torch.cuda.current_stream().cuda_stream))
        return aff


class AnchorColorProb(nn.Module):
    def __init__(self, inChannel=1, outChannel=313, sp_size=16, d_model=64, use_dense_pos=True, spix_pos=False,
                 learning_pos=False, n_clusters=8, random_hint=False, hint2regress=False, enhanced=False,
                 use_mask=False, rank=0, precision=None, init_weights=True):
        super().__init__()
        unsupported = []
        if inChannel != 1: unsupported.append("inChannel=%r" % inChannel)
        if outChannel != 313: unsupported.append("outChannel=%r" % outChannel)
        if sp_size not in (8, 16, 32): unsupported.append("sp_size=%r (8, 16 or 32)" % sp_size)
        if d_model != 64: unsupported.append("d_model=%r" % d_model)
        if not use_dense_pos: unsupported.append("use_dense_pos=False")
        if not enhanced: unsupported.append("enhanced=False")
        if unsupported:
            raise NotImplementedError("outside the MI355X hot path (SURVEY §8b): " + ", ".join(unsupported))
        # learning_pos is accepted and ignored exactly like the reference (model.py:59 hard-codes is_learned=False)
        self.sp_size, self.hint_num, self.random_hint = sp_size, int(n_clusters), bool(random_hint)
        self.enhanced, self.hint2regress, self.spix_pos, self.use_token_mask = True, bool(hint2regress), bool(spix_pos), bool(use_mask)
        self.n_vocab = 313
        self.rank = rank
        self.precision = _PRECISIONS[precision or default_precision()]
        self.sync_kmeans_events = True   # emulate the reference's torch.randint fallback draws (one sync per forward)
        # The activation scales (one power-of-two exponent per tensor; the fp8 planes of the default precision have 14x headroom)
        # are fixed at load time on two SYNTHETIC images.  The first `range_checks` forwards of a context therefore read the clamp
        # counter (one host synchronisation each): if the caller's images clamped anything, the context is re-calibrated on those
        # very images (ranges only widen) and the batch is run again, with a warning.  0 switches the check off.
        self.range_checks = 3
        self._build_tree()
        self._ctx = None
        self._ctx_device = None
        self._workspace = None
        self._ws_need = {}
        self._keep = None
        self._range_checks_left = {}
        if init_weights:
            from .synth import synth_state_dict
            super().load_state_dict(synth_state_dict(130, hint2regress=self.hint2regress), strict=True)

    # ---- checkpoint layout ------------------------------------------------------------------------
    def _build_tree(self):
        for key, shape, dt, kind in state_dict_spec(self.hint2regress):
            parts = key.split(".")
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

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        self._drop_ctx()
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._drop_ctx()
        return out

    def _drop_ctx(self):
        # a DataParallel replica shares its origin's context (its __dict__ is a copy of the origin's, handle included): it must
        # never destroy it - replicas are rebuilt and garbage-collected on every DataParallel.forward
        if getattr(self, "_ctx", None) is not None and self.__dict__.get("_dp_origin") is None:
            _ffi.lib().disco_destroy(self._ctx)
        self._ctx = None

    def __del__(self):
        try:
            self._drop_ctx()
        except Exception:
            pass

    def set_train(self):
        raise NotImplementedError("training is outside the MI355X hot path")

    # ---- native context ---------------------------------------------------------------------------
    def _context(self, device):
        if self._ctx is not None and self._ctx_device == device:
            return self._ctx
        self._drop_ctx()
        L = _ffi.lib()
        opt = _ffi.Options(self.sp_size, self.hint_num, int(self.random_hint), self.precision, 0, int(self.hint2regress),
                           int(self.spix_pos), int(self.use_token_mask))
        ctx = C.c_void_p()
        _ffi.check(L.disco_create(device.index if device.index is not None else torch.cuda.current_device(),
                                  C.byref(opt), C.byref(ctx)))
        try:
            for key, t in self.state_dict().items():
                shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
                if t.dtype == torch.float32:
                    h = t.detach().to("cpu").contiguous()
                    _ffi.check(L.disco_load_tensor(ctx, key.encode(), C.c_void_p(h.data_ptr()), shape, t.dim()))
                else:  # num_batches_tracked: shape only
                    _ffi.check(L.disco_load_tensor(ctx, key.encode(), None, shape, t.dim()))
            _ffi.check(L.disco_finalize(ctx))
        except Exception:
            L.disco_destroy(ctx)
            raise
        self._ctx, self._ctx_device = ctx, device
        self._warn_fp8_fallback()
        return ctx

    def enhance_arithmetic(self):
        """(name, channel disparity): the arithmetic the HourGlass2 of the current context runs on - "mx6", or "mx8" when the channel-disparity
        guard of disco_finalize / disco_calibrate moved it to fp8 corrections (include/disco_hip.h: disco_enhance_arithmetic) - and the
        largest per-block spread of per-channel maxima the last calibration pass measured on its MX-fp6 tensors (after the channel
        equalisation, if one was applied: `equalised_from()` has the spread before it)."""
        if self._ctx is None:
            return None, 0.0
        prec, disp, before = C.c_int(), C.c_float(), C.c_float()
        _ffi.check(_ffi.lib().disco_enhance_arithmetic(self._ctx, C.byref(prec), C.byref(disp), C.byref(before)))
        self._eq_before = float(before.value)
        return {_ffi.PREC_MX6: "mx6", _ffi.PREC_MX8: "mx8", _ffi.PREC_F16X3: "f16x3"}.get(prec.value, str(prec.value)), float(disp.value)

    def equalised_from(self):
        """The channel disparity measured BEFORE the HourGlass2's channels were levelled at load time (0.0: no equalisation was needed)."""
        self.enhance_arithmetic()
        return getattr(self, "_eq_before", 0.0)

    def _warn_fp8_fallback(self):
        name, disp = self.enhance_arithmetic()
        if self.precision in (_ffi.PREC_MX6, _ffi.PREC_X2Q) and name == "mx8" and not getattr(self, "_fallback_warned", False):
            import warnings
            self._fallback_warned = True
            warnings.warn("this checkpoint spreads the channels of a HourGlass2 tensor over a factor %.0f inside one 32-channel block: the MX-fp6 "
                          "correction operands (one scale per block) would lose accuracy, so the HourGlass2 runs on fp8 corrections instead "
                          "(precision \"mx8\": same accuracy as on any checkpoint, 2-3 %% slower)" % disp)

    def calibrate(self, input_grays):
        """Widen the activation ranges (the per-tensor scale exponents) with those of the caller's own L images (N<=64,1,H,W);
        blocking.  The context is calibrated on two synthetic images at load time, and the first `range_checks` forwards
        re-calibrate by themselves when they clamp; call this up front with representative images to avoid that re-run."""
        g = input_grays.contiguous().float()
        if not g.is_cuda or g.dim() != 4 or g.shape[1] != 1:
            raise ValueError("expected a CUDA/HIP tensor (N,1,H,W)")
        with torch.cuda.device(g.device):
            ctx = self._context(g.device)
            _ffi.check(_ffi.lib().disco_calibrate(ctx, _ffi.ptr(g), g.shape[0], g.shape[2], g.shape[3]))
        # a calibration may rebuild the HourGlass2 on another arithmetic (channel levelling, the fp8 fallback): its activation planes then have
        # another size, so every cached workspace requirement is stale - and a workspace sized for the old plan would fail the very re-run
        # the automatic range check makes (DISCO_ENOMEM "workspace too small")
        self._ws_need = {}
        self._workspace = {}
        self._warn_fp8_fallback()

    def _read_clamp_counter(self):
        cnt = C.c_uint64(0)
        _ffi.check(_ffi.lib().disco_saturation_count(self._ctx, _ffi.current_stream(), C.byref(cnt)))
        return int(cnt.value)

    def saturation_count(self):
        """fp8 activation elements clamped since the previous call (one device synchronisation on the current stream).  The automatic
        range checks of the first forwards read - and thereby reset - the same device counter: a forward they found clamping is discarded,
        re-calibrated and run again, so its count is not owed to the caller."""
        if self._ctx is None:
            return 0
        return self._read_clamp_counter()

    def kmeans_fallback_count(self):
        """Images since the previous call whose k-means left the several-workgroup kernel (more than 512 tokens: the --no_resize sizes) for the
        one-workgroup kernel because their workgroups could not be resident together (disco_kmeans_fallback_count).  Results are identical
        either way; one device synchronisation on the current stream."""
        if self._ctx is None:
            return 0
        cnt = C.c_uint64(0)
        _ffi.check(_ffi.lib().disco_kmeans_fallback_count(self._ctx, _ffi.current_stream(), C.byref(cnt)))
        return int(cnt.value)

    def set_profiling(self, level=1):
        """0 off, 1 per-stage hipEvents, 2 additionally an event pair around every MFMA conv launch."""
        self._profiling = int(level)
        if self._ctx is not None:
            _ffi.lib().disco_set_profiling(self._ctx, int(level))

    def conv_profile(self):
        """(launches, total ms, total algorithmic FLOPs) of the conv3x3_mfma launches of the last forward."""
        n, ms, fl = C.c_int(), C.c_float(), C.c_double()
        _ffi.check(_ffi.lib().disco_profile_conv(self._ctx, C.byref(n), C.byref(ms), C.byref(fl)))
        return n.value, ms.value, fl.value

    def conv_profile_bytes(self):
        """Compulsory HBM bytes of the conv3x3_mfma launches of the last forward (activations once in, once out)."""
        b = C.c_double()
        _ffi.check(_ffi.lib().disco_profile_conv_bytes(self._ctx, C.byref(b)))
        return b.value

    def conv_profile_entries(self):
        """[(layer key, ms, algorithmic FLOPs)] per MFMA conv launch of the last forward (profiling level 2)."""
        L = _ffi.lib()
        out, i = [], 0
        while True:
            key, ms, fl = C.c_char_p(), C.c_float(), C.c_double()
            if L.disco_profile_conv_entry(self._ctx, i, C.byref(key), C.byref(ms), C.byref(fl)) != 0:
                return out
            out.append((key.value.decode(), ms.value, fl.value))
            i += 1

    def profile(self):
        """[(stage, ms, algorithmic flops)] of the last forward (after a device sync)."""
        L = _ffi.lib()
        out = []
        for i in range(L.disco_profile_count(self._ctx)):
            name, ms, fl = C.c_char_p(), C.c_float(), C.c_double()
            _ffi.check(L.disco_profile_entry(self._ctx, i, C.byref(name), C.byref(ms), C.byref(fl)))
            out.append((name.value.decode(), ms.value, fl.value))
        return out

    # ---- host-side random draws (same generators the reference consumes) ---------------------------
    def _kmeans_init(self, n, l):
        # clusterkit.py:107 — np.random.choice per image, legacy global RandomState, image order
        return np.ascontiguousarray(
            np.stack([np.random.choice(l, self.hint_num, replace=False) for _ in range(n)]).astype(np.int32))

    def _random_hints(self, n, l):
        # basic.py:42-47 — Python's global `random`
        k = self.hint_num
        return np.ascontiguousarray(
            np.stack([np.asarray(random.sample(range(0, l), random.randint(k, k))) for _ in range(n)]).astype(np.int32))

    @staticmethod
    def _peek_randint(l, count):
        """The next `count` values torch.randint(l,(1,)) would return, without consuming them."""
        g = torch.Generator()
        g.set_state(torch.get_rng_state())
        # one vectorised draw yields the same values as `count` successive randint(l,(1,)) calls on the CPU
        # generator (checked in tests/test_abi_cpu.py::test_peek_randint_matches_sequential_draws)
        return torch.randint(l, (count,), generator=g).tolist()

    # ---- forward ----------------------------------------------------------------------------------
    def forward(self, input_grays, input_colors, test_mode=False, sampled_T=0):
        origin = self._replica_origin(input_grays)
        if origin is not None:
            return origin.forward_with_draws(input_grays, input_colors, test_mode, sampled_T)
        return self.forward_with_draws(input_grays, input_colors, test_mode, sampled_T)

    def train(self, mode=True):
        """eval() is a no-op like on any frozen module; training is outside the hot path."""
        if mode:
            raise NotImplementedError("the MI355X hot path is inference only: AnchorColorProb.train() is not available")
        return super().train(False)

    def _replicate_for_data_parallel(self):
        """main/colorizer/inference.py:76-82 wraps the model in nn.DataParallel whenever the host shows more than one GPU - always, on an
        8 x MI355X node - and calls it with batch 1 (:93,108-109): DataParallel.scatter then yields ONE chunk and replicates the module
        onto device_ids[:1] only, the device the module lives on.  A replica is a copy of __dict__ with empty `_parameters`, so it
        could never rebuild a native context of its own; instead it keeps a reference to its origin and forwards on the ORIGIN's context
        (and the origin's workspace and range-check bookkeeping).  A replica that is actually CALLED on another device - batch > 1 on a
        multi-GPU host - raises and names the supported way (_replica_origin)."""
        replica = super()._replicate_for_data_parallel()
        replica.__dict__["_dp_origin"] = self.__dict__.get("_dp_origin") or self
        return replica

    def _replica_origin(self, like):
        """None for an ordinary module; for a DataParallel replica: the module it was replicated from, after checking that the call is on
        the device that module (and its native context) lives on."""
        origin = self.__dict__.get("_dp_origin")
        if origin is None:
            return None
        home = next(origin.parameters()).device
        if like.device != home:
            raise NotImplementedError(
                "nn.DataParallel scattered a batch onto %s, but the native context of AnchorColorProb lives on %s (its replicas carry no "
                "parameters and share that one context). Batches over several GPUs: one process per GPU with "
                "disentangledcolorization_amd.runner.ShardedColorizer (INTEGRATION.md section 5); batch 1 - the reference's inference.py - "
                "works under DataParallel as it is." % (like.device, home))
        return origin

    def _check_inputs(self, input_grays, input_colors, test_mode, sampled_T=0):
        test_mode = bool(test_mode)
        if not test_mode and self.hint2regress:
            raise NotImplementedError("hint2regress has no test_mode=False forward: models/model.py:178 raises NameError")
        if self.use_token_mask and test_mode and int(sampled_T) > 0:
            raise NotImplementedError("use_mask has no --diverse forward: the reference's key_padding_mask keeps batch 1 (models/model.py:154-159,186)")
        if not input_grays.is_cuda:
            raise _ffi.DiscoError("AnchorColorProb needs CUDA/HIP tensors: the HIP path has no CPU fallback")
        dev = input_grays.device
        gray = input_grays.contiguous().float()
        ab = input_colors.to(dev).contiguous().float()
        n, _, H, W = gray.shape
        if gray.shape[1] != 1 or ab.shape != (n, 2, H, W):
            raise ValueError("expected gray (N,1,H,W) and ab (N,2,H,W)")
        mult = max(16, self.sp_size)       # whole superpixel cells (--psize) and the conv stacks' four stride-2 stages
        if H % mult or W % mult:
            raise ValueError("H and W must be multiples of %d" % mult)
        return test_mode, gray, ab

    def set_progress_event(self, event, after_conv_launches):
        """The next forward_once on the current device records `event` (a torch.cuda.Event that has been recorded at least once, so
        that its handle exists) on its stream behind its `after_conv_launches`-th MFMA conv launch (disco_set_progress_event):
        runner.py staggers its micro-batches with it."""
        dev = torch.device("cuda", torch.cuda.current_device())
        handle = C.c_void_p(None if event is None else event.cuda_event)          # None cancels an armed event
        _ffi.check(_ffi.lib().disco_set_progress_event(self._context(dev), handle, int(after_conv_launches)))

    def max_fallback(self):
        """Upper bound of empty-cluster draws one image can consume (clusterkit.py:176-182: K-1 per pass, 20 passes)."""
        return KMEANS_ITERS * self.hint_num

    @torch.no_grad()
    def forward_once(self, input_grays, input_colors, test_mode=True, sampled_T=0, init_idx=None, hint_pos=None,
                     fallback_stream=None, fallback_bases=None, want_events=True, out=None, range_check=True):
        """ONE native forward with every host-side draw supplied by the caller; consumes no generator state.
        init_idx (n,K) k-means rows / hint_pos (n,K) random-hint tokens; fallback_stream: the values successive
        torch.randint(L,(1,)) calls would return (a prefix of the reference's global draw stream), fallback_bases (n,):
        where in that stream image i's empty-cluster draws start.  Returns (6-tuple, events) with events (n,) int32 =
        draws each image consumed (None when want_events is False: no host synchronisation then - EXCEPT in the first
        `range_checks` (3) forwards of a context, each of which reads the clamp counter once (one synchronisation of the current stream)
        and may re-calibrate and re-run the batch with a warning; set range_checks = 0 before the first forward for a strictly
        asynchronous start, e.g. on ranks whose inputs are known to lie in the calibrated range).
        out: optional preallocated (pal, ref, pred, affinity, spix, mask) tensors to write into (sampled_T = 0 only; runner.py hands
        slices of the whole batch's outputs to its micro-batches instead of concatenating their results).
        range_check=False: this call neither reads the clamp counter nor counts as one of the first forwards (runner.py under a
        process group: the ranks check and re-calibrate TOGETHER, ShardedColorizer._collective_range_check)."""
        origin = self._replica_origin(input_grays)
        if origin is not None:
            return origin.forward_once(input_grays, input_colors, test_mode, sampled_T, init_idx, hint_pos, fallback_stream, fallback_bases,
                                       want_events, out, range_check)
        test_mode, gray, ab = self._check_inputs(input_grays, input_colors, test_mode, sampled_T)
        dev = gray.device
        n, _, H, W = gray.shape
        sp = self.sp_size
        h, w = H // sp, W // sp
        l = h * w
        T = int(sampled_T) if test_mode else 0      # the validation forward ignores sampled_T (model.py:169-171)
        rep = 3 if T > 0 else 1
        K = self.hint_num
        if self.random_hint:
            if hint_pos is None or np.shape(hint_pos) != (n, K):
                raise ValueError("hint_pos must be (n, n_clusters) = (%d, %d), got %s" % (n, K, np.shape(hint_pos)))
            hint_pos = np.ascontiguousarray(hint_pos, dtype=np.int32)
        else:
            if init_idx is None or np.shape(init_idx) != (n, K):
                raise ValueError("init_idx must be (n, n_clusters) = (%d, %d), got %s" % (n, K, np.shape(init_idx)))
            init_idx = np.ascontiguousarray(init_idx, dtype=np.int32)
        max_imgs = max(1, MAX_ACT_BYTES // (64 * H * W * 4 * rep))
        if n > max_imgs:
            # images are independent: run the batch in slices and concatenate.  BALANCED slices (512 images -> 171 + 171 + 170, not
            # 255 + 255 + 2): a two-image tail forward would run on the small-batch paths at a fraction of the throughput
            parts, evs = [], []
            n_slices = -(-n // max_imgs)
            bounds = [(k * n) // n_slices for k in range(n_slices + 1)]
            for i, j in zip(bounds[:-1], bounds[1:]):
                o, e = self.forward_once(gray[i:j], ab[i:j], test_mode, sampled_T, None if init_idx is None else init_idx[i:j],
                                         None if hint_pos is None else hint_pos[i:j], fallback_stream,
                                         None if fallback_bases is None else fallback_bases[i:j], want_events,
                                         None if out is None else tuple(t[i:j] for t in out), range_check)
                parts.append(o); evs.append(e)
            outs = out if out is not None else tuple(torch.cat([p[k] for p in parts], 0) for k in range(6))
            return outs, (np.concatenate(evs) if want_events and not self.random_hint else (np.zeros(n, np.int32) if want_events else None))
        n2 = n * rep
        L = _ffi.lib()
        events = None
        with torch.cuda.device(dev):
            ctx = self._context(dev)
            L.disco_set_profiling(ctx, int(getattr(self, "_profiling", 0)))
            f32 = dict(device=dev, dtype=torch.float32)
            shapes = ((n, 313, h, w), (n2, 2 if self.hint2regress else 313, h, w), (n2, 2, H, W), (n, 9, H, W), (n2, 2, h, w), (n, 1, h, w))
            if out is not None:
                if rep != 1 or len(out) != 6 or any(tuple(t.shape) != sh or t.dtype != torch.float32 or t.device != dev or not t.is_contiguous()
                                                    for t, sh in zip(out, shapes)):
                    raise ValueError("out: six contiguous fp32 tensors of shapes %s on %s (sampled_T = 0 only)" % (shapes, dev))
                pal, ref, pred, aff, spix, mask = out
            else:
                pal, ref, pred, aff, spix, mask = (torch.empty(sh, **f32) for sh in shapes)
            ws_key = (n, H, W, T > 0)
            if ws_key not in self._ws_need:
                need = C.c_size_t()
                _ffi.check(L.disco_workspace_bytes(ctx, n, H, W, T, C.byref(need)))
                self._ws_need[ws_key] = need.value
            need_bytes = self._ws_need[ws_key]
            # one workspace per stream: forwards issued on different streams may overlap on the GPU (runner.py pipelines
            # micro-batches that way), each needs its own activations
            stream_ptr = torch.cuda.current_stream().cuda_stream
            if not isinstance(self._workspace, dict):
                self._workspace = {}
            wsb = self._workspace.get(stream_ptr)
            if wsb is None or wsb.numel() < need_bytes or wsb.device != dev:
                self._workspace[stream_ptr] = None
                wsb = self._workspace[stream_ptr] = torch.empty(need_bytes, device=dev, dtype=torch.uint8)
            a = _ffi.ForwardArgs()
            a.n, a.h, a.w, a.sampled_T, a.test_mode = n, H, W, T, int(test_mode)
            a.d_gray, a.d_ab = gray.data_ptr(), ab.data_ptr()
            a.d_pal_logit, a.d_ref_logit, a.d_pred_colors = pal.data_ptr(), ref.data_ptr(), pred.data_ptr()
            a.d_affinity, a.d_spix_colors, a.d_hint_mask = aff.data_ptr(), spix.data_ptr(), mask.data_ptr()
            a.d_workspace, a.workspace_bytes = wsb.data_ptr(), wsb.numel()
            a.stream = stream_ptr
            if self.random_hint:
                a.h_hint_pos = hint_pos.ctypes.data
                _ffi.check(L.disco_forward(ctx, C.byref(a)))
                self._keep = (hint_pos,)
                events = np.zeros(n, np.int32) if want_events else None
            else:
                MF = self.max_fallback()
                a.h_init_idx = init_idx.ctypes.data
                a.max_fallback = MF
                bases = np.zeros(n, np.int64) if fallback_bases is None else np.asarray(fallback_bases, dtype=np.int64)
                if bases.shape != (n,):
                    raise ValueError("fallback_bases must be (n,)")
                stream_arr = np.asarray(fallback_stream if fallback_stream is not None else self._peek_randint(l, MF), dtype=np.int32)
                if len(stream_arr) < int(bases.max()) + MF:
                    raise ValueError("fallback_stream holds %d draws, image bases need %d" % (len(stream_arr), int(bases.max()) + MF))
                rows = np.ascontiguousarray(stream_arr[bases[:, None] + np.arange(MF)[None, :]])
                a.h_fallback_rows = rows.ctypes.data
                events = np.zeros(n, np.int32) if want_events else None
                a.h_kmeans_events = events.ctypes.data if want_events else None
                _ffi.check(L.disco_forward(ctx, C.byref(a)))
                if want_events and int(events.max()) > MF:
                    raise _ffi.DiscoError("k-means used more than %d empty-cluster draws" % MF)
                self._keep = (init_idx, rows, events)
            left = self._range_checks_left.get(id(ctx), self.range_checks) if range_check else 0
            if left > 0:
                self._range_checks_left = {id(ctx): left - 1}
                clamped = self._read_clamp_counter()
                if clamped:
                    import warnings
                    warnings.warn("%d fp8 activation values were clamped: this input is outside the ranges the context was calibrated on "
                                  "(two synthetic images at load time); re-calibrating on this batch and running it again" % clamped)
                    self.calibrate(gray[:64])       # (the clamped run is discarded: its count is not owed to saturation_count())
                    return self.forward_once(gray, ab, test_mode, sampled_T, init_idx, hint_pos, fallback_stream, fallback_bases, want_events, out)
        if rep > 1:
            aff_out = aff.expand(rep, -1, -1, -1) if n == 1 else aff.repeat_interleave(rep, 0)
            mask_out = mask.expand(rep, -1, -1, -1) if n == 1 else mask.repeat_interleave(rep, 0)
        else:
            aff_out, mask_out = aff, mask
        return (pal, ref, pred, aff_out, spix, mask_out), events

    @torch.no_grad()
    def forward_with_draws(self, input_grays, input_colors, test_mode=True, sampled_T=0, init_idx=None, hint_pos=None):
        """The reference's single-process semantics: host-side draws come from the global generators the reference
        consumes (NumPy legacy RandomState for the k-means rows, Python `random` for random hints, torch's CPU generator
        for empty-cluster fallbacks, clusterkit.py:181-182) unless supplied, and torch's generator advances by exactly the
        number of fallback draws the reference would have made, in image order.  With sync_kmeans_events = False there is
        no host synchronisation: every image reads fallback rows from the start of the stream and nothing is consumed -
        exact only while no empty-cluster event occurs (check `last_kmeans_events()` / bench.py's `kmeans_events`)."""
        origin = self._replica_origin(input_grays)
        if origin is not None:
            return origin.forward_with_draws(input_grays, input_colors, test_mode, sampled_T, init_idx, hint_pos)
        test_mode, gray, ab = self._check_inputs(input_grays, input_colors, test_mode, sampled_T)
        n, _, H, W = gray.shape
        l = (H // self.sp_size) * (W // self.sp_size)
        if self.random_hint:
            hint_pos = self._random_hints(n, l) if hint_pos is None else hint_pos
            return self.forward_once(gray, ab, test_mode, sampled_T, None, hint_pos, want_events=False)[0]
        init_idx = self._kmeans_init(n, l) if init_idx is None else init_idx
        MF = self.max_fallback()
        if not self.sync_kmeans_events:
            return self.forward_once(gray, ab, test_mode, sampled_T, init_idx, None, self._peek_randint(l, MF), None, want_events=False)[0]
        stream = np.asarray(self._peek_randint(l, MF * 2), dtype=np.int32)
        bases = np.zeros(n, np.int64)
        for _ in range(n + 1):
            while len(stream) < int(bases.max()) + MF:
                stream = np.asarray(self._peek_randint(l, len(stream) * 2), dtype=np.int32)
            out, events = self.forward_once(gray, ab, test_mode, sampled_T, init_idx, None, stream, bases)
            new_bases = np.concatenate(([0], np.cumsum(events)[:-1])).astype(np.int64)
            # an image's rows matter only if it drew any: redo when an image that consumed draws started at the wrong place
            if not np.any((events > 0) & (new_bases != bases)):
                break
            bases = new_bases
        for _ in range(int(events.sum())):      # consume what the reference would have consumed
            torch.randint(l, (1,))
        self._last_events = events
        return out

    def last_kmeans_events(self):
        """Per-image empty-cluster draws of the latest synchronised forward (None if there was none)."""
        return getattr(self, "_last_events", None)

