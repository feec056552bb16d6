"""-m gpu: the fp16 + fp8-correction conv kernel (csrc/conv_mx.hip) against torch (float64 reference of the same op).

Tolerance: the two fp8 correction products leave a relative error of ~2^-16 of the accumulated |w||a| mass per output
(tools/precision_sim.py); the checks below allow 1.5e-4 of the output range - two orders of magnitude tighter than plain
fp16 operands (2e-2 in test_gpu_ops.py) and what the end-to-end 1e-3 bar on ab needs with margin."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from disentangledcolorization_amd import _ffi  # noqa: E402

LO, Q = _ffi.PLANE_LO, _ffi.PLANE_Q
TOL = 1.5e-4


@pytest.fixture(scope="module")
def H():
    import gpu_helpers
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _ffi.lib()
    return gpu_helpers


def g(seed):
    return torch.Generator().manual_seed(seed)


def _ref(x, w, b, stride, act, slope, bn, res):
    y = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
    if res is not None: y = y + res.double()
    if act == _ffi.ACT_RELU: y = F.relu(y)
    elif act == _ffi.ACT_LRELU: y = F.leaky_relu(y, slope)
    elif act == _ffi.ACT_TANH: y = torch.tanh(y)
    if bn is not None: y = y * bn[0].double()[None, :, None, None] + bn[1].double()[None, :, None, None]
    return y


def test_mx_layout_roundtrip(H):
    """hi+lo carries ~22 bits; the a8 plane is x to 4 significant bits; hi + al8 is x to ~2^-15 relative."""
    x = torch.randn(2, 64, 9, 11, generator=g(0)) * 3
    a = H.to_act_mx(x, LO | Q)
    assert H.max_err(a.read(0).cpu(), x) < 2e-7 * 16
    a8 = a.read(1).cpu()
    big = x.abs() > x.abs().max() * 2 ** -9                        # scaled values in the NORMAL fp8 range (>= 2^-6)
    assert ((a8 - x).abs()[big] <= x.abs()[big] * 2 ** -4 + 1e-30).all()
    assert H.max_err(a.read(2).cpu(), x) < x.abs().max().item() * 2 ** -15


MX_CASES = [
    # cin, cout, h, w, stride, act, slope, bn, res, out_planes
    (64, 64, 32, 32, 1, _ffi.ACT_LRELU, 0.2, True, False, Q),
    (32, 64, 24, 40, 1, _ffi.ACT_RELU, 0.0, True, False, LO | Q),
    (64, 64, 16, 16, 1, _ffi.ACT_NONE, 0.0, False, True, LO),
    (64, 128, 32, 64, 2, _ffi.ACT_LRELU, 0.2, True, False, Q),
    (128, 256, 16, 16, 2, _ffi.ACT_RELU, 0.0, False, False, Q),
    (256, 32, 8, 8, 1, _ffi.ACT_RELU, 0.0, False, True, LO | Q),
    (96, 64, 17, 33, 1, _ffi.ACT_TANH, 0.0, False, False, LO),     # ragged: partial tiles, odd sizes
    (512, 512, 8, 8, 1, _ffi.ACT_LRELU, 0.2, True, False, Q),
    (64, 64, 64, 64, 1, _ffi.ACT_RELU, 0.0, False, False, Q),      # the 16x32-pixel tile with row reuse
    (256, 256, 32, 32, 1, _ffi.ACT_RELU, 0.0, False, False, Q),
]


@pytest.mark.parametrize("case", MX_CASES)
def test_conv3x3_mx_matches_torch(H, case):
    cin, cout, h, w, stride, act, slope, use_bn, use_res, planes = case
    gen = g(cin * 1000 + cout + h)
    n = 3
    x = torch.randn(n, cin, h, w, generator=gen)
    if cin >= 128: x = F.relu(x)                                   # post-ReLU-like inputs for the deep layers
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    bn = (torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.1) if use_bn else None
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = torch.randn(n, cout, ho, wo, generator=gen) if use_res else None
    want = _ref(x, wt, b, stride, act, slope, bn, res)
    scale = max(1.0, want.abs().max().item())
    osexp = H.sexp_for(want)
    out, sat = H.conv3x3_mx(H.to_act_mx(x), wt, b, stride=stride, act=act, slope=slope, bn_scale=bn[0] if bn else None,
                            bn_shift=bn[1] if bn else None, res=H.to_act_mx(res, LO) if use_res else None, out_planes=planes,
                            out_sexp=osexp)
    assert sat == 0
    if planes & LO:
        assert H.max_err(out.read(0), want) < TOL * scale
    if planes & Q:
        # hi + dequantised al8 reproduces the result to ~2^-15; a8 is the result to 4 significant bits
        assert H.max_err(out.read(2), want) < (TOL + 2 ** -14) * scale
        a8 = out.read(1).cpu().double()
        big = want.abs() > scale * 2 ** -9
        assert ((a8 - want).abs()[big] <= want.abs()[big] * 2 ** -3.9 + TOL * scale).all()
    # fp32 NCHW output of the same layer
    out32, _ = H.conv3x3_mx(H.to_act_mx(x), wt, b, stride=stride, act=act, slope=slope, bn_scale=bn[0] if bn else None,
                            bn_shift=bn[1] if bn else None, out_f32=True) if not use_res else (None, 0)
    if out32 is not None:
        assert H.max_err(out32, want) < TOL * scale


X2Q_CASES = [
    # cin, cout, h, w, stride, act, slope, bn, res, out_planes  (QL: al8-only q planes, what the next x2q layer reads)
    (64, 64, 32, 32, 1, _ffi.ACT_LRELU, 0.2, True, False, _ffi.PLANE_QL),
    (128, 64, 24, 40, 1, _ffi.ACT_RELU, 0.0, True, False, LO | _ffi.PLANE_QL),
    (64, 128, 32, 64, 2, _ffi.ACT_LRELU, 0.2, True, False, _ffi.PLANE_QL),
    (256, 512, 16, 16, 2, _ffi.ACT_RELU, 0.0, False, False, _ffi.PLANE_QL),
    (192, 64, 17, 33, 1, _ffi.ACT_NONE, 0.0, False, True, LO),       # ragged, 3 groups of 64, residual
    (512, 512, 8, 8, 1, _ffi.ACT_LRELU, 0.2, True, False, _ffi.PLANE_QL),
    (64, 64, 64, 64, 1, _ffi.ACT_RELU, 0.0, False, False, LO),       # the 16x32-pixel tile with row reuse
    (256, 256, 32, 32, 1, _ffi.ACT_RELU, 0.0, False, False, _ffi.PLANE_QL),
]


@pytest.mark.parametrize("case", X2Q_CASES)
def test_conv3x3_x2q_matches_torch(H, case):
    """The kernel's second arithmetic (w_h a_h + w_l a_h in fp16, fp8(w) fp8(a_l)): only the activation residual is
    quantised, so the error is ~2^-16 of the |w||a_l 2^11| mass - the check is 4x tighter than for the mx arithmetic."""
    cin, cout, h, w, stride, act, slope, use_bn, use_res, planes = case
    QL = _ffi.PLANE_QL
    gen = g(cin * 1000 + cout + h + 7)
    n = 3
    x = torch.randn(n, cin, h, w, generator=gen)
    if cin >= 128: x = F.relu(x)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    bn = (torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.1) if use_bn else None
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = torch.randn(n, cout, ho, wo, generator=gen) if use_res else None
    want = _ref(x, wt, b, stride, act, slope, bn, res)
    scale = max(1.0, want.abs().max().item())
    src = H.to_act_mx(x, QL)
    assert H.max_err(src.read(2).cpu(), x) < x.abs().max().item() * 2 ** -15          # hi + al8 view of the al8-only layout
    kw = dict(stride=stride, act=act, slope=slope, bn_scale=bn[0] if bn else None, bn_shift=bn[1] if bn else None, x2q=True)
    out, sat = H.conv3x3_mx(src, wt, b, res=H.to_act_mx(res, LO) if use_res else None, out_planes=planes, out_sexp=H.sexp_for(want), **kw)
    assert sat == 0
    if planes & LO:
        assert H.max_err(out.read(0), want) < TOL / 4 * scale
    if planes & QL:
        assert H.max_err(out.read(2), want) < (TOL / 4 + 2 ** -14) * scale
    if not use_res:
        out32, _ = H.conv3x3_mx(src, wt, b, out_f32=True, **kw)
        assert H.max_err(out32, want) < TOL / 4 * scale
    # and it is closer to the exact result than the f16 + fp8x2 arithmetic on the same data
    if not use_res:
        mx32, _ = H.conv3x3_mx(H.to_act_mx(x), wt, b, out_f32=True, **{**kw, "x2q": False})
        assert H.max_err(out32, want) <= H.max_err(mx32, want)


def test_conv3x3_mx_upsample_and_concat_on_read(H):
    gen = g(5)
    a = torch.randn(2, 64, 12, 20, generator=gen)      # half-resolution source, nearest x2 on read
    s = torch.randn(2, 32, 24, 40, generator=gen) * 4  # full-resolution skip with its own scale
    wt = torch.randn(64, 96, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * 96))
    b = torch.randn(64, generator=gen) * 0.1
    up = a.repeat_interleave(2, 2).repeat_interleave(2, 3)
    want = _ref(torch.cat((up, s), 1), wt, b, 1, _ffi.ACT_RELU, 0.0, None, None)
    # tensors that are concatenated on read share ONE scale exponent (the forward ties skip connections at calibration)
    e = min(H.sexp_for(a), H.sexp_for(s))
    out, sat = H.conv3x3_mx(H.to_act_mx(a, sexp=e), wt, b, src1=H.to_act_mx(s, sexp=e), up0=True, act=_ffi.ACT_RELU, out_planes=LO)
    assert sat == 0 and H.max_err(out.read(0), want) < TOL * max(1.0, want.abs().max().item())
    with pytest.raises(_ffi.DiscoError):          # different exponents per source: refused, not silently mis-scaled
        H.conv3x3_mx(H.to_act_mx(a, sexp=e), wt, b, src1=H.to_act_mx(s, sexp=e + 1), up0=True, act=_ffi.ACT_RELU, out_planes=LO)


def test_conv3x3_mx_saturation_counter(H):
    """An output scale that is far too large must be reported, not silently clamped."""
    gen = g(9)
    x = torch.randn(1, 32, 16, 16, generator=gen)
    wt = torch.randn(32, 32, 3, 3, generator=gen) * 0.1
    out, sat = H.conv3x3_mx(H.to_act_mx(x), wt, torch.zeros(32), out_planes=Q, out_sexp=12)
    assert sat > 0


def test_conv3x3_mx_rejects_bad_shapes(H):
    x = torch.randn(1, 48, 8, 8)
    with pytest.raises(_ffi.DiscoError):
        H.conv3x3_mx(H.to_act_mx(x, c_pad=64), torch.randn(16, 48, 3, 3), torch.zeros(16), out_planes=Q)   # 16 output channels cannot carry q planes


@pytest.mark.parametrize("planes", [Q, LO | Q])
def test_conv3x3_mx_depth_to_space(H, planes):
    """The depth-to-space epilogue of the fp16+fp8 arithmetic (HourGlass2's sub-pixel up-convs: `up2.conv1`, `up1.conv1`): 4 C
    phase-major output channels of the low-resolution conv land in an (n, C, 2h, 2w) activation, hi (+ lo) and q planes."""
    gen = g(21)
    n, cin, C, h, w = 2, 64, 32, 13, 19                     # ragged: partial tiles
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(4 * C, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(C, generator=gen) * 0.1                 # the kernel indexes its parameters modulo the phase size
    y = F.relu(F.conv2d(x.double(), wt.double(), b.double().repeat(4), padding=1))          # (n, 4C, h, w), channel = ph*C + c
    want = y.reshape(n, 2, 2, C, h, w).permute(0, 3, 4, 1, 5, 2).reshape(n, C, 2 * h, 2 * w)   # pixel (2y + ph/2, 2x + ph%2)
    out, sat = H.conv3x3_mx(H.to_act_mx(x), wt, b, act=_ffi.ACT_RELU, out_planes=planes, out_sexp=H.sexp_for(want.float()), d2s=True)
    scale = max(1.0, want.abs().max().item())
    assert sat == 0 and H.max_err(out.read(2), want) < TOL * scale      # hi + dequantised al8 plane: both landed in the right pixels
    if planes & LO:
        assert H.max_err(out.read(0), want) < TOL * scale              # hi + lo


@pytest.mark.parametrize("x2q", [False, True])
def test_conv3x3_mx_masked_depth_to_space_is_run_to_run_deterministic(H, x2q):
    """Regression (round 3): the epilogue's 16-byte stores are asm blocks (store-data hazard, conv_mx_kernel.h), which the compiler's
    hazard recogniser does not look into - and in the instantiations that spill SGPRs the store's descriptor / offset registers are
    reloaded with v_readlane right in front of it (VALU-written SGPR -> VMEM read: 5 wait states).  The masked depth-to-space
    instantiation of the f16x2+fp8 arithmetic (the ColorProbNet's up-convs under precision="x2q") wrote its hi plane through stale
    offsets, differently from run to run.  Sub-pixel up-conv weights (4 live taps per phase), tap mask on, five repeats, raw buffers
    byte-identical - and correct."""
    planes = _ffi.PLANE_QL if x2q else Q
    gen = g(5)
    n, cin, C, h, w = 8, 128, 64, 64, 64
    x = torch.relu(torch.randn(n, cin, h, w, generator=gen))
    w3 = torch.randn(C, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    wt = torch.zeros(4 * C, cin, 3, 3)
    for py in range(2):
        for px in range(2):
            for ky in range(3):
                for kx in range(3):          # nearest-x2 upsample + 3x3 == 4 phases of pre-summed 2x2 taps on the low-res grid
                    wt[(py * 2 + px) * C:(py * 2 + px + 1) * C, :, ((py + ky - 1) >> 1) + 1, ((px + kx - 1) >> 1) + 1] += w3[:, :, ky, kx]
    b = torch.randn(C, generator=gen) * 0.1
    want = F.relu(F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w3.double(), b.double(), padding=1))
    src = H.to_act_mx(x, planes)
    packed = H.pack_conv_mx(wt, x2q)
    bufs = []
    for _ in range(5):
        out, sat = H.conv3x3_mx(src, wt, b, act=_ffi.ACT_RELU, out_planes=planes, out_sexp=H.sexp_for(want.float()), packed=packed, x2q=x2q, d2s=True, tapmask=True)
        bufs.append(out.buf.clone())
    for o in bufs[1:]:
        assert torch.equal(o, bufs[0])
    assert sat == 0 and H.max_err(out.read(2), want) < TOL * max(1.0, want.abs().max().item())


# ---- the f16 + fp6x2 arithmetic (AR 3, the HourGlass2's default since round 3) ---------------------------------------------------------
Q6 = _ffi.PLANE_Q6
TOL6 = 4e-4      # the correction products carry 3 mantissa bits and fp6's narrow exponent range: ~2^-14 of the accumulated |w||a| mass
                 # (fp8: 2^-16); end to end this leaves 2-3e-4 on ab against the 1e-3 bar


def test_mx6_layout_roundtrip(H):
    """MX fp6 q planes: one E8M0 scale per pixel and 32 channels (byte 24 of the slot), so a6 is x to 3 mantissa bits for every value
    within a factor 4 of ITS pixel-block's maximum (fp6 e2m3 has 2 exponent bits; below that the subnormal step of the block's scale),
    whatever the tensor's range; hi + al6 is x to 2^-14.5 of the block maximum."""
    x = torch.randn(2, 64, 9, 11, generator=g(0)) * 3
    x[:, :, :4] *= 2.0 ** -6                                   # rows at 1/64 of the rest: a per-tensor scale would flush them to fp6 zero
    a = H.to_act_mx(x, LO | Q6)
    assert H.max_err(a.read(0).cpu(), x) < 2e-7 * 16
    a6 = a.read(1).cpu()
    bmax = x.reshape(2, 2, 32, 9, 11).abs().amax(2, keepdim=True).expand(2, 2, 32, 9, 11).reshape(x.shape)
    big = x.abs() > bmax / 4
    assert ((a6 - x).abs()[big] <= x.abs()[big] * 2 ** -3.9).all()
    assert ((a6 - x).abs() <= bmax * 2 ** -3.9).all()
    assert ((a.read(2).cpu() - x).abs() <= bmax * 2 ** -14.5).all()


MX6_CASES = [
    # cin, cout, h, w, stride, act, slope, bn, res, out_planes
    (64, 64, 32, 32, 1, _ffi.ACT_LRELU, 0.2, True, False, Q6),
    (32, 64, 24, 40, 1, _ffi.ACT_RELU, 0.0, True, False, LO | Q6),
    (64, 128, 32, 64, 2, _ffi.ACT_LRELU, 0.2, True, False, Q6),
    (256, 32, 8, 8, 1, _ffi.ACT_RELU, 0.0, False, True, LO | Q6),
    (96, 64, 17, 33, 1, _ffi.ACT_NONE, 0.0, False, False, Q6),      # ragged: partial tiles, odd sizes
    (256, 256, 32, 32, 1, _ffi.ACT_RELU, 0.0, False, False, Q6),
    (64, 64, 64, 64, 1, _ffi.ACT_RELU, 0.0, False, False, Q6),
]


@pytest.mark.parametrize("case", MX6_CASES)
def test_conv3x3_mx6_matches_torch(H, case):
    cin, cout, h, w, stride, act, slope, use_bn, use_res, planes = case
    gen = g(cin * 1000 + cout + h + 6)
    n = 3
    x = torch.randn(n, cin, h, w, generator=gen)
    if cin >= 128: x = F.relu(x)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    bn = (torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.1) if use_bn else None
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = torch.randn(n, cout, ho, wo, generator=gen) if use_res else None
    want = _ref(x, wt, b, stride, act, slope, bn, res)
    scale = max(1.0, want.abs().max().item())
    kw = dict(stride=stride, act=act, slope=slope, bn_scale=bn[0] if bn else None, bn_shift=bn[1] if bn else None, q6=True)
    out, sat = H.conv3x3_mx(H.to_act_mx(x, Q6), wt, b, res=H.to_act_mx(res, LO) if use_res else None, out_planes=planes,
                            out_sexp=H.sexp_for(want), **kw)
    assert sat == 0
    if planes & LO:
        assert H.max_err(out.read(0), want) < TOL6 * scale
    assert H.max_err(out.read(2), want) < (TOL6 + 2 ** -12) * scale           # hi + dequantised al6: the fp6 slots landed where they belong
    a6 = out.read(1).cpu().double()
    assert (a6 - want).abs().max() <= scale * 2 ** -3.9 + TOL6 * scale
    if not use_res:
        out32, _ = H.conv3x3_mx(H.to_act_mx(x, Q6), wt, b, out_f32=True, **kw)
        assert H.max_err(out32, want) < TOL6 * scale
    # five repeats: byte-identical raw buffers
    first = out.buf.clone()
    for _ in range(3):
        again, _ = H.conv3x3_mx(H.to_act_mx(x, Q6), wt, b, res=H.to_act_mx(res, LO) if use_res else None, out_planes=planes,
                                out_sexp=H.sexp_for(want), **kw)
        assert torch.equal(again.buf, first)


# (cin, cout, h, w, n): shapes whose launch takes NJ = 4 on 256 CUs (dispatch_mx_ar: whole groups of four per workgroup at no worse balance than
# the per-image launch finds), the forward's stride-2 layers among them; the last two do not (too few tiles: the per-image loop) and pin the fallback
NJ4_CASES = [(64, 128, 256, 256, 4), (64, 128, 256, 256, 8), (128, 256, 128, 128, 8), (256, 512, 64, 64, 16), (32, 64, 250, 254, 8),
             (64, 128, 128, 192, 12), (64, 128, 64, 96, 12), (32, 64, 33, 47, 4)]


@pytest.mark.parametrize("case", NJ4_CASES)
def test_conv3x3_stride2_weight_chunks_shared_by_four_images(H, case):
    """Round 5: batches whose images split into whole groups of four per workgroup take the stride-2 tile with NJ = 4 (a weight chunk fetched
    once per four images, four accumulator sets: conv_mx_kernel.h).  Every accumulator sees the same chunks and taps in the same order, so
    the batch must equal - bit for bit - its images run one at a time (one image: the per-image loop), in both arithmetics that have
    stride-2 layers (f16x3: SpixelNet / ColorProbNet; f16 + fp6x2: HourGlass2), and match torch."""
    cin, cout, h, w, n = case
    gen = g(cin * 31 + cout + h + n)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    want = _ref(x, wt, b, 2, _ffi.ACT_LRELU, 0.2, None, None)
    scale = max(1.0, want.abs().max().item())
    # f16x3
    got = H.from_act(H.conv3x3(H.to_act(x), wt, b, stride=2, act=_ffi.ACT_LRELU, slope=0.2)).cpu()
    assert H.max_err(got, want) < 2e-5 * scale
    for i in range(n):
        one = H.from_act(H.conv3x3(H.to_act(x[i:i + 1]), wt, b, stride=2, act=_ffi.ACT_LRELU, slope=0.2)).cpu()
        assert torch.equal(one[0], got[i])
    # f16 + fp6x2
    kw = dict(stride=2, act=_ffi.ACT_LRELU, slope=0.2, q6=True, out_planes=LO | Q6, out_sexp=H.sexp_for(want))
    packed = H.pack_conv_mx(wt, 2)
    sx = H.sexp_for(x)                                   # one input scale for the batch and for its images alone
    out, sat = H.conv3x3_mx(H.to_act_mx(x, Q6, sx), wt, b, packed=packed, **kw)
    assert sat == 0 and H.max_err(out.read(0), want) < TOL6 * scale
    planes = [out.read(k).cpu() for k in range(3)]
    for i in range(n):
        one, _ = H.conv3x3_mx(H.to_act_mx(x[i:i + 1], Q6, sx), wt, b, packed=packed, **kw)
        for k in range(3):
            assert torch.equal(one.read(k).cpu()[0], planes[k][i])


def test_conv3x3_mx6_heavy_tailed_rows_block_scaled_weights(H):
    """Round 4: the fp6 WEIGHT operands carry one E8M0 scale per (output channel, 32-input-channel block, tap) - byte 24 of the weight
    slot, taken by the MFMA as its weight-side scale operand - where round 3 scaled a whole row of 9 Cin weights with one exponent: a
    row with a single 30x outlier then pushed every other weight of the row below fp6's subnormal step and lost their w a_lo / w_lo a
    corrections (the error of a plain fp16 product: ~2^-11 of the accumulated mass).  Same tolerance as the Gaussian cases."""
    gen = g(77)
    n, cin, cout, h, w = 2, 128, 64, 24, 24
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    for co in range(cout):                               # one 30 sigma weight per row, in a different block / tap each
        wt[co, (7 * co) % cin, co % 3, (co // 3) % 3] = 30.0 * math.sqrt(2.0 / (9 * cin)) * (1 if co & 1 else -1)
    b = torch.randn(cout, generator=gen) * 0.1
    want = _ref(x, wt, b, 1, _ffi.ACT_NONE, 0.0, None, None)
    scale = max(1.0, want.abs().max().item())
    out32, _ = H.conv3x3_mx(H.to_act_mx(x, Q6), wt, b, out_f32=True, q6=True)
    # the part of the output the outliers do not touch must be as accurate as with Gaussian rows
    err = H.max_err(out32, want)
    print("mx6 heavy-tailed rows: max err %.3e of range %.2f (plain fp16 operands would leave ~%.1e)" % (err, scale, 2 ** -11 * scale))
    assert err < TOL6 * scale
    t = torch.distributions.StudentT(3.0).sample((cout, cin, 3, 3)) * math.sqrt(2.0 / (9 * cin)) / math.sqrt(3.0)
    want = _ref(x, t, b, 1, _ffi.ACT_NONE, 0.0, None, None)
    out32, _ = H.conv3x3_mx(H.to_act_mx(x, Q6), t, b, out_f32=True, q6=True)
    assert H.max_err(out32, want) < TOL6 * max(1.0, want.abs().max().item())


def test_mx6_weight_slots_carry_block_scales(H):
    """The packed image itself: per (32-cout block, chunk, tap) 2 KiB = 64 lanes x 32 bytes in two 1 KiB pieces; a lane's 32 bytes are
    the slot of row (lane & 31): wl6 (lane < 32) / w6 (lane >= 32), 32 six-bit fields in bytes 0-23 (field f = channel
    mx6_field_channel(f)), E8M0 in byte 24.  Dequantised, w6 must be w to 3 mantissa bits relative to ITS block's maximum."""
    gen = g(5)
    cout, cin = 32, 64
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * 0.05
    wt[:, :32] *= 2.0 ** -7                               # block 0 of every row at 1/128 of block 1: a row scale would zero it
    buf, _ = H.pack_conv_mx(wt, 2)
    raw = buf.cpu().numpy()
    nck = cin // 16
    def field_channel(f): return 8 * ((f & 15) >> 2) + 4 * (f >> 4) + (f & 3)
    def fp6(c):
        e, m = (c >> 3) & 3, c & 7
        v = m * 0.125 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 1)
        return -v if c & 32 else v
    worst = 0.0
    for g32 in range(cin // 32):
        for tap in range(9):
            base = ((0 * nck + 2 * g32 + 1) * 9 + tap) * 2048
            for lane in (32, 33, 47, 63):                 # w6 slots of rows 0, 1, 15, 31
                slot = bytes(raw[base + lane * 16: base + lane * 16 + 16]) + bytes(raw[base + 1024 + lane * 16: base + 1024 + lane * 16 + 16])
                sc = slot[24]
                bits = int.from_bytes(slot[:24], "little")
                row = lane & 31
                blk = wt[row, g32 * 32:(g32 + 1) * 32, tap // 3, tap % 3]
                for f in range(32):
                    got = fp6((bits >> (6 * f)) & 63) * 2.0 ** (sc - 127 + 11)      # the w6 side carries 2^-11 for the al6 planes' 2^11
                    worst = max(worst, abs(got - blk[field_channel(f)].item()) / blk.abs().max().item())
    assert worst <= 2 ** -3.9, worst


def test_conv3x3_mx6_two_source_fp8_in_fp6_out_and_depth_to_space(H):
    """The two joints of the fp6 HourGlass2: its first layer is a two-source f16+fp8x2 layer (fp8 planes from the upfeat / gray kernels)
    that WRITES fp6 planes; its up-convs are masked 4-phase convs with the depth-to-space epilogue."""
    gen = g(31)
    a = torch.randn(2, 64, 24, 40, generator=gen); s = torch.randn(2, 32, 24, 40, generator=gen)
    wt = torch.randn(64, 96, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * 96)); b = torch.randn(64, generator=gen) * 0.1
    want = _ref(torch.cat((a, s), 1), wt, b, 1, _ffi.ACT_RELU, 0.0, None, None)
    e = min(H.sexp_for(a), H.sexp_for(s))
    out, sat = H.conv3x3_mx(H.to_act_mx(a, sexp=e), wt, b, src1=H.to_act_mx(s, sexp=e), act=_ffi.ACT_RELU, out_planes=Q6, out_sexp=H.sexp_for(want))
    assert sat == 0 and H.max_err(out.read(2), want) < (TOL + 2 ** -12) * max(1.0, want.abs().max().item())
    with pytest.raises(_ffi.DiscoError):      # a one-source f16+fp8x2 layer cannot write fp6 planes (that path is not compiled in)
        H.conv3x3_mx(H.to_act_mx(a), torch.randn(64, 64, 3, 3) * 0.05, b, out_planes=Q6)
    n, cin, C, h, w = 2, 128, 32, 13, 19
    x = torch.relu(torch.randn(n, cin, h, w, generator=gen))
    w3 = torch.randn(C, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    w4 = torch.zeros(4 * C, cin, 3, 3)
    for py in range(2):
        for px in range(2):
            for ky in range(3):
                for kx in range(3):
                    w4[(py * 2 + px) * C:(py * 2 + px + 1) * C, :, ((py + ky - 1) >> 1) + 1, ((px + kx - 1) >> 1) + 1] += w3[:, :, ky, kx]
    bb = torch.randn(C, generator=gen) * 0.1
    want = F.relu(F.conv2d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), w3.double(), bb.double(), padding=1))
    out, sat = H.conv3x3_mx(H.to_act_mx(x, Q6), w4, bb, act=_ffi.ACT_RELU, out_planes=Q6, out_sexp=H.sexp_for(want.float()), q6=True, d2s=True, tapmask=True)
    assert sat == 0 and H.max_err(out.read(2), want) < (TOL6 + 2 ** -12) * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("shape", [(64, 64, 32, 32), (64, 64, 17, 33), (32, 128, 24, 40)])
def test_conv3x3_mx_tail_chunk_matches_torch(H, shape):
    """The HourGlass2's input layer: cat(64 features, gray) -> 64.  The features come with fp8 planes, the single extra channel as the
    16-channel fp16 tail source (x_hi, x_lo, x_hi) against the weights (w_h, w_h, w_l): ONE K = 16 MFMA per tap forms its exact
    three-product split (pack variant 3, disco_op_gray_tail; conv_mx_kernel.h chunk KIND 3).  Against F.conv2d on the concatenation;
    the extra channel alone (features zero) must come out at f16x3 accuracy, far below the fp8-corrected channels' tolerance."""
    c0, co, h, w = shape
    gen = g(c0 * 7 + h)
    n = 3
    x = F.relu(torch.randn(n, c0, h, w, generator=gen))
    gr = torch.randn(n, 1, h, w, generator=gen) * 3
    wt = torch.randn(co, c0 + 1, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * c0))
    wt[:, c0] *= 4                       # make the extra channel matter
    b = torch.randn(co, generator=gen) * 0.1
    packed = H.pack_conv_mx(wt, 3)
    for feats, tol in ((x, TOL), (torch.zeros_like(x), 2e-6)):
        want = _ref(torch.cat([feats, gr], 1), wt, b, 1, _ffi.ACT_LRELU, 0.2, None, None)
        scale = max(1.0, want.abs().max().item())
        sx = H.sexp_for(torch.cat([feats.flatten(), gr.flatten()]))
        a0 = H.to_act_mx(feats, sexp=sx)
        a1 = H.to_gray_tail(gr, sx)
        out, sat = H.conv3x3_mx(a0, wt, b, src1=a1, act=_ffi.ACT_LRELU, slope=0.2, out_planes=LO, out_sexp=H.sexp_for(want), packed=packed)
        assert sat == 0
        assert H.max_err(out.read(0), want) < tol * scale
    again, _ = H.conv3x3_mx(a0, wt, b, src1=a1, act=_ffi.ACT_LRELU, slope=0.2, out_planes=LO, out_sexp=H.sexp_for(want), packed=packed)
    assert torch.equal(again.buf, out.buf)
    with pytest.raises(_ffi.DiscoError):          # 32 k + 1 input channels only
        H.pack_conv_mx(torch.randn(64, 64, 3, 3), 3)
