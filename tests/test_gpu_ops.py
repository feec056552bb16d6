"""-m gpu: every HIP operator against the CPU oracle (oracle/disco_ref.py, plain torch fp32 ops)
and against the golden vectors captured from the reference (tests/golden/components.npz)."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from disentangledcolorization_amd import _ffi  # noqa: E402
from oracle import disco_ref as R  # noqa: E402


@pytest.fixture(scope="module")
def H():
    import gpu_helpers
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    _ffi.lib()  # fails loudly if the extension is missing
    return gpu_helpers


@pytest.fixture(scope="module")
def comp(golden_dir):
    return np.load(os.path.join(golden_dir, "components.npz"))


def g(seed):
    return torch.Generator().manual_seed(seed)


def test_layout_roundtrip(H):
    x = torch.randn(2, 19, 7, 9, generator=g(0))
    y = H.from_act(H.to_act(x, 32), 19).cpu()
    assert H.max_err(x, y) < 2e-7 * 4   # hi+lo carries ~22 bits


CONV_CASES = [
    # cin, cout, h, w, stride, act, slope, bn, res
    (16, 16, 32, 32, 1, _ffi.ACT_LRELU, 0.1, False, False),
    (32, 64, 24, 40, 1, _ffi.ACT_RELU, 0.0, True, False),
    (64, 64, 16, 16, 1, _ffi.ACT_NONE, 0.0, False, True),
    (64, 128, 32, 64, 2, _ffi.ACT_LRELU, 0.2, True, False),
    (128, 256, 16, 16, 2, _ffi.ACT_RELU, 0.0, False, False),
    (256, 32, 8, 8, 1, _ffi.ACT_RELU, 0.0, False, True),
    (48, 64, 17, 33, 1, _ffi.ACT_TANH, 0.0, False, False),   # ragged: partial tiles, odd sizes
    (512, 512, 8, 8, 1, _ffi.ACT_LRELU, 0.2, True, False),
]


def _ref_conv(x, w, b, stride, act, slope, bn, res):
    y = F.conv2d(x, w, b, stride=stride, padding=1)
    if res is not None:
        y = y + res
    if act == _ffi.ACT_RELU: y = F.relu(y)
    elif act == _ffi.ACT_LRELU: y = F.leaky_relu(y, slope)
    elif act == _ffi.ACT_TANH: y = torch.tanh(y)
    if bn is not None:
        y = y * bn[0][None, :, None, None] + bn[1][None, :, None, None]
    return y


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3_matches_torch(H, case, prec=_ffi.PREC_F16X3):
    cin, cout, h, w, stride, act, slope, use_bn, use_res = case
    gen = g(cin * 1000 + cout)
    x = torch.randn(2, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    bn = (torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.1) if use_bn else None
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = torch.randn(2, cout, ho, wo, generator=gen) if use_res else None
    want = _ref_conv(x, wt, b, stride, act, slope, bn, res)
    got = H.from_act(H.conv3x3(H.to_act(x), wt, b, stride=stride, act=act, slope=slope,
                               bn_scale=bn[0] if bn else None, bn_shift=bn[1] if bn else None,
                               res=H.to_act(res) if use_res else None, precision=prec))
    assert H.max_err(got, want) < 2e-5 * max(1.0, want.abs().max().item())       # fp32-class


@pytest.mark.parametrize("case", [(64, 64, 16, 16, 1, 2), (64, 128, 64, 96, 2, 3), (256, 512, 32, 32, 2, 3), (64, 128, 32, 64, 2, 2), (16, 16, 48, 48, 1, 4)])
def test_conv3x3_is_run_to_run_deterministic_and_lo_exact(H, case):
    """Regression: the lo plane of an output used to be stored while the next image's LDS-DMA was still queued; the store read
    its data registers late and lanes 12-15 / 28-31 of one dword arrived stale - an error of the size of the lo plane (~1e-3),
    different from run to run, on exactly these shapes.  Five repeats must agree bit for bit and stay at fp32-level error."""
    cin, cout, h, w, stride, n = case
    gen = g(cin * 1000 + cout)
    x = torch.randn(n, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    want = F.conv2d(x, wt, b, stride=stride, padding=1)
    xa = H.to_act(x)
    outs = [H.from_act(H.conv3x3(xa, wt, b, stride=stride, precision=_ffi.PREC_F16X3)).cpu() for _ in range(5)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    assert (outs[0] - want).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("exps", [(3, 5, 5), (-9, -6, -4), (12, 14, 10), (0, 4, 0)])
@pytest.mark.parametrize("use_bn", [False, True])
def test_conv3x3_scale_exponents(H, exps, use_bn):
    """One power-of-two exponent per activation tensor (common.h, struct Act): sources stored as x 2^sexp_in, residual as
    r 2^sexp_res, the output leaves as y 2^sexp_out; the epilogue folds the exact factors into its parameters.  Results must equal
    the unscaled run to fp32 rounding (they are bit-identical unless a plane leaves fp16's normal range), for inputs whose true
    magnitudes make fp16 storage impossible without the exponent (x ~ 2^-sexp_in)."""
    e_in, e_out, e_res = exps
    gen = g(77 + e_in)
    cin, cout, h, w = 64, 64, 20, 36
    x = torch.randn(2, cin, h, w, generator=gen) * 2.0 ** -e_in           # e.g. |x| ~ 4000 for e_in = -12: beyond fp16 for sums
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1 * 2.0 ** -e_in
    res = torch.randn(2, cout, h, w, generator=gen) * 2.0 ** -e_res
    bn = (torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.1 * 2.0 ** -e_in) if use_bn else None
    want = _ref_conv(x.double(), wt.double(), b.double(), 1, _ffi.ACT_LRELU, 0.2, (bn[0].double(), bn[1].double()) if bn else None, res.double())
    got = H.from_act_scaled(H.conv3x3(H.to_act_scaled(x, e_in), wt, b, act=_ffi.ACT_LRELU, slope=0.2, bn_scale=bn[0] if bn else None,
                                      bn_shift=bn[1] if bn else None, res=H.to_act_scaled(res, e_res), sexp_in=e_in, sexp_out=e_out, sexp_res=e_res), e_out)
    assert H.max_err(got, want) < 2e-5 * want.abs().max().item()


S2_CASES = [
    # cin, cout, h, w, act, slope, bn   (stride 2)
    (64, 128, 64, 96, _ffi.ACT_LRELU, 0.2, True),
    (16, 32, 34, 50, _ffi.ACT_LRELU, 0.1, False),     # ragged output 17x25: partial tiles at both edges
    (128, 256, 16, 16, _ffi.ACT_RELU, 0.0, False),
    (256, 512, 32, 32, _ffi.ACT_LRELU, 0.2, True),
    (32, 64, 2, 2, _ffi.ACT_NONE, 0.0, False),        # 1x1 output: every tap but the centre block is padding
    (32, 64, 33, 47, _ffi.ACT_RELU, 0.0, False),      # odd input sizes
]


@pytest.mark.parametrize("case", S2_CASES)
def test_conv3x3_stride2(H, case):
    """The stride-2 tiles (de-interleaved halo columns) against torch."""
    cin, cout, h, w, act, slope, use_bn = case
    gen = g(cin * 77 + cout + h)
    x = torch.randn(3, cin, h, w, generator=gen)
    wt = torch.randn(cout, cin, 3, 3, generator=gen) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=gen) * 0.1
    bn = (torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen) * 0.1) if use_bn else None
    want = _ref_conv(x, wt, b, 2, act, slope, bn, None)
    got = H.from_act(H.conv3x3(H.to_act(x), wt, b, stride=2, act=act, slope=slope, bn_scale=bn[0] if bn else None, bn_shift=bn[1] if bn else None))
    assert got.shape == want.shape and H.max_err(got, want) < 2e-5 * max(1.0, want.abs().max().item())


def test_op_entry_points_reject_bad_sizes(H):
    """Every op validates its sizes on the host (error code + message, no launch): zero / negative dimensions, a zero
    superpixel size (would divide by zero), mismatched channel counts."""
    L = _ffi.lib()
    t = torch.zeros(4096, device=H.DEV)
    for rc in (L.disco_op_poolfeat(_ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), None, None, 1, 2, 16, 16, 0, _ffi.ptr(t), 1 << 20, H.stream()),
               L.disco_op_upfeat(_ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), 1, 2, 0, 4, 16, H.stream()),
               L.disco_op_kmeans_anchors(_ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), None, 0, _ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), None, 1, 16, 0, 64, 0, H.stream()),
               L.disco_op_kmeans_anchors(_ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), None, 0, _ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), None, 1, 16, 4, 65, 0, H.stream()),
               L.disco_op_encoder_stack(_ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), 0, 16, _ffi.ptr(t), 1 << 20, H.stream()),
               L.disco_op_rgb8_to_lab(_ffi.ptr(t), _ffi.ptr(t), _ffi.ptr(t), None, 1, 8, 8, 4, 8, H.stream())):
        assert rc < 0 and L.disco_last_error()
    d = _ffi.ConvDesc(1, 16, 16, 24, 0, 0, 0, 16, 1, _ffi.ACT_NONE, 0.0, 0)          # 24 input channels: not a multiple of 16
    assert L.disco_op_conv3x3(C.byref(d), _ffi.ptr(t), None, _ffi.ptr(t), None, None, None, None, _ffi.ptr(t), H.stream()) < 0
    d = _ffi.ConvDesc(0, 16, 16, 16, 0, 0, 0, 16, 1, _ffi.ACT_NONE, 0.0, 0, 0)
    assert L.disco_op_conv3x3(C.byref(d), _ffi.ptr(t), None, _ffi.ptr(t), None, None, None, None, _ffi.ptr(t), H.stream()) < 0
    torch.cuda.synchronize()


def test_conv3x3_upsample_and_concat_on_read(H):
    gen = g(5)
    a = torch.randn(2, 32, 12, 20, generator=gen)      # half-resolution source, nearest x2 on read
    s = torch.randn(2, 16, 24, 40, generator=gen)      # full-resolution skip
    wt = torch.randn(64, 48, 3, 3, generator=gen) * 0.05
    b = torch.randn(64, generator=gen) * 0.1
    want = F.relu(F.conv2d(torch.cat((R.up2(a), s), 1), wt, b, padding=1))
    got = H.from_act(H.conv3x3(H.to_act(a), wt, b, src1=H.to_act(s), up0=True, act=_ffi.ACT_RELU))
    assert H.max_err(got, want) < 2e-5 * want.abs().max().item()


def test_deconv4x4(H):
    """ConvTranspose2d 4x4 s2 as a 4-phase 3x3 conv: runs on conv3x3_mx_kernel's depth-to-space epilogue (AR = 2), the forward's own path."""
    gen = g(6)
    x = torch.randn(2, 32, 9, 11, generator=gen)
    wt = torch.randn(32, 16, 4, 4, generator=gen) * 0.1
    b = torch.randn(16, generator=gen) * 0.1
    want = F.leaky_relu(F.conv_transpose2d(x, wt, b, stride=2, padding=1), 0.1)
    L = _ffi.lib()
    nb = C.c_size_t()
    _ffi.check(L.disco_op_deconv4x4_pack(None, 32, 16, None, C.byref(nb)))
    packed = torch.empty(nb.value, device=H.DEV, dtype=torch.uint8)
    _ffi.check(L.disco_op_deconv4x4_pack(_ffi.ptr(wt.contiguous()), 32, 16, _ffi.ptr(packed), C.byref(nb)))
    xa = H.to_act(x)
    out = torch.empty(2, 2, 18, 22, 16, device=H.DEV, dtype=torch.float16)
    bd = b.to(H.DEV)
    _ffi.check(L.disco_op_deconv4x4(_ffi.ptr(xa), _ffi.ptr(packed), _ffi.ptr(bd), _ffi.ptr(out), 2, 9, 11, 32, 16, 0.1,
                                    0, H.stream()))
    assert H.max_err(H.from_act(out), want) < 2e-5 * want.abs().max().item()


def test_pool_unpool_sizes_golden(H, comp):
    prob = torch.from_numpy(comp["pool_prob"]).to(H.DEV)
    feat = torch.from_numpy(comp["pool_feat"]).to(H.DEV)
    n, c, hh, ww = feat.shape
    h, w = hh // 16, ww // 16
    L = _ffi.lib()
    pooled = torch.empty(n, c, h, w, device=H.DEV); conf = torch.empty(n, 1, h, w, device=H.DEV)
    sizes = torch.empty(n, h * w, device=H.DEV)
    ws = torch.empty(n * h * w * 9 * (c + 2) * 4, device=H.DEV, dtype=torch.uint8)
    _ffi.check(L.disco_op_poolfeat(_ffi.ptr(feat), _ffi.ptr(prob), _ffi.ptr(pooled), _ffi.ptr(conf), _ffi.ptr(sizes), n, c,
                                   hh, ww, 16, _ffi.ptr(ws), ws.numel(), H.stream()))
    assert H.max_err(pooled, torch.from_numpy(comp["pool_out"])) < 1e-5
    assert H.max_err(conf, torch.from_numpy(comp["pool_conf"])) < 1e-6
    assert torch.equal(sizes.cpu().reshape(n, 1, h, w), torch.from_numpy(comp["spix_size"]))   # exact
    tok = torch.from_numpy(comp["up_tok"]).to(H.DEV)
    out = torch.empty(n, tok.shape[1], hh, ww, device=H.DEV)
    _ffi.check(L.disco_op_upfeat(_ffi.ptr(tok), _ffi.ptr(prob), _ffi.ptr(out), n, tok.shape[1], h, w, 16, H.stream()))
    assert H.max_err(out, torch.from_numpy(comp["up_out"])) < 1e-6


@pytest.mark.parametrize("hw", [(16, 16), (32, 48), (8, 12)])
def test_position_encoding_golden(H, comp, hw):
    h, w = hw
    pos = torch.empty(h * w, 64, device=H.DEV)
    _ffi.check(_ffi.lib().disco_op_position_encoding(_ffi.ptr(pos), h, w, H.stream()))
    want = torch.from_numpy(comp["pos_%dx%d" % hw]).flatten(1).t()
    assert H.max_err(pos, want) < 2e-6


def _encoder_weights(sd, path):
    keys = ["self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
            "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias",
            "norm2.weight", "norm2.bias"]
    return torch.cat([sd[f"{path}.layers.{l}.{k}"].reshape(-1) for l in range(6) for k in keys])


@pytest.mark.parametrize("n,hw", [(3, (16, 16)), (1, (8, 12)), (1, (32, 48))])
def test_encoder_stack_matches_oracle(H, synth_sd, n, hw):
    h, w = hw
    l = h * w
    x = torch.randn(n, l, 64, generator=g(l))
    pos = R.position_encoding(h, w).flatten(1).t().contiguous()
    want = R.encoder_stack(synth_sd, "wildpath", x, pos[None].expand(n, -1, -1))
    wts = _encoder_weights(synth_sd, "wildpath").to(H.DEV)
    assert wts.numel() == _ffi.lib().disco_op_encoder_weight_floats()
    xd, pd = x.to(H.DEV), pos.to(H.DEV)
    out = torch.empty_like(xd)
    ws = torch.empty(n * l * 704 * 4, device=H.DEV, dtype=torch.uint8)
    _ffi.check(_ffi.lib().disco_op_encoder_stack(_ffi.ptr(xd), _ffi.ptr(pd), _ffi.ptr(wts), _ffi.ptr(out), n, l,
                                                 _ffi.ptr(ws), ws.numel(), H.stream()))
    assert H.max_err(out, want) < 2e-5


@pytest.mark.parametrize("n,hw", [(3, (16, 16)), (40, (16, 16)), (1, (8, 12)), (1, (32, 32)), (2, (32, 48)), (1, (48, 48))])
def test_encoder_stack_with_key_padding_mask_matches_oracle(H, synth_sd, n, hw):
    """use_mask (model.py:121-125 -> transformer2d.py:53-54): the reference's float key_padding_mask adds 1.0 to the scores of the keys whose
    superpixel holds fewer than 25 pixels, in every layer.  Every attention form: attention_kernel<1> (few workgroups), <4> (40 images),
    the key-split MFMA form (1 024 / 1 536 tokens) and the four-query-tile MFMA form (2 304 tokens) against the oracle's stack with the
    same bias; and the mask must matter (the unmasked stack gives another result)."""
    h, w = hw
    l = h * w
    x = torch.randn(n, l, 64, generator=g(l + n))
    pos = R.position_encoding(h, w).flatten(1).t().contiguous()
    sizes = torch.randint(0, 512, (n, l), generator=g(3 * l + n)).float() / 256.0
    sizes[:, ::5] = torch.randint(0, 25, sizes[:, ::5].shape, generator=g(l)).float() / 256.0      # a fifth of the superpixels below 25 pixels
    sizes[0, 1] = 25.0 / 256.0                                                                      # the boundary itself is NOT small (<)
    bias = R.entry_mask(sizes, 16)
    assert 0.1 < float(bias.mean()) < 0.4 and bias[0, 1] == 0
    want = R.encoder_stack(synth_sd, "hintpath", x[:4], pos[None].expand(min(n, 4), -1, -1), key_bias=bias[:4])
    plain = R.encoder_stack(synth_sd, "hintpath", x[:1], pos[None], key_bias=None)
    wts = _encoder_weights(synth_sd, "hintpath").to(H.DEV)
    xd, pd, sz = x.to(H.DEV), pos.to(H.DEV), sizes.to(H.DEV)
    out = torch.empty_like(xd)
    ws = torch.empty(n * l * 704 * 4, device=H.DEV, dtype=torch.uint8)
    _ffi.check(_ffi.lib().disco_op_encoder_stack_masked(_ffi.ptr(xd), _ffi.ptr(pd), _ffi.ptr(wts), _ffi.ptr(sz), _ffi.ptr(out), n, l,
                                                        _ffi.ptr(ws), ws.numel(), H.stream()))
    torch.cuda.synchronize()
    assert H.max_err(out[:4], want) < 2e-5
    assert H.max_err(out[:1], plain) > 1e-3, "the mask did not change anything"
    # an image's result does not depend on the batch it is part of, with the mask as without
    if n > 1:
        one = torch.empty_like(xd[:1])
        _ffi.check(_ffi.lib().disco_op_encoder_stack_masked(_ffi.ptr(xd[:1].contiguous()), _ffi.ptr(pd), _ffi.ptr(wts), _ffi.ptr(sz[:1].contiguous()),
                                                            _ffi.ptr(one), 1, l, _ffi.ptr(ws), ws.numel(), H.stream()))
        torch.cuda.synchronize()
        assert torch.equal(one[0], out[0])


@pytest.mark.parametrize("n,hw", [(1, (16, 16)), (3, (16, 16)), (2, (9, 11)), (1, (32, 48)), (5, (7, 3))])
def test_encoder_tail_path_equals_the_tiled_one(H, synth_sd, n, hw):
    """The two ways the stack runs a layer's second half: 64-row tiles (post_attention_kernel + a q/k/v launch per layer) and 16-row tiles on
    v_mfma_f32_16x16x4_f32 with the next layer's in-projection fused in (encoder_tail_kernel: the latency path of small token counts).
    The op takes the second when its workspace has room for the packed weight image.  Bit-identical, ragged last tiles included."""
    h, w = hw
    l = h * w
    x = torch.randn(n, l, 64, generator=g(7 * l + n))
    pos = R.position_encoding(h, w).flatten(1).t().contiguous()
    wts = _encoder_weights(synth_sd, "hintpath").to(H.DEV)
    xd, pd = x.to(H.DEV), pos.to(H.DEV)
    outs = []
    for extra in (0, 4 << 20):
        out = torch.empty_like(xd)
        ws = torch.empty(n * l * 384 * 4 + 256 + extra, device=H.DEV, dtype=torch.uint8)
        _ffi.check(_ffi.lib().disco_op_encoder_stack(_ffi.ptr(xd), _ffi.ptr(pd), _ffi.ptr(wts), _ffi.ptr(out), n, l,
                                                     _ffi.ptr(ws), ws.numel(), H.stream()))
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    want = R.encoder_stack(synth_sd, "hintpath", x, pos[None].expand(n, -1, -1))
    assert H.max_err(outs[1], want) < 2e-5


_ATTN_AB = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gpu_helpers as H
from test_gpu_ops import _attn_ab_run
np.savez(sys.argv[1], **_attn_ab_run(H))
"""

_ATTN_AB_SHAPES = [(1, (32, 48)), (2, (32, 32)), (3, (16, 16)), (2, (9, 11)), (1, (8, 12)), (5, (7, 3)), (1, (1, 5)), (1, (40, 52))]


def _attn_ab_run(H):
    """Both encoder stacks' weights over token counts from 5 to 2 080 (ragged last key tiles and query tiles, counts below one half-tile of
    keys, whole chunks + remainders): name -> output."""
    from disentangledcolorization_amd import synth
    sd = synth.synth_state_dict(130)
    out = {}
    for case, (n, (h, w)) in enumerate(_ATTN_AB_SHAPES):
        l = h * w
        x = torch.randn(n, l, 64, generator=g(31 * l + n))
        pos = R.position_encoding(h, w).flatten(1).t().contiguous()
        wts = _encoder_weights(sd, "wildpath" if case % 2 == 0 else "hintpath").to(H.DEV)
        xd, pd = x.to(H.DEV), pos.to(H.DEV)
        o = torch.empty_like(xd)
        ws = torch.empty(n * l * 384 * 4 + 256 + (4 << 20), device=H.DEV, dtype=torch.uint8)
        _ffi.check(_ffi.lib().disco_op_encoder_stack(_ffi.ptr(xd), _ffi.ptr(pd), _ffi.ptr(wts), _ffi.ptr(o), n, l,
                                                     _ffi.ptr(ws), ws.numel(), H.stream()))
        torch.cuda.synchronize()
        out["o%d" % case] = o.cpu().numpy()
    return out


def test_attention_on_the_matrix_cores_matches_oracle_and_the_valu_kernel(H, synth_sd, tmp_path):
    """attention_mfma_kernel (S^T = K Q^T on 32x32x2 fp32 MFMAs, P V on 4x4x1 ones with the S^T accumulators in place as the operand; the
    default from 1 024 tokens on) forced at EVERY token count in a subprocess (DISCO_ATTN_MFMA=1), against attention_kernel forced at every
    count (DISCO_ATTN_MFMA=0) and against the oracle's encoder stack: the same mathematics in another summation order."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    res = {}
    for mode in ("1", "0"):
        out = str(tmp_path / ("attn%s.npz" % mode))
        subprocess.run([sys.executable, "-c", _ATTN_AB % (os.path.dirname(here), here), out], check=True,
                       env=dict(os.environ, DISCO_ATTN_MFMA=mode), cwd=os.path.dirname(here))
        res[mode] = np.load(out)
    for case, (n, (h, w)) in enumerate(_ATTN_AB_SHAPES):
        l = h * w
        x = torch.randn(n, l, 64, generator=g(31 * l + n))
        pos = R.position_encoding(h, w).flatten(1).t().contiguous()
        want = R.encoder_stack(synth_sd, "wildpath" if case % 2 == 0 else "hintpath", x, pos[None].expand(n, -1, -1))
        a, b = torch.from_numpy(res["1"]["o%d" % case]), torch.from_numpy(res["0"]["o%d" % case])
        assert torch.isfinite(a).all()
        assert (a - want).abs().max().item() < 2e-5, (case, n, l)
        assert (a - b).abs().max().item() < 1e-5, (case, n, l)


@pytest.mark.parametrize("hw", [(16, 16), (32, 32), (32, 48), (48, 48), (64, 64)])
def test_encoder_stack_result_does_not_depend_on_the_batch(H, synth_sd, hw):
    """An image's tokens come out of the stack bit for bit the same alone and as one of 9 images - at 256 tokens (attention_kernel, whose
    query tiling follows the grid size) and at 1 024 ... 4 096 (attention_mfma_kernel, whose form follows the token count alone: a rule by
    grid size gave different roundings to the same image in a batch of 64 and alone)."""
    h, w = hw
    l = h * w
    x = torch.randn(9, l, 64, generator=g(5 * l))
    pos = R.position_encoding(h, w).flatten(1).t().contiguous()
    wts = _encoder_weights(synth_sd, "wildpath").to(H.DEV)
    outs = []
    for n in (1, 9):
        xd, pd = x[:n].contiguous().to(H.DEV), pos.to(H.DEV)
        out = torch.empty_like(xd)
        ws = torch.empty(n * l * 384 * 4 + 256 + (4 << 20), device=H.DEV, dtype=torch.uint8)
        _ffi.check(_ffi.lib().disco_op_encoder_stack(_ffi.ptr(xd), _ffi.ptr(pd), _ffi.ptr(wts), _ffi.ptr(out), n, l,
                                                     _ffi.ptr(ws), ws.numel(), H.stream()))
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0][0], outs[1][0])


def _kmeans_gpu(H, x, sizes, init, fallback, k, d=64, channel_major=0):
    n, l = x.shape[0], (x.shape[2] if channel_major else x.shape[1])
    xd, sd_ = x.to(H.DEV).contiguous(), sizes.to(H.DEV).contiguous()
    idx = torch.as_tensor(np.asarray(init), dtype=torch.int32).to(H.DEV)
    fb = torch.as_tensor(np.asarray(fallback), dtype=torch.int32).to(H.DEV) if fallback is not None else None
    mf = fb.shape[1] if fb is not None else 0
    assign = torch.empty(n, l, dtype=torch.int32, device=H.DEV); anchor = torch.empty(n, k, dtype=torch.int32, device=H.DEV)
    mask = torch.empty(n, l, device=H.DEV); info = torch.empty(n, 2, dtype=torch.int32, device=H.DEV)
    # with scratch: images of more than 512 tokens run on several workgroups (kmeans_coop_kernel); DISCO_KMEANS_V1 / _COOP switch paths
    wsb = _ffi.lib().disco_op_kmeans_workspace_bytes(n, l)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=H.DEV)
    _ffi.check(_ffi.lib().disco_op_kmeans_anchors_ws(_ffi.ptr(xd), _ffi.ptr(sd_), _ffi.ptr(idx), _ffi.ptr(fb), mf, _ffi.ptr(assign),
                                                     _ffi.ptr(anchor), _ffi.ptr(mask), _ffi.ptr(info), n, l, k, d, channel_major,
                                                     _ffi.ptr(ws), wsb, H.stream()))
    torch.cuda.synchronize()
    return assign.cpu().long(), anchor.cpu().long(), mask.cpu(), info.cpu()


def test_kmeans_anchors_golden_and_oracle(H, comp):
    x = torch.from_numpy(comp["km_x"])            # (4,256,64); the last one has 200 identical rows
    init = comp["km_init"]
    sizes = (torch.randint(0, 512, (4, 256), generator=g(3)).float() / 256.0)
    fallback = torch.randint(0, 256, (4, 160), generator=g(4))   # <= (K-1)*20 events
    want_assign, want_anchor, want_mask, events = [], [], [], []
    for i in range(4):
        a, passes, ev = R.kmeans_one(x[i], init[i], 8, fallback_rows=[int(v) for v in fallback[i]])
        want_assign.append(a); events.append(ev)
    want_assign = torch.stack(want_assign)
    want_anchor, want_mask = R.anchors_from_clusters(want_assign, sizes, 8)
    assign, anchor, mask, info = _kmeans_gpu(H, x, sizes, init, fallback.numpy(), 8)
    assert torch.equal(assign, want_assign)
    assert torch.equal(anchor, want_anchor)
    assert torch.equal(mask, want_mask)
    assert info[:, 1].tolist() == events and events[3] > 0
    # first three (no fallback draws involved) are also the reference's own assignments
    assert np.array_equal(assign[:3].numpy(), comp["km_ids"][:3].astype(np.int64))


@pytest.mark.parametrize("l,k,d", [(64, 2, 64), (256, 16, 64), (256, 8, 64), (200, 32, 64), (130, 5, 64), (255, 3, 64), (384, 8, 64), (400, 8, 64), (1000, 5, 64), (1536, 16, 64),
                                   (4096, 32, 64), (4608, 8, 64), (8192, 16, 64), (32768, 32, 64), (96, 8, 2), (1536, 16, 2)])
def test_kmeans_every_path_matches_oracle(H, l, k, d):
    """All four kernel paths - points in LDS (L <= 384), tiled 1024-thread path with the member list in LDS (L <= 4096) or in
    global memory (beyond, while the per-segment counts fit in LDS), scan fallback (32768 points, K = 32) - for 64-feature token rows and for the 2-feature channel-major colours of the validation forward;
    clustered points (so the iterations really move), duplicated points (empty clusters -> fallback rows)."""
    n = 3
    gen = g(l * 31 + k)
    centres = torch.randn(n, k, d, generator=gen) * 2.0
    which = torch.randint(0, k, (n, l), generator=gen)
    x = torch.gather(centres, 1, which[..., None].expand(-1, -1, d)) + torch.randn(n, l, d, generator=gen) * 0.7
    x[2, : l // 2] = x[2, 0]                                   # image 2: half of the points identical -> empty clusters
    init = np.stack([np.random.RandomState(l + i).choice(l, k, replace=False) for i in range(n)]).astype(np.int32)
    init[2, : max(1, k // 2)] = np.arange(max(1, k // 2))      # ... and several initial rows inside the identical half
    sizes = torch.randint(0, 512, (n, l), generator=gen).float() / 256.0
    fallback = torch.randint(0, l, (n, 20 * k), generator=gen)
    want_assign, events = [], []
    for i in range(n):
        a, passes, ev = R.kmeans_one(x[i], init[i], k, fallback_rows=[int(v) for v in fallback[i]])
        want_assign.append(a); events.append(ev)
    want_assign = torch.stack(want_assign)
    want_anchor, want_mask = R.anchors_from_clusters(want_assign, sizes, k)
    xin = x.transpose(1, 2).contiguous() if d == 2 else x      # (n,2,l) channel-major like NCHW colours
    assign, anchor, mask, info = _kmeans_gpu(H, xin, sizes, init, fallback.numpy(), k, d, 1 if d == 2 else 0)
    assert torch.equal(assign, want_assign)
    assert torch.equal(anchor, want_anchor) and torch.equal(mask, want_mask)
    assert info[:, 1].tolist() == events and (k < 4 or events[2] > 0)


_KM_AB = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gpu_helpers as H
from test_gpu_ops import _kmeans_gpu, _km_ab_inputs
out = {}
for case, (x, sizes, init, fallback, k) in enumerate(_km_ab_inputs()):
    a, an, m, info = _kmeans_gpu(H, x, sizes, init, fallback, k)
    out.update({"a%%d" %% case: a.numpy(), "an%%d" %% case: an.numpy(), "m%%d" %% case: m.numpy(), "i%%d" %% case: info.numpy()})
np.savez(sys.argv[1], **out)
"""


def _km_ab_inputs():
    """Token sets for the two k-means kernels side by side: clustered points, points on a coarse grid (exact ties between centres), half of
    the points identical (empty clusters -> fallback rows), ragged sizes, K from 2 to 32."""
    cases = []
    for seed, (n, l, k) in enumerate([(16, 256, 8), (6, 256, 16), (4, 256, 32), (5, 200, 8), (5, 97, 5), (3, 64, 2), (3, 256, 8),
                                      (3, 1536, 8), (2, 1000, 5), (2, 4096, 32), (2, 5000, 16), (2, 300, 8)]):
        gen = g(900 + seed)
        centres = torch.randn(n, k, 64, generator=gen) * 2.0
        which = torch.randint(0, k, (n, l), generator=gen)
        x = torch.gather(centres, 1, which[..., None].expand(-1, -1, 64)) + torch.randn(n, l, 64, generator=gen) * 0.7
        if seed == 6:
            x = torch.round(x)                               # a coarse grid: distance ties, repeated rows
        x[n - 1, : l // 2] = x[n - 1, 0]
        init = np.stack([np.random.RandomState(50 * seed + i).choice(l, k, replace=False) for i in range(n)]).astype(np.int32)
        init[n - 1, : max(1, k // 2)] = np.arange(max(1, k // 2))
        sizes = torch.randint(0, 512, (n, l), generator=gen).float() / 256.0
        fallback = torch.randint(0, l, (n, 20 * k), generator=gen).numpy()
        cases.append((x, sizes, init, fallback, k))
    return cases


def test_kmeans_small_kernel_equals_the_general_one(H, tmp_path):
    """kmeans_small_kernel (<= 256 points of 64 features: the 256 x 256 image's latency path) and kmeans_tiled_kernel (more points: the
    --no_resize sizes) against kmeans_anchor_kernel, which a subprocess runs on the same inputs under DISCO_KMEANS_V1=1 (its LDS-list and
    global-list paths): assignments, anchors, hint masks, pass counts and empty-cluster events identical, element by element."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "v1.npz")
    env = dict(os.environ, DISCO_KMEANS_V1="1")
    subprocess.run([sys.executable, "-c", _KM_AB % (os.path.dirname(here), here), out], check=True, env=env, cwd=os.path.dirname(here))
    ref = np.load(out)
    # ... and the one-workgroup tiled kernel (DISCO_KMEANS_COOP=0) against the same reference: in-process the larger sets take kmeans_coop_kernel
    out2 = str(tmp_path / "tiled.npz")
    subprocess.run([sys.executable, "-c", _KM_AB % (os.path.dirname(here), here), out2], check=True, env=dict(os.environ, DISCO_KMEANS_COOP="0"),
                   cwd=os.path.dirname(here))
    tiled = np.load(out2)
    assert all(np.array_equal(tiled[k], ref[k]) for k in ref.files)
    some_events = 0
    for case, (x, sizes, init, fallback, k) in enumerate(_km_ab_inputs()):
        a, an, m, info = _kmeans_gpu(H, x, sizes, init, fallback, k)
        assert np.array_equal(a.numpy(), ref["a%d" % case]) and np.array_equal(an.numpy(), ref["an%d" % case])
        assert np.array_equal(m.numpy(), ref["m%d" % case]) and np.array_equal(info.numpy(), ref["i%d" % case])
        some_events += int(info[:, 1].sum())
    assert some_events > 0


def _kmeans_coop_call(H, x, sizes, init, fallback, k, stream=None):
    """disco_op_kmeans_anchors_ws with scratch on `stream` (asynchronous); returns the device buffers + a closure that reads the result and
    the number of images that fell back to the one-workgroup kernel."""
    n, l = x.shape[:2]
    st = stream or torch.cuda.current_stream()
    with torch.cuda.stream(st):
        xd, sd_ = x.to(H.DEV).contiguous(), sizes.to(H.DEV).contiguous()
        idx = torch.as_tensor(np.asarray(init), dtype=torch.int32).to(H.DEV)
        fb = torch.as_tensor(np.asarray(fallback), dtype=torch.int32).to(H.DEV)
        assign = torch.empty(n, l, dtype=torch.int32, device=H.DEV); anchor = torch.empty(n, k, dtype=torch.int32, device=H.DEV)
        mask = torch.empty(n, l, device=H.DEV); info = torch.empty(n, 2, dtype=torch.int32, device=H.DEV)
        wsb = _ffi.lib().disco_op_kmeans_workspace_bytes(n, l)
        assert wsb > 0
        ws = torch.empty(wsb, dtype=torch.uint8, device=H.DEV)
    keep = (xd, sd_, idx, fb, ws)

    def launch():
        _ffi.check(_ffi.lib().disco_op_kmeans_anchors_ws(_ffi.ptr(xd), _ffi.ptr(sd_), _ffi.ptr(idx), _ffi.ptr(fb), fb.shape[1], _ffi.ptr(assign),
                                                         _ffi.ptr(anchor), _ffi.ptr(mask), _ffi.ptr(info), n, l, k, 64, 0, _ffi.ptr(ws), wsb,
                                                         C.c_void_p(st.cuda_stream)))

    def read():
        cnt = C.c_int(-1)
        _ffi.check(_ffi.lib().disco_op_kmeans_fallbacks(_ffi.ptr(ws), n, l, C.c_void_p(st.cuda_stream), C.byref(cnt)))
        return (assign.cpu(), anchor.cpu(), mask.cpu(), info.cpu()), cnt.value, keep
    return launch, read


@pytest.mark.parametrize("inject", [1, 2])
def test_kmeans_coop_degrades_to_the_one_workgroup_kernel_instead_of_trapping(H, monkeypatch, inject):
    """kmeans_coop_kernel's workgroups wait for each other.  Round 5 bounded the wait with a deadline that TRAPPED (a sticky
    hipErrorLaunchFailure: the end of a serving process).  Now residency is proven per image before anything is exchanged, and an image
    whose workgroups do not all arrive (inject 1: workgroup 0 never registers) or that loses one later (inject 2: workgroup 0 leaves after
    admission; the others run into the backstop) is computed by kmeans_tiled_kernel right behind: the SAME assignments, anchors, hint mask,
    pass counts and events, no error, the context alive.  (clusterkit.py:49-58 is a per-image loop that cannot hang; neither can this.)"""
    import time
    cases = [c for c in _km_ab_inputs() if c[0].shape[1] > 512][:2 if inject == 2 else 4]
    for x, sizes, init, fallback, k in cases:
        n = x.shape[0]
        monkeypatch.delenv("DISCO_KMEANS_COOP_INJECT", raising=False)
        launch, read = _kmeans_coop_call(H, x, sizes, init, fallback, k)
        launch()
        want, fell, _ = read()
        assert fell == 0, "an undisturbed launch must be admitted"
        monkeypatch.setenv("DISCO_KMEANS_COOP_INJECT", str(inject))
        launch2, read2 = _kmeans_coop_call(H, x, sizes, init, fallback, k)
        t0 = time.perf_counter()
        launch2()
        got, fell, _ = read2()
        dt = time.perf_counter() - t0
        assert fell == n, "every image must have been handed to the one-workgroup kernel (%d of %d)" % (fell, n)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
        assert dt < (0.2 if inject == 1 else 3.0), "the admission time-out is milliseconds, the backstop under a second: took %.3f s" % dt
        monkeypatch.delenv("DISCO_KMEANS_COOP_INJECT")
        launch()                                            # the context is alive and the several-workgroup kernel still works
        again, fell, _ = read()
        assert fell == 0 and all(torch.equal(a, b) for a, b in zip(again, want))


def test_kmeans_coop_launches_that_oversubscribe_the_gpu_do_not_deadlock(H):
    """The advisor's scenario: each launch passes the 'a quarter of the CUs' heuristic, but EIGHT of them on eight streams ask for twice the
    CUs the GPU has - launches can end up partially resident, each waiting for workgroups the others keep off the CUs.  The first form of
    the kernel spun there until its 5e9-cycle deadline and trapped.  With admission, half-arrived images give their CUs back and are
    computed by the one-workgroup kernel: every launch returns the undisturbed result, whatever the interleaving was."""
    x, sizes, init, fallback, k = [c for c in _km_ab_inputs() if c[0].shape[1] == 1536][0]
    reps = 21                                           # 21 images x 3 workgroups = 63 of the 64 a launch may take
    xs = x[:1].repeat(reps, 1, 1) + torch.arange(reps).reshape(-1, 1, 1) * 1e-3
    szs, ini, fbk = sizes[:1].repeat(reps, 1), np.repeat(init[:1], reps, 0), np.repeat(fallback[:1], reps, 0)
    launch, read = _kmeans_coop_call(H, xs, szs, ini, fbk, k)
    launch()
    want, fell, _ = read()
    assert fell == 0
    streams = [torch.cuda.Stream() for _ in range(8)]
    calls = [_kmeans_coop_call(H, xs, szs, ini, fbk, k, st) for st in streams]
    torch.cuda.synchronize()
    total = 0
    for _ in range(5):
        for launch_i, _r in calls:
            launch_i()
        for _l, read_i in calls:
            got, fell, _ = read_i()
            total += fell
            for a, b in zip(got, want):
                assert torch.equal(a, b)
    torch.cuda.synchronize()
    print("oversubscribed coop launches: %d of %d images fell back to the one-workgroup kernel" % (total, 5 * 8 * reps))


@pytest.mark.parametrize("t", [0, 1, 2])
def test_select_colors_golden(H, comp, t):
    prob = torch.from_numpy(comp["samp_prob"])
    n, _, h, w = prob.shape
    logit = torch.log(prob).to(H.DEV).contiguous()      # softmax(log p) == p up to rounding
    colors = torch.empty(n, 2, h * w, device=H.DEV); labels = torch.empty(n, h * w, dtype=torch.int32, device=H.DEV)
    _ffi.check(_ffi.lib().disco_op_select_colors(_ffi.ptr(logit), _ffi.ptr(colors), _ffi.ptr(labels), n, h * w, t, H.stream()))
    assert torch.equal(colors.cpu().reshape(n, 2, h, w), torch.from_numpy(comp["samp_T%d" % t]))


def test_nearest_bin_golden(H, comp):
    ab = torch.from_numpy(comp["enc_ab"]).to(H.DEV)
    n, _, h, w = ab.shape
    labels = torch.empty(n, h * w, dtype=torch.int32, device=H.DEV)
    _ffi.check(_ffi.lib().disco_op_nearest_bin(_ffi.ptr(ab), _ffi.ptr(labels), n, h * w, H.stream()))
    assert torch.equal(labels.cpu().long().reshape(n, 1, h, w), torch.from_numpy(comp["enc_label"]))


# ---- §8f "next" rows: the helpers either side of the forward (disentangledcolorization_amd/basic.py) -----------------


def test_basic_mirror_pool_unpool(H, comp):
    from disentangledcolorization_amd import basic

    prob = torch.from_numpy(comp["pool_prob"]).to(H.DEV)
    feat = torch.from_numpy(comp["pool_feat"]).to(H.DEV)
    pooled, conf = basic.poolfeat(feat, prob, 16, 16, True)
    assert H.max_err(pooled, torch.from_numpy(comp["pool_out"])) < 1e-5 and H.max_err(conf, torch.from_numpy(comp["pool_conf"])) < 1e-6
    assert torch.equal(basic.get_spixel_size(prob, 16, 16).cpu(), torch.from_numpy(comp["spix_size"]))
    tok = torch.from_numpy(comp["up_tok"]).to(H.DEV)
    assert H.max_err(basic.upfeat(tok, prob, 16, 16), torch.from_numpy(comp["up_out"])) < 1e-6
    assert basic.tensor2array(tok).shape == (2, 4, 6, 5)
    with pytest.raises(_ffi.DiscoError):
        basic.upfeat(tok.cpu(), prob.cpu(), 16, 16)


@pytest.mark.parametrize("t", [0, 1, 2, 3])
def test_basic_mirror_decode_ind2ab(H, comp, t):
    from disentangledcolorization_amd import basic

    lg = torch.from_numpy(comp["dec_logit"]).to(H.DEV)
    got = basic.ColorLabel().decode_ind2ab(lg, T=t)
    assert torch.equal(got.cpu(), torch.from_numpy(comp["dec_ab_T%d" % t]))


def test_basic_mirror_colour_space(H, comp):
    from disentangledcolorization_amd import basic

    lab = basic.rgb2lab(torch.from_numpy(comp["cs_rgb"]).to(H.DEV))
    assert H.max_err(lab, torch.from_numpy(comp["cs_lab"])) < 5e-6
    rgb = basic.lab2rgb(torch.from_numpy(comp["cs_lab_in"]).to(H.DEV))
    assert H.max_err(rgb, torch.from_numpy(comp["cs_rgb_out"])) < 5e-6
    # round trip at full benchmark size (64 x 256 x 256): rgb -> lab -> rgb is the identity on in-gamut colours
    big = torch.rand(64, 3, 256, 256, generator=g(9)).to(H.DEV)
    assert H.max_err(basic.lab2rgb(basic.rgb2lab(big)), big) < 2e-4


def test_basic_mirror_hint_overlay_and_image_io(H, golden_dir):
    """§8f rows 1-2 through the C ABI: mark_color_hints bit-exact against the reference's golden output; the fused
    fetch (pad-to-16 quirk + /255 + RGB->Lab + split) and save (Lab->RGB->uint8, de-padded) kernels against the oracle."""
    from disentangledcolorization_amd import basic

    gd = np.load(os.path.join(golden_dir, "posthoc.npz"))
    gray, target, base, gate = (torch.from_numpy(gd[k]).to(H.DEV) for k in ("gray", "target", "base", "gate"))
    for ks in (3, 5):
        assert torch.equal(basic.mark_color_hints(gray, target, gate, ks).cpu(), torch.from_numpy(gd["marked_k%d" % ks]))
        assert torch.equal(basic.mark_color_hints(gray, target, gate, kernel_size=ks, base_ABs=base).cpu(),
                           torch.from_numpy(gd["marked_base_k%d" % ks]))
    with pytest.raises(_ffi.DiscoError):
        basic.mark_color_hints(gray, target, gate, kernel_size=4)
    for key, T in (("ann_ab_T038", 0.38), ("ann_ab_T150", 1.5)):      # annealed-mean decoding, default T = 0.38
        got = basic.ColorLabel().decode_ind2ab(torch.from_numpy(gd["ann_logit"]).to(H.DEV), T=T)
        assert H.max_err(got, torch.from_numpy(gd[key])) < 5e-6
    assert H.max_err(basic.ColorLabel().decode_ind2ab(torch.from_numpy(gd["ann_logit"]).to(H.DEV)), torch.from_numpy(gd["ann_ab_T038"])) < 5e-6
    rs = np.random.RandomState(3)
    for h, w in [(37, 50), (32, 50), (37, 48), (32, 48), (250, 333)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = R.fetch_from_rgb8(img, org_size=True)
        got = basic.fetch_data_from_rgb8(img, org_size=True)
        assert got[3] == want[3] == (h, w)
        for a, b in zip(got[:3], want[:3]):
            assert a.shape == b.shape and H.max_err(a, b) < 5e-6
        lab = torch.cat((want[0], want[1]), 1)
        back = basic.normLabs_to_rgb8(lab.to(H.DEV), h, w).cpu().numpy()
        ref8 = R.labs_to_rgb8(lab, h, w)
        # truncation to uint8: a 1-ulp difference of the float result may cross an integer boundary
        assert back.shape == ref8.shape and np.abs(back.astype(int) - ref8.astype(int)).max() <= 1
        assert np.abs(back[0].astype(int) - img.astype(int)).max() <= 1          # the uint8 round trip
    # the default (resize to 256x256) branch: the resized uint8 image must equal the oracle's cv2 restatement BIT FOR BIT
    # (integer fixed-point arithmetic), gray / ab / rgb follow within float tolerance
    for h, w in [(37, 53), (480, 640), (612, 612), (512, 512), (256, 256), (1200, 900)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        want = R.fetch_from_rgb8(img, org_size=False)
        got = basic.fetch_data_from_rgb8(img, org_size=False, return_resized=True)
        assert got[3] == want[3] == (h, w)              # the original size, as fetch_data returns it in both branches
        assert np.array_equal(got[4].cpu().numpy(), R.cv2_resize_linear_u8(img, 256, 256)), (h, w)
        for a, b in zip(got[:3], want[:3]):
            assert a.shape == b.shape and H.max_err(a, b) < 5e-6


def test_spixelseg_dropin(H, golden_dir, synth_sd):
    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.model import SpixelSeg

    gd = np.load(os.path.join(golden_dir, "spixelseg.npz"))
    n, h, w, seed = (int(v) for v in gd["recipe"])
    m = SpixelSeg(inChannel=1, outChannel=9, batchNorm=True)
    assert sorted(m.state_dict().keys()) == list(gd["keys"])
    m.load_state_dict({k[len("segnet."):]: v for k, v in synth_sd.items() if k.startswith("segnet.")})   # strict
    m = m.cuda().eval()
    gray, _ = synth.synth_inputs(n, h, w, seed=seed)
    prob = m(gray.cuda())
    torch.cuda.synchronize()
    assert H.max_err(prob, torch.from_numpy(gd["prob"])) < 1e-4
    with pytest.raises(RuntimeError):
        m.load_state_dict({"net.conv0a.0.weight": torch.zeros(16, 1, 3, 3)})
