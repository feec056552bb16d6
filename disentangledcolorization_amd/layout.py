"""Checkpoint layout of DISCO's AnchorColorProb (461 tensors).

Reference: models/model.py:33-76 (module tree), models/network.py:147-218
(ColorProbNet), :260-291 (SpixelNet), :125-134 (HourGlass2),
models/transformer2d.py:9-49 (encoder layers); file contract in
main/utils_train.py:140-151 (torch.load(...)['state_dict'], strict load).

`state_dict_spec()` returns the ordered list of (key, shape, dtype-name, kind)
that `torch.nn.Module.state_dict()` of the reference produces; it is what
`AnchorColorProb.load_state_dict` checks strictly and what `synth.py` fills.
"""
from collections import OrderedDict

# (name, cin, cout) in registration order — models/network.py:263-282
_SEG_CONVS = [
    ("conv0a", 1, 16), ("conv0b", 16, 16), ("conv1a", 16, 32), ("conv1b", 32, 32),
    ("conv2a", 32, 64), ("conv2b", 64, 64), ("conv3a", 64, 128), ("conv3b", 128, 128),
    ("conv4a", 128, 256), ("conv4b", 256, 256),
]
_SEG_DEC = [  # deconv(cin,cout) followed by conv(2*cout -> cout)
    ("deconv3", 256, 128, "conv3_1"), ("deconv2", 128, 64, "conv2_1"),
    ("deconv1", 64, 32, "conv1_1"), ("deconv0", 32, 16, "conv0_1"),
]

# ColorProbNet encoder blocks: (block, [(idx, cin, cout)], bn_idx, bn_channels)
_REP_SN = [
    ("conv1_2", [(0, 1, 64), (2, 64, 64)], 4, 64),
    ("conv2_3", [(0, 64, 128), (2, 128, 128), (4, 128, 128)], 6, 128),
    ("conv3_3", [(0, 128, 256), (2, 256, 256), (4, 256, 256)], 6, 256),
    ("conv4_3", [(0, 256, 512), (2, 512, 512), (4, 512, 512)], 6, 512),
    ("conv5_3", [(0, 512, 512), (2, 512, 512), (4, 512, 512)], 6, 512),
    ("conv6_3", [(0, 512, 512), (2, 512, 512), (4, 512, 512)], 6, 512),
    ("conv7_3", [(0, 512, 512), (2, 512, 512), (4, 512, 512)], 6, 512),
]

N_ENC_LAYERS = 6
D_MODEL = 64
D_FF = 256
N_HEAD = 8
N_VOCAB = 313


def _conv(spec, key, cin, cout, bias=True, k=3):
    spec.append((key + ".weight", (cout, cin, k, k), "float32", "conv_w"))
    if bias:
        spec.append((key + ".bias", (cout,), "float32", "bias"))


def _sn_conv(spec, key, cin, cout):
    # torch.nn.utils.spectral_norm (legacy hook): bias first, then weight_orig/u/v
    spec.append((key + ".bias", (cout,), "float32", "bias"))
    spec.append((key + ".weight_orig", (cout, cin, 3, 3), "float32", "sn_w"))
    spec.append((key + ".weight_u", (cout,), "float32", "sn_u"))
    spec.append((key + ".weight_v", (9 * cin,), "float32", "sn_v"))


def _bn(spec, key, c):
    spec.append((key + ".weight", (c,), "float32", "bn_w"))
    spec.append((key + ".bias", (c,), "float32", "bn_b"))
    spec.append((key + ".running_mean", (c,), "float32", "bn_mean"))
    spec.append((key + ".running_var", (c,), "float32", "bn_var"))
    spec.append((key + ".num_batches_tracked", (), "int64", "bn_count"))


def state_dict_spec(hint2regress=False):
    """hint2regress: --hint2regress checkpoints carry trg_word_emb (64,67) and trg_word_prj (2,64) (model.py:63-64)."""
    s = []
    # ---- segnet (SpixelSeg.net = SpixelNet) ----
    p = "segnet.net."
    for name, cin, cout in _SEG_CONVS:
        _conv(s, p + name + ".0", cin, cout, bias=False)
        _bn(s, p + name + ".1", cout)
    for dname, cin, cout, cname in _SEG_DEC:
        s.append((p + dname + ".0.weight", (cin, cout, 4, 4), "float32", "deconv_w"))
        s.append((p + dname + ".0.bias", (cout,), "float32", "bias"))
        _conv(s, p + cname + ".0", 2 * cout, cout, bias=False)
        _bn(s, p + cname + ".1", cout)
    _conv(s, p + "pred_mask0", 16, 9)
    # ---- repnet (ColorProbNet) ----
    p = "repnet."
    for blk, convs, bn_idx, bn_c in _REP_SN:
        for idx, cin, cout in convs:
            _sn_conv(s, f"{p}{blk}.{idx}", cin, cout)
        _bn(s, f"{p}{blk}.{bn_idx}", bn_c)
    _conv(s, p + "conv8up.1", 512, 256)
    _conv(s, p + "conv3short8.0", 256, 256)
    _conv(s, p + "conv8_3.1", 256, 256)
    _conv(s, p + "conv8_3.3", 256, 256)
    _bn(s, p + "conv8_3.5", 256)
    _conv(s, p + "conv9up.1", 256, 128)
    _conv(s, p + "conv9_2.0", 128, 128)
    _bn(s, p + "conv9_2.2", 128)
    _conv(s, p + "conv10up.1", 128, 64)
    _conv(s, p + "conv10_2.1", 64, 64)
    # ---- enhanceNet (HourGlass2) ----
    p = "enhanceNet."
    _conv(s, p + "inConv.inConv.0", 65, 64)
    _conv(s, p + "inConv.conv.0", 64, 64)
    _bn(s, p + "inConv.conv.2", 64)
    for name, cin, cout in (("down1", 64, 128), ("down2", 128, 256)):
        _conv(s, f"{p}{name}.conv.0", cin, cout)
        _conv(s, f"{p}{name}.conv.2", cout, cout)
        _bn(s, f"{p}{name}.conv.4", cout)
    for r in range(3):
        _conv(s, f"{p}residual.{r}.conv.0", 256, 256)
        _sn_conv(s, f"{p}residual.{r}.conv.1", 256, 256)
        _conv(s, f"{p}residual.{r}.conv.3", 256, 256)
    for name, cin, cout in (("up2", 256, 128), ("up1", 128, 64)):
        _conv(s, f"{p}{name}.conv1", cin, cout)
        _conv(s, f"{p}{name}.combine", 2 * cout, cout)
        _conv(s, f"{p}{name}.conv2.0", cout, cout)
        _conv(s, f"{p}{name}.conv2.2", cout, cout)
        _bn(s, f"{p}{name}.conv2.4", cout)
    _conv(s, p + "outConv", 64, 2)
    # ---- transformer encoders ----
    for path in ("wildpath", "hintpath"):
        for l in range(N_ENC_LAYERS):
            q = f"{path}.layers.{l}."
            s.append((q + "self_attn.in_proj_weight", (3 * D_MODEL, D_MODEL), "float32", "lin_w"))
            s.append((q + "self_attn.in_proj_bias", (3 * D_MODEL,), "float32", "lin_b"))
            s.append((q + "self_attn.out_proj.weight", (D_MODEL, D_MODEL), "float32", "lin_w"))
            s.append((q + "self_attn.out_proj.bias", (D_MODEL,), "float32", "lin_b"))
            s.append((q + "linear1.weight", (D_FF, D_MODEL), "float32", "lin_w"))
            s.append((q + "linear1.bias", (D_FF,), "float32", "lin_b"))
            s.append((q + "linear2.weight", (D_MODEL, D_FF), "float32", "lin_w"))
            s.append((q + "linear2.bias", (D_MODEL,), "float32", "lin_b"))
            for n in ("norm1", "norm2"):
                s.append((q + n + ".weight", (D_MODEL,), "float32", "ln_w"))
                s.append((q + n + ".bias", (D_MODEL,), "float32", "ln_b"))
    s.append(("mid_word_prj.weight", (N_VOCAB, D_MODEL), "float32", "lin_w"))
    s.append(("trg_word_emb.weight", (D_MODEL, D_MODEL + (2 if hint2regress else N_VOCAB) + 1), "float32", "lin_w"))
    s.append(("trg_word_prj.weight", (2 if hint2regress else N_VOCAB, D_MODEL), "float32", "lin_w"))
    return s


def spec_dict():
    return OrderedDict((k, (shape, dt, kind)) for k, shape, dt, kind in state_dict_spec())
