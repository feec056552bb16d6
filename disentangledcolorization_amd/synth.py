"""Deterministic synthetic checkpoint with the real DISCO layout.

The real DISCO checkpoint is a Google-Drive download (checkpoints/disco_download.sh:1) and is
not available offline, and a freshly constructed reference model produces inf/NaN in eval mode
(un-iterated spectral-norm u/v).  This module manufactures a *well-conditioned*
checkpoint of identical layout (layout.py, 461 tensors) from a seed:

* every tensor is drawn from NumPy's legacy `RandomState` (bit-stable across platforms and
  NumPy versions) in `layout.state_dict_spec()` order;
* spectral-norm `weight_u/weight_v` are the converged power-iteration vectors of
  `weight_orig.view(Cout,-1)` (float64, fixed iteration count), so `sigma = u.(W v)` is the true
  spectral norm like in a trained checkpoint;
* BatchNorm running statistics come from the per-layer table `_BN_STATS` below — the batch
  statistics measured once by `oracle/calibrate_synth.py` (sequentially, layer after layer, on a
  fixed calibration batch) and frozen here as rounded scalars — with a per-channel jitter, so
  activations stay O(1) through all 67 convs without any data-dependent step at run time.

It is used by bench.py, the tests and oracle/make_golden.py.  It is data generation, not part
of the colorization arithmetic.
"""
import zlib
from collections import OrderedDict

import numpy as np

from .layout import state_dict_spec

# key of the BN layer -> (mean, var) of its input over the calibration batch
# (written by oracle/calibrate_synth.py; do not edit by hand)
_BN_STATS = {
    "segnet.net.conv0a.1": (-0.00157, 0.6459),
    "segnet.net.conv0b.1": (-0.08291, 1.413),
    "segnet.net.conv1a.1": (-0.114, 1.42),
    "segnet.net.conv1b.1": (-0.1646, 0.8011),
    "segnet.net.conv2a.1": (-0.08685, 1.236),
    "segnet.net.conv2b.1": (0.03309, 1.17),
    "segnet.net.conv3a.1": (-0.05107, 1.108),
    "segnet.net.conv3b.1": (-0.001175, 1.301),
    "segnet.net.conv4a.1": (0.02593, 1.136),
    "segnet.net.conv4b.1": (-0.07939, 1.341),
    "segnet.net.conv3_1.1": (0.05326, 0.9527),
    "segnet.net.conv2_1.1": (0.1737, 0.995),
    "segnet.net.conv1_1.1": (-0.05846, 0.9412),
    "segnet.net.conv0_1.1": (-0.1607, 1.235),
    "repnet.conv1_2.4": (0.04032, 0.004802),
    "repnet.conv2_3.6": (0.06767, 0.02405),
    "repnet.conv3_3.6": (0.071, 0.02352),
    "repnet.conv4_3.6": (0.06598, 0.01902),
    "repnet.conv5_3.6": (0.087, 0.03251),
    "repnet.conv6_3.6": (0.06997, 0.0223),
    "repnet.conv7_3.6": (0.08138, 0.02743),
    "repnet.conv8_3.5": (0.9336, 2.045),
    "repnet.conv9_2.2": (0.7344, 1.376),
    "enhanceNet.inConv.conv.2": (0.2984, 0.306),
    "enhanceNet.down1.conv.4": (0.4376, 0.4732),
    "enhanceNet.down2.conv.4": (0.5503, 0.7149),
    "enhanceNet.up2.conv2.4": (1.267, 3.727),
    "enhanceNet.up1.conv2.4": (0.6954, 1.122),
}


def _power_iteration(w2d: np.ndarray, rs: np.random.RandomState, iters: int = 40):
    w = w2d.astype(np.float64)
    u = rs.standard_normal(w.shape[0])
    u /= np.linalg.norm(u)
    v = None
    for _ in range(iters):
        v = w.T @ u
        v /= np.linalg.norm(v)
        u = w @ v
        u /= np.linalg.norm(u)
    return u.astype(np.float32), v.astype(np.float32)


def bn_running_stats(bn_key: str, channels: int, mean: float, var: float, seed: int):
    """Per-channel running_mean / running_var of BN layer `bn_key` around the layer scalars.

    Drawn from a key-derived stream (not the sequential one) so that the calibration pass can
    regenerate exactly these arrays while it walks the network."""
    rs = np.random.RandomState((zlib.crc32(bn_key.encode()) + seed) % (2 ** 32))
    rm = mean + np.sqrt(var) * 0.25 * rs.standard_normal(channels)
    rv = var * rs.uniform(0.6, 1.4, channels)
    return rm.astype(np.float32), rv.astype(np.float32)


def synth_numpy(seed: int = 130, bn_stats=None, hint2regress: bool = False) -> "OrderedDict[str, np.ndarray]":
    """The checkpoint as NumPy arrays (float32, int64 for num_batches_tracked)."""
    bn_stats = _BN_STATS if bn_stats is None else bn_stats
    rs = np.random.RandomState(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    pending_sn = None
    for key, shape, dt, kind in state_dict_spec(hint2regress):
        if kind in ("conv_w", "sn_w"):
            cout, cin, kh, kw = shape
            a = rs.standard_normal(shape) * np.sqrt(2.0 / (cin * kh * kw))
            if key == "enhanceNet.outConv.weight":
                a = a * 0.3  # keep the pre-tanh ab inside tanh's near-linear range
            if kind == "sn_w":
                pending_sn = a.reshape(cout, -1)
        elif kind == "deconv_w":
            cin, cout, kh, kw = shape
            a = rs.standard_normal(shape) * np.sqrt(2.0 / (cin * 4))  # 4 taps reach each output pixel
        elif kind == "sn_u":
            u, v = _power_iteration(pending_sn, rs)
            out[key] = u
            out[key[:-1] + "v"] = v
            continue
        elif kind == "sn_v":
            continue  # written together with weight_u
        elif kind == "bias":
            a = rs.standard_normal(shape) * 0.05
        elif kind == "bn_w":
            a = rs.uniform(0.7, 1.3, shape)
        elif kind == "bn_b":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_mean":
            bn_key = key[: -len(".running_mean")]
            m, v = bn_stats.get(bn_key, (0.0, 1.0))
            a = bn_running_stats(bn_key, shape[0], m, v, seed)[0]
        elif kind == "bn_var":
            bn_key = key[: -len(".running_var")]
            m, v = bn_stats.get(bn_key, (0.0, 1.0))
            a = bn_running_stats(bn_key, shape[0], m, v, seed)[1]
        elif kind == "bn_count":
            out[key] = np.asarray(10, dtype=np.int64)
            continue
        elif kind == "lin_w":
            bound = np.sqrt(6.0 / (shape[0] + shape[1]))
            a = rs.uniform(-bound, bound, shape)
        elif kind == "lin_b":
            a = rs.uniform(-0.05, 0.05, shape)
        elif kind == "ln_w":
            a = rs.uniform(0.8, 1.2, shape)
        elif kind == "ln_b":
            a = rs.uniform(-0.1, 0.1, shape)
        else:
            raise AssertionError(kind)
        out[key] = np.ascontiguousarray(a, dtype=np.float32)
    # restore layout order (weight_v was inserted right after weight_u already)
    ordered = OrderedDict((k, out[k]) for k, _, _, _ in state_dict_spec(hint2regress))
    return ordered


def synth_state_dict(seed: int = 130, bn_stats=None, hint2regress: bool = False):
    """The checkpoint as an OrderedDict of torch CPU tensors, loadable (strict) by the reference.
    hint2regress: the two head tensors take their --hint2regress shapes; they are drawn last, so every other
    tensor is identical to the plain checkpoint of the same seed."""
    import torch

    return OrderedDict((k, torch.from_numpy(np.array(v))) for k, v in synth_numpy(seed, bn_stats, hint2regress).items())


def synth_inputs(n: int, h: int = 256, w: int = 256, seed: int = 5, ab_scale: float = 0.0):
    """Config-1/2 synthetic inputs: L ~ U(-1,1), ab = 0 (or U(-ab_scale, ab_scale))."""
    import torch

    rs = np.random.RandomState(seed)
    gray = rs.uniform(-1.0, 1.0, (n, 1, h, w)).astype(np.float32)
    ab = (rs.uniform(-1.0, 1.0, (n, 2, h, w)) * ab_scale).astype(np.float32)
    return torch.from_numpy(gray), torch.from_numpy(ab)


def small_superpixel_variant(sd, slot: int = 0, boost: float = 4.0):
    """The checkpoint `sd` with SpixelNet's affinity head biased towards neighbour slot `slot` (pred_mask0.bias[slot] += boost): nearly
    every pixel then votes for that neighbour cell, so the cells along the opposite image border end up with (almost) no pixels of their
    own - superpixels smaller than 25 pixels, which is what `use_mask` marks (model.py:121-124).  The plain synthetic checkpoint produces
    none, so a use_mask test on it would compare two identical forwards.  Test data generation."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    out["segnet.net.pred_mask0.bias"][slot] += boost
    return out


def student_t_variant(sd, df: float, seed: int = 7, prefixes=("repnet.", "enhanceNet.")):
    """The checkpoint `sd` with the 3x3 conv weights under `prefixes` redrawn from a Student-t(df) distribution at the SAME per-tensor
    standard deviation (heavy tails as trained conv weights have them: a few weights per row far above the rest - what a block-scaled
    low-precision format has to survive; synth_numpy() draws Gaussians, whose row maximum is ~4 sigma).  Spectral-norm layers get the
    converged u / v of their new weight_orig.  Everything else (biases, BN tables, token path) is kept.  Test data generation."""
    import torch

    rs = np.random.RandomState(seed)
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    for key, shape, _, kind in state_dict_spec(any(k == "trg_word_emb.weight" and v.shape[1] == 67 for k, v in sd.items())):
        if kind not in ("conv_w", "sn_w") or not key.startswith(tuple(prefixes)) or key not in out:
            continue
        t = rs.standard_t(df, size=shape)
        a = (t / t.std() * float(sd[key].double().std())).astype(np.float32)
        out[key] = torch.from_numpy(a)
        if kind == "sn_w":
            u, v = _power_iteration(a.reshape(shape[0], -1), rs)
            base = key[: -len("weight_orig")]
            out[base + "weight_u"] = torch.from_numpy(u)
            out[base + "weight_v"] = torch.from_numpy(v)
    return out


def bn_gamma_spread_variant(sd, decades: float, seed: int = 11):
    """The checkpoint `sd` with CHANNEL DISPARITY inside the HourGlass2's tensors: four of its BatchNorms (the ones whose outputs feed
    convolutions only) get their affine scaled per channel by g_c = 10^U(-decades/2, +decades/2), and every consumer's weights of that
    input channel are divided by g_c - the same function in exact arithmetic (BN is the last op of its block: conv -> ReLU -> BN,
    network.py:10-28), but tensors whose channels differ by up to `decades` orders of magnitude, as trained BN gammas make them.  What
    it probes: formats that share one scale over a 32-channel block (the MX fp6 planes) or one exponent per tensor.  Test data generation."""
    import torch

    rs = np.random.RandomState(seed)
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    plan = [("enhanceNet.inConv.conv.2", 64, [("enhanceNet.down1.conv.0.weight", 0), ("enhanceNet.up1.combine.weight", 64)]),
            ("enhanceNet.down1.conv.4", 128, [("enhanceNet.down2.conv.0.weight", 0), ("enhanceNet.up2.combine.weight", 128)]),
            ("enhanceNet.up2.conv2.4", 128, [("enhanceNet.up1.conv1.weight", 0)]),
            ("enhanceNet.up1.conv2.4", 64, [("enhanceNet.outConv.weight", 0)])]
    for bn, ch, consumers in plan:
        g = torch.from_numpy((10.0 ** rs.uniform(-decades / 2, decades / 2, ch)).astype(np.float32))
        out[bn + ".weight"] = out[bn + ".weight"] * g
        out[bn + ".bias"] = out[bn + ".bias"] * g
        for key, off in consumers:
            w = out[key].clone()
            w[:, off:off + ch] = w[:, off:off + ch] / g[None, :, None, None]
            out[key] = w
    return out
