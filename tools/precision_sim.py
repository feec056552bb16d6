"""CPU emulation of candidate conv arithmetics on the oracle's network, end to end (design study for the conv kernel).

Every 3x3 conv of the oracle is replaced by an emulation of how the HIP kernel would compute it:
  x3    : (w_h + w_l)(a_h + a_l) minus lo*lo  ~ fp32 (what the round-1 kernel does; emulated as the fp32 conv)
  x1    : w_h a_h                                              (1 fp16 MFMA per product)
  wh    : w_h (a_h + a_l)                                      (2 fp16 MFMAs: weights rounded to fp16)
  ah    : (w_h + w_l) a_h                                      (2 fp16 MFMAs: activations rounded to fp16)
  mx8   : w_h a_h + q8(w_h) q8(a_l) + q8(w_l) q8(a_h)          (1 fp16 MFMA + 2 block-scaled fp8 e4m3 K=64 MFMAs)
          q8 = fp8 e4m3 with a power-of-two scale per 32 consecutive input channels (per pixel / per cout and tap)
  mx8u  : same with ONE power-of-two scale per tensor (per layer input, per layer weight)
  x2q   : w_h a_h + w_l a_h + q8(w) q8(a_l)    (2 fp16 MFMAs + 1 fp8: the weight residual exact, the activation residual in fp8)
  x2qw  : w_h a_h + w_h a_l + q8(w_l) q8(a_h)  (the mirror image)
  mx8w1 : w_h a_h + q8(w_l) q8(a_h)            (activation residual dropped altogether)
  mx6b / mx6u / mx6uh : as mx8 with fp6 e2m3 operands (the K=64 MFMA runs them in half the passes of fp8): MX block scales per 32
          channels / one scale per activation tensor / that with 2 bits of headroom
and the end-to-end deviation of pred_colors from the fp32 oracle and anchor agreement are reported.

    python tools/precision_sim.py [--size 128] [--seeds 4] [--modes mx8,mx8u,wh]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.disco_ref as ref  # noqa: E402
from disentangledcolorization_amd import synth  # noqa: E402
from disentangledcolorization_amd.gamut import gamut_points  # noqa: E402

F8 = torch.float8_e4m3fn


def split16(x):
    h = x.half().float()
    return h, (x - h).half().float()


def q8_block(x, dim):
    """fp8 e4m3 with a power-of-two scale per 32 consecutive elements along `dim` (sized to keep the block max <= 256)."""
    x = x.movedim(dim, -1)
    shp = x.shape
    c = shp[-1]
    pad = (-c) % 32
    if pad:
        x = F.pad(x, (0, pad))
    xb = x.reshape(*x.shape[:-1], -1, 32)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    e = torch.floor(torch.log2(amax)) - 7
    s = torch.exp2(e)
    q = (xb / s).to(F8).float() * s
    q = q.reshape(*x.shape)[..., :c]
    return q.reshape(shp).movedim(-1, dim)


def q6(x):
    """fp6 e2m3 (OCP MX: 1 sign, 2 exponent, 3 mantissa bits; max 7.5, subnormal step 0.125), round to nearest even, saturating."""
    ax = x.abs().clamp(max=7.5)
    e = torch.floor(torch.log2(ax.clamp_min(1.0)))            # 0 for the subnormal / first binade, up to 2
    step = torch.exp2(e - 3)
    return torch.sign(x) * torch.round(ax / step) * step        # torch.round = half to even


def q6_block(x, dim):
    """fp6 e2m3 with a power-of-two scale per 32 consecutive elements along `dim`: block max in (3.75, 7.5]."""
    x = x.movedim(dim, -1)
    shp = x.shape
    c = shp[-1]
    pad = (-c) % 32
    if pad:
        x = F.pad(x, (0, pad))
    xb = x.reshape(*x.shape[:-1], -1, 32)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    s = torch.exp2(torch.ceil(torch.log2(amax / 7.5)))
    q = q6(xb / s) * s
    q = q.reshape(*x.shape)[..., :c]
    return q.reshape(shp).movedim(-1, dim)


def q6_tensor(x, headroom_bits=0):
    amax = x.abs().max().clamp_min(1e-38)
    s = torch.exp2(torch.ceil(torch.log2(amax / 7.5)) + headroom_bits)
    return q6(x / s) * s


def q8_tensor(x, headroom_bits=3):
    amax = x.abs().max().clamp_min(1e-38)
    s = torch.exp2(torch.floor(torch.log2(amax)) - (8 - headroom_bits))
    return (x / s).clamp(-448, 448).to(F8).float() * s


class Emu:
    def __init__(self, mode_of):
        self.mode_of = mode_of
        self.cache = {}

    def __call__(self, sd, key, x, stride=1):
        mode = self.mode_of(key, x.shape[1])
        w = ref.conv_weight(sd, key)
        b = sd.get(key + ".bias")
        if mode == "x3":
            return F.conv2d(x, w, b, stride=stride, padding=1)
        wh, wl = split16(w)
        ah, al = split16(x)
        cv = lambda a, ww: F.conv2d(a, ww, None, stride=stride, padding=1)
        if mode == "x1":
            y = cv(ah, wh)
        elif mode == "wh":
            y = cv(ah + al, wh)
        elif mode == "ah":
            y = cv(ah, wh + wl)
        elif mode == "mx8":
            y = cv(ah, wh) + cv(q8_block(al, 1), q8_block(wh, 1)) + cv(q8_block(ah, 1), q8_block(wl, 1))
        elif mode == "x2q":      # weight-side correction exact in fp16, only the activation residual in fp8
            y = cv(ah, wh) + cv(ah, wl) + cv(q8_tensor(al), q8_tensor(w))
        elif mode == "x2qw":     # the mirror image: activation-side correction exact, weight residual in fp8
            y = cv(ah, wh) + cv(al, wh) + cv(q8_tensor(ah), q8_tensor(wl))
        elif mode == "mx8w1":    # only the weight residual corrected (fp8), the activation residual dropped: 3 units per 32 channels
            y = cv(ah, wh) + cv(q8_tensor(ah), q8_tensor(wl))
        elif mode == "mx6b":     # both corrections in fp6 e2m3 with MX block scales (per pixel / per cout and tap, 32 channels)
            y = cv(ah, wh) + cv(q6_block(al, 1), q6_block(wh, 1)) + cv(q6_block(ah, 1), q6_block(wl, 1))
        elif mode == "mx6w1":    # only the weight residual corrected (block-scaled fp6), the activation residual dropped
            y = cv(ah, wh) + cv(q6_block(ah, 1), q6_block(wl, 1))
        elif mode == "mx6u":     # fp6 with ONE scale per tensor (weights: per cout row)
            y = cv(ah, wh) + cv(q6_tensor(al), q6_block(wh, 1)) + cv(q6_tensor(ah), q6_block(wl, 1))
        elif mode == "mx6uh":    # ... with 2 bits of calibration headroom on the activation side
            y = cv(ah, wh) + cv(q6_tensor(al, 2), q6_block(wh, 1)) + cv(q6_tensor(ah, 2), q6_block(wl, 1))
        elif mode == "mx8u":
            y = cv(ah, wh) + cv(q8_tensor(al), q8_tensor(wh)) + cv(q8_tensor(ah), q8_tensor(wl))
        else:
            raise ValueError(mode)
        return y if b is None else y + b.view(1, -1, 1, 1)


def run(mode_of, sd, gray, ab, seed, k=8):
    saved = ref.conv3x3
    ref.conv3x3 = Emu(mode_of)
    try:
        np.random.seed(seed); torch.manual_seed(seed)
        out = ref.DiscoOracle(sd, gamut_points(), n_clusters=k).forward(gray, ab)
    finally:
        ref.conv3x3 = saved
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--seeds", type=int, default=4)
    ap.add_argument("--n", type=int, default=2)
    ap.add_argument("--modes", default="x1,wh,ah,mx8,mx8u")
    ap.add_argument("--scope", default="all", help="all | enhance | repnet : which convs get the mode (others x3)")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    sd = synth.synth_state_dict(130)
    for mode in args.modes.split(","):
        def mode_of(key, cin, mode=mode):
            if cin % 32 or cin < 64:
                return "x3"
            if args.scope == "enhance" and not key.startswith("enhanceNet"):
                return "x3"
            if args.scope == "repnet" and not key.startswith("repnet"):
                return "x3"
            if args.scope == "anchor" and not (key.startswith("repnet") or key.startswith("segnet")):
                return "x3"
            return mode
        worst, flips = 0.0, 0
        for s in range(args.seeds):
            gray, ab = synth.synth_inputs(args.n, args.size, args.size, seed=100 + s)
            base = run(lambda k_, c: "x3", sd, gray, ab, 130)
            got = run(mode_of, sd, gray, ab, 130)
            err = (got[2] - base[2]).abs().max().item()
            same = torch.equal(got[5], base[5])
            flips += 0 if same else 1
            worst = max(worst, err)
            print(f"  mode {mode:5s} seed {s}: max|d ab| = {err:.3e}  anchors {'identical' if same else 'DIFFER'}  pal {(got[0]-base[0]).abs().max().item():.2e}", flush=True)
        print(f"mode {mode:5s} scope {args.scope}: worst max|d ab| = {worst:.3e}, anchor flips in {flips}/{args.seeds} batches", flush=True)


if __name__ == "__main__":
    main()
