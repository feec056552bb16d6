#!/usr/bin/env python
"""Per-layer error contribution of the fp16+fp8-correction conv arithmetic on the ANCHOR path (CPU emulation, tools/precision_sim.py):
one conv layer at a time computes  w_h a_h + fp8(w_l) fp8(a) + fp8(w) fp8(a_l)  while every other layer is exact fp32; reported is
the deviation it leaves at pal_logit (= the encoder output the k-means anchors are decided on).  Question behind it: is the ~3e-5 of
the all-layers mode dominated by a few layers (then those could stay on f16x3 and the rest go fast)?  Answer: no - see
profiles/r02_per_layer_precision.txt.      python tools/per_layer_precision.py [--size 256]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import precision_sim as ps  # noqa: E402
import oracle.disco_ref as ref  # noqa: E402
from disentangledcolorization_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
args = ap.parse_args()
torch.set_num_threads(os.cpu_count())
sd = synth.synth_state_dict(130)
gray, ab = synth.synth_inputs(2, args.size, args.size, seed=100)
base = ps.run(lambda k, c: "x3", sd, gray, ab, 130)
keys = []


class Rec(ps.Emu):
    def __call__(self, sd_, key, x, stride=1):
        keys.append((key, x.shape[1]))
        return super().__call__(sd_, key, x, stride)


saved = ref.conv3x3
ref.conv3x3 = Rec(lambda k, c: "x3")
np.random.seed(130); torch.manual_seed(130)
ref.DiscoOracle(sd, ps.gamut_points(), n_clusters=8).forward(gray, ab)
ref.conv3x3 = saved
anchor_path = lambda k, c: (k.startswith("repnet") or k.startswith("segnet")) and c % 32 == 0 and c >= 64
layers = [k for k, c in keys if anchor_path(k, c)]
print(len(layers), "anchor-path conv layers with Cin % 32 == 0")
tot = 0.0
for key in layers:
    got = ps.run(lambda k, c, key=key: "mx8" if k == key else "x3", sd, gray, ab, 130)
    e = (got[0] - base[0]).abs().max().item()
    tot += e * e
    print(f"{key:28s} max|d pal_logit| = {e:.2e}", flush=True)
print("root-sum-square of the single-layer deviations: %.2e" % tot ** 0.5)
got = ps.run(lambda k, c: "mx8" if anchor_path(k, c) else "x3", sd, gray, ab, 130)
print("all of them together:                            %.2e" % (got[0] - base[0]).abs().max().item())
