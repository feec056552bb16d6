#!/usr/bin/env python
"""Per-layer table of one forward (GPU box): every MFMA conv launch with its hipEvent duration, algorithmic
TFLOP/s and share of the step, plus the stage totals.   python tools/profile_layers.py [--batch 64]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disentangledcolorization_amd import synth  # noqa: E402
from disentangledcolorization_amd.model import AnchorColorProb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--precision", default=None)
args = ap.parse_args()
m = AnchorColorProb(n_clusters=8, enhanced=True, precision=args.precision).cuda().eval()      # None: the package default
m.sync_kmeans_events = False
m.set_profiling(2)
gray, ab = synth.synth_inputs(args.batch, args.size, args.size, seed=5)
gray, ab = gray.cuda(), ab.cuda()
acc = {}
reps = 3
for it in range(2 + reps):
    np.random.seed(130)
    m(gray, ab, True, 0)
    torch.cuda.synchronize()
    if it >= 2:
        for i, (k, ms, fl) in enumerate(m.conv_profile_entries()):
            a = acc.setdefault(i, [k, 0.0, fl]); a[1] += ms / reps
stages = m.profile()
tot = sum(ms for _, ms, _ in stages)
print("stage totals (ms):", {k: round(ms, 3) for k, ms, _ in stages}, "sum", round(tot, 2))
conv_total = sum(v[1] for v in acc.values())
print("conv launches: %d, total %.2f ms" % (len(acc), conv_total))
for i in sorted(acc):
    k, ms, fl = acc[i]
    print(f"{i:3d} {k:34s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF alg  {100 * ms / tot:5.1f}%")
