#!/usr/bin/env python
"""One forward of N x 256x256 as 1 / 2 / 4 concurrent micro-batches (ShardedColorizer(micro_batches=...): slices of the batch on their own
HIP streams, joined at the end) - forwards issued back to back, synchronised latency and throughput (GPU box)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
from disentangledcolorization_amd.runner import ShardedColorizer
m = AnchorColorProb(n_clusters=8, enhanced=True).cuda().eval(); m.range_checks = 0
for n in (4, 8, 16, 32):
    g, a = synth.synth_inputs(n, 256, 256, seed=5); g, a = g.cuda(), a.cuda()
    ref = None
    for mb in (1, 2, 4):
        if mb > n // 2: continue
        r = ShardedColorizer.from_model(m, micro_batches=mb, exact_fallback=False)
        def f():
            np.random.seed(1); torch.manual_seed(1)
            return r.colorize(g, a, n, 0, gather=False)
        for _ in range(5): o = f()
        r.wait(); torch.cuda.synchronize()
        p = o[0].clone()
        if ref is None: ref = p
        same = torch.equal(p, ref)
        t0 = time.perf_counter()
        for _ in range(50): f(); r.wait(); torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 50 * 1e3
        t0 = time.perf_counter()
        for _ in range(100): f()
        r.wait(); torch.cuda.synchronize()
        thr = (time.perf_counter() - t0) / 100 * 1e3
        print("n=%2d micro_batches=%d: latency %.3f ms, back-to-back %.3f ms = %.0f img/s, identical %s" % (n, mb, lat, thr, n / thr * 1e3, same), flush=True)
