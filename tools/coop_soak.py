#!/usr/bin/env python
"""Soak of the several-workgroup k-means under concurrency (GPU box): forwards at the --no_resize sizes issued over two HIP streams
(runner.ShardedColorizer.pipeline: two forwards' k-means launches - workgroups that spin on each other - and the other forward's
persistent conv workgroups share the GPU), every result compared bit for bit with the one-stream forward.   python tools/coop_soak.py [iters]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
from disentangledcolorization_amd.runner import ShardedColorizer

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
m = AnchorColorProb(n_clusters=8, enhanced=True).cuda().eval()
m.range_checks = 0
for n, h, w in [(1, 512, 768), (4, 512, 768), (2, 1024, 1024), (3, 768, 512)]:
    g, a = synth.synth_inputs(n, h, w, seed=3, ab_scale=0.3)
    g, a = g.cuda(), a.cuda()
    r1 = ShardedColorizer.from_model(m, micro_batches=1, exact_fallback=False)
    np.random.seed(7); torch.manual_seed(7)
    ref_p, ref_m = r1.colorize(g, a, n, 0, gather=False)
    r1.wait(); torch.cuda.synchronize()
    ref_p, ref_m = ref_p.clone(), ref_m.clone()
    r2 = ShardedColorizer.from_model(m, micro_batches=1, exact_fallback=False)
    r2.pipeline = True
    outs = []
    for _ in range(iters):
        np.random.seed(7); torch.manual_seed(7)
        p, hm = r2.colorize(g, a, n, 0, gather=False)
        outs.append((p, hm))
    r2.wait(); torch.cuda.synchronize()
    bad = sum(1 for p, hm in outs if not (torch.equal(p, ref_p) and torch.equal(hm, ref_m)))
    print("%d x %dx%d: %d pipelined forwards, %d differ from the one-stream result; %d images went to the one-workgroup k-means" % (n, h, w, iters, bad, m.kmeans_fallback_count()), flush=True)
    assert bad == 0
print("ok")
