"""Forward time + stage split at the --no_resize sizes and mid batches (GPU box): python tools/forward_points.py (A/B switches through the environment)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
m = AnchorColorProb(n_clusters=8, enhanced=True).cuda().eval()
m.sync_kmeans_events = False; m.range_checks = 0
for n, h, w in [(1, 512, 768), (8, 512, 768), (16, 512, 512), (1, 1024, 1024), (16, 256, 256), (32, 256, 256), (64, 256, 256)]:
    g, a = synth.synth_inputs(n, h, w, seed=1, ab_scale=0.3)
    g, a = g.cuda(), a.cuda()
    for _ in range(3):
        np.random.seed(1); m(g, a, True, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        np.random.seed(1); m(g, a, True, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    m.set_profiling(1); np.random.seed(1); m(g, a, True, 0); torch.cuda.synchronize(); m.set_profiling(0)
    st = {k_: round(ms, 2) for k_, ms, _ in m.profile()}
    print("%3d x %4dx%-4d %8.2f ms  %s" % (n, h, w, dt, st), flush=True)
