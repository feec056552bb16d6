"""Does the one-image latency depend on what the context ran before (bench.py measures it after batch-64 forwards: 1.53-1.60 ms where
tools/small_batch_latency.py on a fresh model says 1.48)?   python tools/latency_state_probe.py   (GPU box)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
import bench

sd = synth.synth_state_dict(130)
m = AnchorColorProb(n_clusters=8, enhanced=True, init_weights=False); m.load_state_dict(sd); m = m.cuda().eval()
g64, a64 = synth.synth_inputs(64, 256, 256, seed=5)
g64, a64 = g64.cuda(), a64.cuda()
g1, a1 = g64[:1].contiguous(), a64[:1].contiguous()
sync = torch.cuda.synchronize
def lat(tag):
    print("%-46s median %.3f ms" % (tag, bench.single_image_latency(m, g1, a1, sync)), flush=True)
lat("fresh model")
lat("again")
for _ in range(5):
    np.random.seed(1); m(g64, a64, True, 0)
sync()
lat("after five 64-image forwards")
lat("again")
time.sleep(2.0)
lat("after 2 s idle")
m.set_profiling(2); np.random.seed(1); m(g64, a64, True, 0); sync(); m.set_profiling(0)
lat("after a profiled 64-image forward")
