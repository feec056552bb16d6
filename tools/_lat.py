import torch, numpy as np, time, sys, os
sys.path.insert(0, os.getcwd())
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
m = AnchorColorProb(n_clusters=8, enhanced=True).cuda().eval(); m.sync_kmeans_events=False; m.set_profiling(1)
for (n,h,w) in ((1,256,256),(2,256,256),(4,256,256),(8,256,256),(16,256,256),(64,256,256),(1,512,768),(8,512,768)):
    g, a = synth.synth_inputs(n, h, w, seed=1, ab_scale=0.3); g, a = g.cuda(), a.cuda()
    for _ in range(3):
        np.random.seed(1); m(g, a, True, 0)
    torch.cuda.synchronize()
    t0=time.perf_counter()
    for _ in range(10):
        np.random.seed(1); m(g, a, True, 0)
    torch.cuda.synchronize()
    dt=(time.perf_counter()-t0)/10*1e3
    st = {k_: round(ms,2) for k_, ms, _ in m.profile()}
    print((n,h,w), "%.2f ms/forward  %.0f img/s" % (dt, n/dt*1e3), "repnet", st["repnet"], "enhance", st["enhance"], "segnet", st["segnet"])
