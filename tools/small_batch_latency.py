"""Latency and back-to-back time of model.forward_once on small batches of 256x256 images (GPU box); DISCO_FORK_SEGNET=0/1 for the A/B of
profiles/r04_small_batch_fork_ab.txt.  Prints a hash of the six outputs per point."""
import os, sys, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
sd = synth.synth_state_dict(130)
m = AnchorColorProb(n_clusters=8, enhanced=True, init_weights=False); m.load_state_dict(sd); m = m.cuda().eval()
m.range_checks = 0
for n in (1, 2, 4, 8, 16):
    gray, ab = synth.synth_inputs(n, 256, 256, seed=5)
    gray, ab = gray.cuda(), ab.cuda()
    idx = np.stack([np.random.RandomState(i).choice(256, 8, replace=False) for i in range(n)]).astype(np.int32)
    def fwd():
        return m.forward_once(gray, ab, True, 0, idx, None, None, None, False)[0]
    for _ in range(5): o = fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): o = fwd(); torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / 100 * 1e3
    t0 = time.perf_counter()
    for _ in range(200): o = fwd()
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / 200 * 1e3
    h = hashlib.sha256(b"".join(t.cpu().numpy().tobytes() for t in o)).hexdigest()[:12]
    print(f"n={n:2d}: latency {lat:.3f} ms, back-to-back {thr:.3f} ms/forward = {n / thr * 1e3:.0f} img/s   outputs sha {h}")
