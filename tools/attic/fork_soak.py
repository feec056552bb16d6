"""Concurrency soak of the small-batch fork (GPU box): forwards of 1-3 images on 6 streams at once, every result compared bit for bit with the serial one."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
sd = synth.synth_state_dict(130)
m = AnchorColorProb(n_clusters=8, enhanced=True, init_weights=False); m.load_state_dict(sd); m = m.cuda().eval()
m.range_checks = 0
S = 6
batches = []
for i in range(S):
    n = 1 + i % 3
    g, a = synth.synth_inputs(n, 256, 256, seed=200 + i)
    idx = np.stack([np.random.RandomState(300 + 10 * i + j).choice(256, 8, replace=False) for j in range(n)]).astype(np.int32)
    batches.append((g.cuda(), a.cuda(), idx))
want = [tuple(t.clone() for t in m.forward_once(g, a, True, 0, idx, None, None, None, False)[0]) for g, a, idx in batches]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(S)]
bad = 0
for rep in range(150):
    outs = []
    for i in range(S):
        with torch.cuda.stream(streams[i]):
            g, a, idx = batches[i]
            outs.append(m.forward_once(g, a, True, 0, idx, None, None, None, False)[0])
    torch.cuda.synchronize()
    for i in range(S):
        if not all(torch.equal(x, y) for x, y in zip(outs[i], want[i])): bad += 1
print("forked small forwards on", S, "streams, 150 rounds:", bad, "of", 150 * S, "differ from the serial result")
