import os, sys, ctypes as C, numpy as np, torch
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import gpu_helpers as H
from disentangledcolorization_amd import _ffi, synth
from disentangledcolorization_amd.model import AnchorColorProb
from disentangledcolorization_amd.runner import global_draws, peek_randint
L = _ffi.lib()
sp = lambda st: C.c_void_p(st.cuda_stream)
m = AnchorColorProb(n_clusters=8, enhanced=True).cuda().eval(); m.sync_kmeans_events = False
gray, ab = synth.synth_inputs(64, 256, 256, seed=5); gray, ab = gray.cuda(), ab.cuda()
for _ in range(4): m(gray[:8], ab[:8], True, 0)
torch.cuda.synchronize()
n = 8
g = torch.Generator().manual_seed(0)
feat = torch.randn(n, 64, 256, 256, generator=g).to(H.DEV)
prob = torch.softmax(torch.randn(n, 9, 256, 256, generator=g), 1).to(H.DEV)
NB = int(os.environ.get("NBG", "7"))
bg_streams = [torch.cuda.Stream() for _ in range(NB)]
ts = torch.cuda.Stream()
def pool(st):
    pooled = torch.empty(n, 64, 16, 16, device=H.DEV); conf = torch.empty(n, 1, 16, 16, device=H.DEV)
    ws = torch.empty(n * 256 * 9 * 66 * 4, dtype=torch.uint8, device=H.DEV)
    _ffi.check(L.disco_op_poolfeat(_ffi.ptr(feat), _ffi.ptr(prob), _ffi.ptr(pooled), _ffi.ptr(conf), None, n, 64, 256, 256, 16, _ffi.ptr(ws), ws.numel(), sp(st)))
    return pooled
with torch.cuda.stream(ts): ref = pool(ts).clone()
torch.cuda.synchronize()
np.random.seed(130); idx, _ = global_draws(64, 256, 8, False); fs = peek_randint(256, m.max_fallback())
bad = 0; total = 0
for rnd in range(int(os.environ.get("ROUNDS", "12"))):
    keep = []
    for rep in range(3):
        for i, st in enumerate(bg_streams):
            with torch.cuda.stream(st):
                keep.append(m.forward_once(gray[8*i:8*i+8], ab[8*i:8*i+8], True, 0, idx[8*i:8*i+8], None, fs, None, False)[0])
    outs = []
    with torch.cuda.stream(ts):
        for _ in range(60): outs.append(pool(ts))
    torch.cuda.synchronize()
    for o in outs:
        total += 1
        if not torch.equal(o, ref):
            bad += 1
            if bad <= 4:
                d = (o != ref); print("pool differs: cells/channels", torch.nonzero(d.flatten(2).any(2)).tolist()[:8], flush=True)
print(f"background forwards on {NB} streams: {bad} of {total} pool runs differ")
