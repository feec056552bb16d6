import os, sys, ctypes as C, torch
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import gpu_helpers as H
from disentangledcolorization_amd import _ffi, synth
from test_gpu_ops import _encoder_weights
L = _ffi.lib()
sp = lambda st: C.c_void_p(st.cuda_stream)
sd = synth.synth_state_dict(130)
wts = _encoder_weights(sd, "wildpath").to(H.DEV)
S = int(os.environ.get("NSTREAMS", "8")); n, l = 8, 256
NCONV = int(os.environ.get("NCONV", "6")); USE_POOL = int(os.environ.get("USE_POOL", "1"))
class Chain:
    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.x = H.to_act(torch.randn(n, 256, 64, 64, generator=g))
        w = torch.randn(256, 256, 3, 3, generator=g) * (2.0 / (9 * 256)) ** 0.5
        self.packed = H.pack_conv(w); self.bias = torch.zeros(256, device=H.DEV)
        self.d = _ffi.ConvDesc(n, 64, 64, 256, 0, 0, 0, 256, 1, _ffi.ACT_RELU, 0.0, _ffi.PREC_F16X3, 0, 0, 0)
        self.bufs = [torch.empty(2, n, 64, 64, 256, device=H.DEV, dtype=torch.float16) for _ in range(2)]
        self.feat = torch.randn(n, 64, 256, 256, generator=g).to(H.DEV)
        self.prob = torch.softmax(torch.randn(n, 9, 256, 256, generator=g), 1).to(H.DEV)
        self.tok0 = torch.randn(n, l, 64, generator=g).to(H.DEV)
        self.pos = torch.randn(l, 64, generator=g).to(H.DEV)
        self.pooled = torch.empty(n, 64, 16, 16, device=H.DEV); self.conf = torch.empty(n, 1, 16, 16, device=H.DEV)
        self.pws = torch.empty(n * 256 * 9 * 66 * 4, dtype=torch.uint8, device=H.DEV)
        self.ews = torch.empty(n * l * 704 * 4, device=H.DEV, dtype=torch.uint8)
        self.out = torch.empty(n, l, 64, device=H.DEV)
    def run(self, st):
        src = self.x
        for i in range(NCONV):
            dst = self.bufs[i & 1]
            _ffi.check(L.disco_op_conv3x3(C.byref(self.d), _ffi.ptr(src), None, _ffi.ptr(self.packed), _ffi.ptr(self.bias), None, None, None, _ffi.ptr(dst), sp(st)))
            src = dst
        tok = self.tok0
        if USE_POOL:
            _ffi.check(L.disco_op_poolfeat(_ffi.ptr(self.feat), _ffi.ptr(self.prob), _ffi.ptr(self.pooled), _ffi.ptr(self.conf), None, n, 64, 256, 256, 16, _ffi.ptr(self.pws), self.pws.numel(), sp(st)))
            with torch.cuda.stream(st):
                tok = self.pooled.flatten(2).transpose(1, 2).contiguous()
        _ffi.check(L.disco_op_encoder_stack(_ffi.ptr(tok), _ffi.ptr(self.pos), _ffi.ptr(wts), _ffi.ptr(self.out), n, l, _ffi.ptr(self.ews), self.ews.numel(), sp(st)))
        self._keep = tok
chains = [Chain(i) for i in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
refs = []
for c, st in zip(chains, streams):
    c.run(st); torch.cuda.synchronize(); refs.append((c.out.clone(), c.bufs[(NCONV - 1) & 1].clone() if NCONV else None))
bad = {"enc": 0, "conv": 0}
for step in range(int(os.environ.get("STEPS", "40"))):
    for c, st in zip(chains, streams): c.run(st)
    torch.cuda.synchronize()
    for i, c in enumerate(chains):
        if not torch.equal(c.out, refs[i][0]):
            bad["enc"] += 1
            if bad["enc"] <= 5:
                d = (c.out - refs[i][0]).abs(); print(f"step {step} stream {i}: encoder output differs, max {d.max().item():.3e}, rows {torch.nonzero(d.flatten(1).max(1)[0] > 0).flatten().tolist()}", flush=True)
        if NCONV and not torch.equal(c.bufs[(NCONV - 1) & 1], refs[i][1]): bad["conv"] += 1
print(f"streams {S} convs {NCONV} pool {USE_POOL}: mismatches {bad}")
