import math, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch, torch.nn.functional as F
import gpu_helpers as H
from disentangledcolorization_amd import _ffi
LO, Q = 1, 2
def run(name, x, wt, b, **kw):
    want = F.conv2d(x.double(), wt.double(), b.double(), stride=kw.get("stride", 1), padding=1)
    out, sat = H.conv3x3_mx(H.to_act_mx(x), wt, b, out_planes=LO, **kw)
    got = out.read(0).cpu().double()
    err = (got - want).abs()
    print(f"{name}: max err {err.max().item():.3e} (max|want| {want.abs().max().item():.3e}) sat {sat}; err by channel-block of 8: "
          + " ".join(f"{err[:, c:c+8].max().item():.1e}" for c in range(0, want.shape[1], 8)))
    return got, want
g = torch.Generator().manual_seed(1)
# 1) exactly representable operands: only the fp16 main product contributes
x = torch.randint(-4, 5, (1, 32, 8, 8), generator=g).float()
w = torch.randint(-2, 3, (32, 32, 3, 3), generator=g).float() / 4
run("exact-int cin32 cout32 8x8", x, w, torch.zeros(32))
x = torch.randint(-4, 5, (2, 64, 16, 32), generator=g).float()
w = torch.randint(-2, 3, (64, 64, 3, 3), generator=g).float() / 4
run("exact-int cin64 cout64 16x32", x, w, torch.zeros(64))
# 2) centre-tap identity weights: out = x
w = torch.zeros(32, 32, 3, 3); w[torch.arange(32), torch.arange(32), 1, 1] = 1
x = torch.randn(1, 32, 8, 8, generator=g)
got, want = run("identity randn", x, w, torch.zeros(32))
# 3) weights with lo parts only (x exactly representable): tests the wl8 x a8 half
x = torch.randint(-4, 5, (1, 32, 8, 8), generator=g).float()
w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
run("x exact, w random (wl8*a8 half)", x, w, torch.zeros(32))
# 4) x with lo parts, weights exact: tests the w8 x al8 half
x = torch.randn(1, 32, 8, 8, generator=g)
w = torch.randint(-2, 3, (32, 32, 3, 3), generator=g).float() / 4
run("x random, w exact (w8*al8 half)", x, w, torch.zeros(32))
x = torch.randn(1, 32, 8, 8, generator=g)
w = torch.randn(32, 32, 3, 3, generator=g) * 0.1
run("both random", x, w, torch.zeros(32))
