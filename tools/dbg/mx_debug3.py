import math, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch, torch.nn.functional as F
import gpu_helpers as H
torch.set_printoptions(linewidth=220, precision=3, sci_mode=False)
LO, Q = 1, 2
w = torch.zeros(32, 32, 3, 3); w[torch.arange(32), torch.arange(32), 1, 1] = 1
packed = H.pack_conv_mx(w)
for c0 in [0, 1, 2, 3, 4, 8, 16, 17, 31]:
    x = torch.zeros(1, 32, 8, 8); x[0, c0] = 1.0
    out, sat = H.conv3x3_mx(H.to_act_mx(x, sexp=0), w, torch.zeros(32), out_planes=LO, packed=packed)
    got = out.read(0).cpu()
    print("in ch", c0, "-> out channels (pixel 3,4):", [(c, round(v, 3)) for c, v in enumerate(got[0, :, 3, 4].tolist()) if v != 0])
# weights: single nonzero
for (co, ci) in [(0, 0), (0, 2), (2, 0), (5, 20), (20, 5)]:
    w = torch.zeros(32, 32, 3, 3); w[co, ci, 1, 1] = 1
    x = torch.zeros(1, 32, 8, 8)
    for c in range(32): x[0, c] = c + 1
    out, sat = H.conv3x3_mx(H.to_act_mx(x, sexp=0), w, torch.zeros(32), out_planes=LO)
    got = out.read(0).cpu()
    print("w[%d,%d]=1 -> out (pixel 3,4):" % (co, ci), [(c, round(v, 3)) for c, v in enumerate(got[0, :, 3, 4].tolist()) if v != 0])
