import math, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import torch, torch.nn.functional as F
import gpu_helpers as H
torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
LO, Q = 1, 2
w = torch.zeros(32, 32, 3, 3); w[torch.arange(32), torch.arange(32), 1, 1] = 1
x = torch.zeros(1, 32, 8, 8)
for c in range(32): x[0, c] = c + torch.arange(64).reshape(8, 8) / 100.0
out, sat = H.conv3x3_mx(H.to_act_mx(x), w, torch.zeros(32), out_planes=LO)
got = out.read(0).cpu()
print("channel 0:\n", got[0, 0]); print("channel 1:\n", got[0, 1]); print("channel 9:\n", got[0, 9]); print("channel 17:\n", got[0, 17])
print("pixel (3,4) all channels:", got[0, :, 3, 4])
print("input roundtrip hi+al8 pixel(3,4):", H.to_act_mx(x, LO | Q).read(0)[0, :, 3, 4].cpu())
