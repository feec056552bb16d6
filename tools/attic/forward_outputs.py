"""One forward of 2 images under the library DISCO_HIP_LIB names: saves the six outputs (A/B of two builds: DISCO_HIP_LIB=... python tools/forward_outputs.py out.npz [oracle.npz])."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
sd = synth.synth_state_dict(130)
m = AnchorColorProb(n_clusters=8, enhanced=True, init_weights=False); m.load_state_dict(sd); m = m.cuda().eval()
gray, ab = synth.synth_inputs(2, 256, 256, seed=5)
np.random.seed(130); torch.manual_seed(130)
out = m(gray.cuda(), ab.cuda(), True, 0); torch.cuda.synchronize()
np.savez(sys.argv[1], **{"o%d" % i: o.cpu().numpy() for i, o in enumerate(out)})
if len(sys.argv) > 2:
    from disentangledcolorization_amd.gamut import gamut_points
    from oracle.disco_ref import DiscoOracle
    np.random.seed(130); torch.manual_seed(130)
    want = DiscoOracle(sd, gamut_points(), n_clusters=8).forward(gray, ab)
    np.savez(sys.argv[2], **{"o%d" % i: o.numpy() for i, o in enumerate(want)})
