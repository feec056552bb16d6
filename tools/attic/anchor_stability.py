#!/usr/bin/env python
"""How often do the anchors (k-means on the encoder output -> per-cluster argmax) of a conv precision mode differ from the
fp32 CPU oracle / from the f16x3 mode?  GPU box.   python tools/anchor_stability.py [--n256 64] [--n512 8]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.gamut import gamut_points
from disentangledcolorization_amd.model import AnchorColorProb
from oracle.disco_ref import DiscoOracle

ap = argparse.ArgumentParser()
ap.add_argument("--n256", type=int, default=64)
ap.add_argument("--n512", type=int, default=8)
ap.add_argument("--oracle", type=int, default=1)
ap.add_argument("--modes", default="f16x3,mx8")
ap.add_argument("--seed", type=int, default=1000, help="input seed of the 256^2 set (the 512^2 set uses seed + 1000)")
args = ap.parse_args()
sd = synth.synth_state_dict(130)
models = {}
for prec in args.modes.split(","):
    m = AnchorColorProb(n_clusters=8, enhanced=True, precision=prec, init_weights=False)
    m.load_state_dict(sd); models[prec] = m.cuda().eval()
torch.set_num_threads(min(os.cpu_count() or 1, 32))
for (n, size, seed) in ((args.n256, 256, args.seed), (args.n512, 512, args.seed + 1000)):
    if n <= 0: continue
    gray, ab = synth.synth_inputs(n, size, size, seed=seed)
    res = {}
    for prec, m in models.items():
        np.random.seed(130); torch.manual_seed(130)
        out = m(gray.cuda(), ab.cuda(), True, 0); torch.cuda.synchronize()
        res[prec] = [o.cpu() for o in out]
    if args.oracle:
        t0 = time.time()
        np.random.seed(130); torch.manual_seed(130)
        res["oracle"] = DiscoOracle(sd, gamut_points(), n_clusters=8).forward(gray, ab)
        print("oracle %d x %d^2 in %.0fs" % (n, size, time.time() - t0), flush=True)
    ref = "oracle" if args.oracle else "f16x3"
    for prec in models:
        diff = [i for i in range(n) if not torch.equal(res[prec][5][i], res[ref][5][i])]
        same = [i for i in range(n) if i not in diff]
        e_ab = max((res[prec][2][i] - res[ref][2][i]).abs().max().item() for i in same) if same else float("nan")
        e_pal = (res[prec][0] - res[ref][0]).abs().max().item()
        print(f"{size}^2 x{n}: {prec:6s} vs {ref}: anchors differ in {len(diff)} images {diff[:8]}; max|pal| {e_pal:.2e}; max|ab| over anchor-identical images {e_ab:.2e}", flush=True)
