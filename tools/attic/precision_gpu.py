"""Measured end-to-end error of each precision mode against the fp32 oracle (test infrastructure: imports oracle/ as the checker).

For several seeded inputs / sizes and the stress checkpoints of tests/test_gpu_forward.py, prints max|ab| (fp32 ab/110 units;
the parity bar is 1e-3) and whether the anchors are identical, per precision mode.

    python tools/precision_gpu.py [--modes mx6,mx8,x2q,f16x3]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle.disco_ref as R  # noqa: E402
from disentangledcolorization_amd import synth  # noqa: E402
from disentangledcolorization_amd.gamut import gamut_points  # noqa: E402
from disentangledcolorization_amd.model import AnchorColorProb  # noqa: E402
from test_gpu_forward import _stress_variant  # noqa: E402


def seed(s):
    np.random.seed(s); torch.manual_seed(s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="mx6,mx8,x2q,f16x3")
    ap.add_argument("--weights", default="", help="e.g. t4,t3: Student-t redraws of the 3x3 conv weights instead of the usual cases; run once more with "
                    "DISCO_MX6_ROW_SCALE=1 in the environment for round 3's one-exponent-per-row fp6 weight scaling")
    ap.add_argument("--gamma", default="", help="e.g. 1,2,3: HourGlass2 BN affines spread per channel over that many decades, compensated in the consumers' weights "
                    "(synth.bn_gamma_spread_variant): channel disparity inside the block-scaled tensors")
    ap.add_argument("--quick", action="store_true", help="three inputs on the synthetic checkpoint and one stress checkpoint")
    args = ap.parse_args()
    torch.set_num_threads(min(32, os.cpu_count()))       # (one thread per core of a 256-core host thrashes: minutes per oracle forward)
    base = synth.synth_state_dict(130)
    q = gamut_points()
    cases = [("synth s%d %dx%d" % (s, h, w), base, s, n, h, w) for s, n, h, w in
             ((100, 2, 128, 128), (101, 2, 128, 128), (102, 2, 128, 128), (103, 1, 256, 256), (104, 1, 192, 320), (105, 2, 64, 96))]
    cases += [(which, _stress_variant(base, which), 19, 2, 128, 128) for which in
              ("repnet_x256", "repnet_x1_256", "sn_sigma_16", "enhance_skip_x256")]
    if args.quick:
        cases = cases[:2] + cases[4:5] + cases[-1:]
    if args.weights:
        # heavy-tailed checkpoints (synth.student_t_variant): HourGlass2 alone (what the fp6 weight operands see) and with ColorProbNet
        cases = []
        for tag in args.weights.split(","):
            df = float(tag[1:])
            for wseed in (7, 8):
                cases.append(("%s enhanceNet w%d" % (tag, wseed), synth.student_t_variant(base, df, wseed, ("enhanceNet.",)), 19, 2, 128, 128))
            cases.append(("%s repnet+enhanceNet" % tag, synth.student_t_variant(base, df, 7), 19, 2, 128, 128))
        cases.append(("gaussian (synth 130)", base, 19, 2, 128, 128))
    if args.gamma:
        cases = [("gaussian (synth 130)", base, 19, 2, 128, 128)]
        for d in args.gamma.split(","):
            cases.append(("BN gamma spread %s decades" % d, synth.bn_gamma_spread_variant(base, float(d)), 19, 2, 128, 128))
    worst = {}
    models = {}

    def model(mode, sd):
        if (mode, id(sd)) not in models:
            os.environ["DISCO_PRECISION"] = mode
            m = AnchorColorProb(n_clusters=8, enhanced=True, init_weights=False)
            m.load_state_dict(sd)
            models[(mode, id(sd))] = m.cuda().eval()
        return models[(mode, id(sd))]

    for name, sd, s, n, h, w in cases:
        gray, ab = synth.synth_inputs(n, h, w, seed=s)
        seed(130); want = R.DiscoOracle(sd, q, n_clusters=8).forward(gray, ab)
        line = f"{name:24s} (|ab_ref| max {want[2].abs().max().item():.2f} std {want[2].std().item():.2f})"
        for mode in args.modes.split(","):
            m = model(mode, sd)
            seed(130); out = m(gray.cuda(), ab.cuda(), True, 0)
            torch.cuda.synchronize()
            e = (out[2].cpu().double() - want[2].double()).abs().max().item()
            same = torch.equal(out[5].cpu(), want[5])
            worst[mode] = max(worst.get(mode, 0.0), e)
            ar, disp = m.enhance_arithmetic()
            line += f"  {mode} {e:.2e}{'' if same else ' ANCHORS DIFFER'}" + (f" [HourGlass2 on {ar}, block disparity {disp:.1f}" + (f", levelled from {m.equalised_from():.1f}" if m.equalised_from() else "") + "]" if mode in ("mx6", "x2q") else "")
        print(line, flush=True)
    print("worst: " + "  ".join(f"{k} {v:.2e}" for k, v in worst.items()))


if __name__ == "__main__":
    main()
