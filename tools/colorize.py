#!/usr/bin/env python
"""End-to-end use of the drop-in on an MI355X: the flow of main/colorizer/inference.py:93-131 for one image list,
with every tensor op on the device (image decode/encode through PIL on the host).

    python tools/colorize.py --checkpt disco.pth.rar --out out_dir img1.png img2.jpg        # real DISCO checkpoint
    python tools/colorize.py --out out_dir img.png                                           # synthetic weights (plumbing)

Per image: uint8 RGB -> resize to 256x256 (cv2.resize INTER_LINEAR semantics; the reference's default) or, with --no_resize,
pad to multiples of 16 -> Lab (fetch_data_from_rgb8) ->
AnchorColorProb.forward(gray, ab, True, T) -> Lab -> uint8 RGB, de-padded (normLabs_to_rgb8) -> PNG; with --anchors also
the anchor overlay (upfeat of hint_mask + mark_color_hints, inference.py:128-131)."""
import argparse
import os
import random
import sys

import numpy as np
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disentangledcolorization_amd import basic  # noqa: E402
from disentangledcolorization_amd.model import AnchorColorProb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("images", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--checkpt", default=None, help="torch checkpoint with a 'state_dict' entry (train_colorizer.py:109-113)")
    ap.add_argument("--n_clusters", type=int, default=8)
    ap.add_argument("--random_hint", action="store_true")
    ap.add_argument("--diverse", action="store_true")
    ap.add_argument("--hint2regress", action="store_true")
    ap.add_argument("--spix_pos", action="store_true")
    ap.add_argument("--anchors", action="store_true", help="also save the anchor overlay")
    ap.add_argument("--no_resize", action="store_true", help="keep the original size (padded to multiples of 16), inference.py:148")
    ap.add_argument("--resize_to", type=int, default=256, help="side of the resized input (the reference hard-codes 256, inference.py:33; its --psize is the superpixel size)")
    ap.add_argument("--psize", type=int, default=16, choices=[8, 16, 32], help="superpixel size (inference.py:147; inputs are padded / resized to multiples of 16, so 32 needs sizes that are multiples of 32)")
    ap.add_argument("--seed", type=int, default=130)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    # inference.py:58-60
    np.random.seed(args.seed); torch.manual_seed(args.seed); random.seed(args.seed)
    model = AnchorColorProb(inChannel=1, outChannel=313, sp_size=args.psize, d_model=64, use_dense_pos=True, spix_pos=args.spix_pos,
                            learning_pos=False, n_clusters=args.n_clusters, random_hint=args.random_hint,
                            hint2regress=args.hint2regress, enhanced=True, init_weights=args.checkpt is None)
    if args.checkpt:
        model.load_state_dict(torch.load(args.checkpt, map_location="cpu")["state_dict"])      # strict (utils_train.py:151)
    model = model.cuda().eval()
    for path in args.images:
        rgb8 = np.asarray(Image.open(path).convert("RGB"))
        gray, ab, _, (H, W) = basic.fetch_data_from_rgb8(rgb8, org_size=args.no_resize, psize=args.resize_to)
        if not args.no_resize:
            H, W = gray.shape[2], gray.shape[3]         # batch_depadding (inference.py:138-139): no crop unless --no_resize
        _, _, pred_ab, affinity, _, hint_mask = model(gray, ab, True, 2 if args.diverse else 0)
        stem = os.path.splitext(os.path.basename(path))[0]
        for i in range(pred_ab.shape[0]):
            lab = torch.cat((gray, pred_ab[i:i + 1]), 1)
            out8 = basic.normLabs_to_rgb8(lab, H, W)[0].cpu().numpy()
            Image.fromarray(out8).save(os.path.join(args.out, stem + ("-c%d" % i if args.diverse else "") + ".png"))
        if args.anchors and not args.diverse:
            gates = basic.upfeat(hint_mask, affinity, args.psize, args.psize)
            marked = basic.mark_color_hints(gray, pred_ab, gates, base_ABs=pred_ab)
            Image.fromarray(basic.normLabs_to_rgb8(marked, H, W)[0].cpu().numpy()).save(os.path.join(args.out, stem + "-anchors.png"))
        print("colorized", path, "(%dx%d)" % (W, H))
    clamped = model.saturation_count()
    if clamped:       # (the first forwards of a context re-calibrate by themselves; this catches later images)
        print("warning: %d fp8 activation values were clamped on these images - call model.calibrate(gray) on representative inputs" % clamped,
              file=sys.stderr)


if __name__ == "__main__":
    main()
