#!/usr/bin/env python
"""Certify-or-recompute, with numbers (round-3 verdict item 8): could a cheaper conv arithmetic on the anchor-deciding stacks be made
safe by flagging the images whose discrete decisions are close calls and recomputing only those in f16x3?

For every image: run the default (f16x3 anchor path), x2q and mx8all forwards on the GPU; recover the k-means input (the wild-path
encoder output) from the f16x3 pal_logit (pal_logit = W_mid enc, W_mid 313x64 of rank 64: least squares in float64) and the
superpixel sizes from the affinity output; replay the reference k-means (oracle kmeans_one's arithmetic, instrumented) and record
  m_km     = min over passes and tokens of (second-smallest - smallest squared distance)           [assignment margin]
  m_anchor = min over clusters of (best - second-best anchor score)                              [argmax margin]
Then: which images do x2q / mx8all decide differently from f16x3, what are their margins, and what fraction of ALL images a margin
threshold flags that catches every disagreement.    python tools/anchor_margins.py [--sets 1000,3000,5000,7000 --n 480]"""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disentangledcolorization_amd import synth
from disentangledcolorization_amd.model import AnchorColorProb
from oracle import disco_ref as R

ap = argparse.ArgumentParser()
ap.add_argument("--sets", default="1000,3000,5000,7000")
ap.add_argument("--n", type=int, default=480)
ap.add_argument("--modes", default="x2q,mx8all")
args = ap.parse_args()
K = 8
sd = synth.synth_state_dict(130)
modes = ["mx8"] + args.modes.split(",")          # ("mx8" and the default "mx6" share the f16x3 anchor path: identical anchors)
models = {}
for prec in modes:
    m = AnchorColorProb(n_clusters=K, enhanced=True, precision=prec, init_weights=False)
    m.load_state_dict(sd); m = m.cuda().eval(); m.range_checks = 0; models[prec] = m
W = sd["mid_word_prj.weight"].double()                      # (313, 64)
Wp = torch.linalg.pinv(W)                                   # (64, 313)


def margins(enc, sizes, init_idx):
    """Instrumented replay of kmeans_one + anchors_from_clusters (float32 like the reference)."""
    x = enc.float()
    cent = x[torch.as_tensor(init_idx, dtype=torch.long)].clone()
    m_km, passes = float("inf"), 0
    while True:
        dist = ((x[:, None, :] - cent[None, :, :]) ** 2.0).sum(-1)
        two = torch.topk(dist, 2, dim=1, largest=False)[0]
        m_km = min(m_km, float((two[:, 1] - two[:, 0]).min()))
        assign = torch.argmin(dist, 1)
        prev = cent.clone()
        for j in range(K):
            mem = x[assign == j]
            if mem.shape[0] == 0:
                return None                                  # empty-cluster images: not part of this study (none in these sets)
            cent[j] = mem.mean(0)
        passes += 1
        if torch.sqrt(((cent - prev) ** 2).sum(1)).sum() ** 2 < 1e-4 or passes >= 20:
            break
    onehot = (assign[None, :] == torch.arange(K)[:, None]).float()
    score = onehot + sizes[None, :] * 0.01
    top = torch.topk(score, 2, dim=1)[0]
    anchor = torch.argmax(score, 1)
    mask = torch.zeros(x.shape[0]); mask.scatter_add_(0, anchor, torch.ones(K))
    return m_km, float((top[:, 0] - top[:, 1]).min()), mask


rows = []          # (set, index, m_km, m_anchor, replay_ok, differs per mode...)
for seed in [int(s) for s in args.sets.split(",")]:
    for c0 in range(0, args.n, 64):
        nb = min(64, args.n - c0)
        gray_all, ab_all = synth.synth_inputs(args.n, 256, 256, seed=seed)
        gray, ab = gray_all[c0:c0 + nb], ab_all[c0:c0 + nb]
        np.random.seed(130 + c0)
        idx = np.stack([np.random.choice(256, K, replace=False) for _ in range(nb)]).astype(np.int32)
        outs = {}
        for prec, m in models.items():
            outs[prec] = [o.cpu() for o in m.forward_with_draws(gray.cuda(), ab.cuda(), True, 0, init_idx=idx)]
        pal = outs["mx8"][0].double().flatten(2)             # (n, 313, L)
        enc = torch.einsum("dc,ncl->nld", Wp, pal)           # (n, L, 64)
        sizes = R.spixel_size(outs["mx8"][3], 16).reshape(nb, -1)
        for i in range(nb):
            r = margins(enc[i], sizes[i], idx[i])
            if r is None:
                continue
            m_km, m_an, mask = r
            ok = bool(torch.equal(mask, outs["mx8"][5][i].flatten()))
            rows.append((seed, c0 + i, m_km, m_an, ok) + tuple(not torch.equal(outs[p][5][i], outs["mx8"][5][i]) for p in modes[1:]))
    print("set %d done (%d images so far)" % (seed, len(rows)), flush=True)

rows = np.array(rows, dtype=object)
n = len(rows)
km = np.array([r[2] for r in rows], float); an = np.array([r[3] for r in rows], float)
print("\n%d images (256x256, K=8 clustering anchors, synthetic checkpoint); reference decision = the default mode (f16x3 anchor path)" % n)
print("replay of the k-means from the recovered encoder output reproduces the GPU's anchors in %d of %d images" % (sum(bool(r[4]) for r in rows), n))
print("m_km quantiles     (1%% 5%% 25%% 50%%): %s" % np.array2string(np.quantile(km, [0.01, 0.05, 0.25, 0.5]), precision=3))
print("m_anchor quantiles (1%% 5%% 25%% 50%%): %s" % np.array2string(np.quantile(an, [0.01, 0.05, 0.25, 0.5]), precision=3))
for k, p in enumerate(modes[1:]):
    bad = np.array([bool(r[5 + k]) for r in rows])
    print("\n%s: anchors differ from the default mode in %d images (%.2f %%)" % (p, bad.sum(), 100.0 * bad.mean()))
    if not bad.any():
        continue
    for j in np.nonzero(bad)[0]:
        print("   set %d image %3d: m_km %.3e (rank %4d of %d)  m_anchor %.3e (rank %4d)" % (rows[j][0], rows[j][1], km[j], (km < km[j]).sum(), n, an[j], (an < an[j]).sum()))
    # a flag "m_km < t1 or m_anchor < t2" that catches every disagreement: the cheapest thresholds are the largest margins among the bad images
    best = None
    for t1 in sorted(set(km[bad])) + [0.0]:
        rest = bad & ~(km <= t1)
        t2 = an[rest].max() if rest.any() else -1.0
        rate = ((km <= t1) | (an <= t2)).mean()
        if best is None or rate < best[0]:
            best = (rate, t1, t2)
    print("   cheapest certificate that catches all of them: flag (m_km <= %.3e or m_anchor <= %.3e) -> %.1f %% of ALL images flagged" % (best[1], best[2], 100 * best[0]))
