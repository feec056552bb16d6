#!/usr/bin/env python
"""Anchor exactness as a NUMBER, and its cause (round-5 verdict, weak item 1 / next item 2).

The north_star asks for bit-exact anchor indices.  Anchors are a discrete function (k-means on the wild-path encoder output, then a
per-cluster argmax: clusterkit.py:167-206, anchor_gen.py:96-101) of features that ANY implementation reproduces only to fp32 rounding,
so an image whose k-means passes through a near-tie can be decided differently by two exact-fp32 implementations that merely add in
another order.  This tool measures, image by image, on the same inputs and the same k-means rows:

    A  the CPU oracle, fp32, oneDNN convolutions                       (= the reference's arithmetic on this box: the pinned oracle)
    B  the CPU oracle, fp32, oneDNN OFF (ATen's im2col + sgemm convs)  (exact fp32, ANOTHER summation order: the control)
    C  the CPU oracle evaluated in fp64 (k-means itself in fp32)       (what the network "means")
    H  the HIP path (f16x3 on the anchor-deciding stacks)

and reports  rate(H != A)  next to  rate(B != A)  and  rate(C != A),  plus the overlap: of the images H decides differently from A,
how many does the fp32 control B or the fp64 evaluation C ALSO decide differently (= images that are near-ties for every implementation).

    G  the oracle's code on torch-ROCm tensors, MIOpen off (im2col + rocBLAS gemm convolutions): exact fp32 in a third order, on the GPU
    D  the same in fp64
    X  the oracle itself (A's code and settings) on another host CPU: oneDNN picks its kernels and its reduction split by ISA and thread count

Workers write one small file per (set, chunk of 16 images, variant) as they go (tools/anchor_study.sh starts eight CPU workers and one GPU
worker side by side); `report` merges what exists:
    python tools/anchor_study.py run --variants A --part 3/8 --threads 32 --dir gpurun_out/anchor_study
    python tools/anchor_study.py run --variants HGD --dir gpurun_out/anchor_study
    python tools/anchor_study.py report --dir profiles/r06_anchor_study --out profiles/r06_anchor_mismatch.json
The H files are re-made whenever a kernel source changes (seconds); the report's JSON carries bench.py's source_hash() and feeds the
bench line's `anchor_mismatch_rate`.  Only the encoder half of the oracle runs (segnet + repnet + pooling + wild path + k-means): the
anchors do not depend on the rest."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

K = 8
CHUNK = 16
MF = 20 * K


def parse_sets(s):
    out = []
    for part in s.split(","):
        hw, n = part.split(":")
        h, w = hw.split("x")
        out.append((int(h), int(w), int(n)))
    return out


def chunk_inputs(set_id, chunk, h, w):
    """CHUNK images of one set: seeded per chunk (and per image size: set_id is only a label), so that any worker can make its own."""
    from disentangledcolorization_amd import synth
    set_id = (h * 31 + w) % 89
    gray, ab = synth.synth_inputs(CHUNK, h, w, seed=100000 + 10000 * set_id + chunk)
    l = (h // 16) * (w // 16)
    idx = np.stack([np.random.RandomState(7000000 + 100000 * set_id + 100 * chunk + i).choice(l, K, replace=False) for i in range(CHUNK)]).astype(np.int32)
    fb = np.random.RandomState(9000000 + 1000 * set_id + chunk).randint(0, l, (CHUNK, MF)).astype(np.int32)
    return gray, ab, idx, fb


def variant_anchors(variant, gray, ab, idx, fb, cache={}):
    """Anchors of one chunk under one evaluation of the reference's arithmetic (module docstring).  A / B: CPU; G / D: torch on the GPU with
    MIOpen off (ATen's im2col + rocBLAS gemm convolutions: exact fp32 products and sums in yet another order / the same in fp64)."""
    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.gamut import gamut_points
    from oracle import disco_ref as R
    if "sd" not in cache:
        cache["sd"] = synth.synth_state_dict(130)
    key = variant
    if key not in cache:
        sd = cache["sd"]
        if variant in "CD":
            sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
        if variant in "GD":
            sd = {k: v.cuda() for k, v in sd.items()}
        cache[key] = R.DiscoOracle(sd, gamut_points(), n_clusters=K)
    o = cache[key]
    if variant in "CD":
        gray, ab = gray.double(), ab.double()
    with torch.no_grad():
        if variant in "GD":
            with torch.backends.cudnn.flags(enabled=False):
                aff, feats, src, pos, spix_ab, sizes = o.tokens(gray.cuda(), ab.cuda())
                enc = R.encoder_stack(o.sd, "wildpath", src, pos.to(src.device))
            enc, sizes = enc.cpu(), sizes.cpu()
        elif variant == "B":
            with torch.backends.mkldnn.flags(enabled=False):
                aff, feats, src, pos, spix_ab, sizes = o.tokens(gray, ab)
                enc = R.encoder_stack(o.sd, "wildpath", src, pos)
        else:
            aff, feats, src, pos, spix_ab, sizes = o.tokens(gray, ab)
            enc = R.encoder_stack(o.sd, "wildpath", src, pos)
        mask, info = o.anchors(enc.float(), sizes.float(), idx, [list(r) for r in fb])
    return info["anchor"].numpy().astype(np.int16)


def hip_anchors(m, gray, ab, idx, fb, l):
    out, ev = m.forward_once(gray.cuda(), ab.cuda(), True, 0, idx, None, fb.reshape(-1), np.arange(CHUNK, dtype=np.int64) * MF, True)
    mask = out[5].reshape(CHUNK, l).cpu().numpy()
    H = np.zeros((CHUNK, K), np.int16)
    for i in range(CHUNK):                           # the hint mask is the anchors as a multiset: expand it back, sorted
        toks = np.repeat(np.arange(l), mask[i].astype(np.int64))
        assert len(toks) == K
        H[i] = toks
    return H


def load_all(dirs, h, w, v):
    """{chunk: anchors (CHUNK, K)} of one set and variant from loose chunk files (<dir>/HxW_cNNNN_V.npy) and packed stores (<dir>/HxW_V.npz)."""
    import glob
    out = {}
    for d in dirs:
        pk = os.path.join(d, "%dx%d_%s.npz" % (h, w, v))
        if os.path.exists(pk):
            z = np.load(pk)
            for c, a in zip(z["chunks"], z["anchors"]):
                out[int(c)] = a.astype(np.int32)
        for path in glob.glob(os.path.join(d, "%dx%d_c*_%s.npy" % (h, w, v))):
            out[int(os.path.basename(path).split("_c")[1][:4])] = np.load(path).astype(np.int32)
    return out


def stage_pack(args):
    """Loose chunk files of --dir (+ what --store already holds) -> one small npz per (set, variant) under --store (tracked: profiles/)."""
    os.makedirs(args.store, exist_ok=True)
    for (h, w, n) in parse_sets(args.sets):
        for v in args.variants:
            tab = load_all([args.store, args.dir], h, w, v)
            if tab:
                cs = sorted(tab)
                np.savez_compressed(os.path.join(args.store, "%dx%d_%s.npz" % (h, w, v)), chunks=np.asarray(cs, np.int32),
                                    anchors=np.stack([tab[c] for c in cs]).astype(np.int16))
                print("%dx%d %s: %d chunks" % (h, w, v, len(cs)))


def stage_run(args):
    """One worker: the chunks c with c % of == part of every set, the variants asked for, ONE FILE PER (set, chunk, variant) written as soon
    as it exists - a run that is cut short keeps what it has (the report stage takes what it finds)."""
    part, of = (int(v) for v in args.part.split("/"))
    torch.set_num_threads(args.threads)
    os.makedirs(args.dir, exist_ok=True)
    m = None
    t0 = time.time()
    for (h, w, n) in parse_sets(args.sets):
        l = (h // 16) * (w // 16)
        have = {v: set(load_all([args.store], h, w, v)) for v in args.variants if v not in args.redo}
        for c in range(n // CHUNK):
            if c % of != part:
                continue
            inputs = None
            for v in args.variants:
                if v in args.limit and c >= args.limit[v]:
                    continue
                path = os.path.join(args.dir, "%dx%d_c%04d_%s.npy" % (h, w, c, v))
                if os.path.exists(path) or c in have.get(v, ()):
                    continue
                if inputs is None:
                    inputs = chunk_inputs(0, c, h, w)
                if v == "H":
                    if m is None:
                        from disentangledcolorization_amd import synth
                        from disentangledcolorization_amd.model import AnchorColorProb
                        m = AnchorColorProb(n_clusters=K, enhanced=True, init_weights=False)
                        m.load_state_dict(synth.synth_state_dict(130))
                        m = m.cuda().eval()
                        m.range_checks = 0
                    a = hip_anchors(m, *inputs, l)
                else:
                    a = variant_anchors(v, *inputs)
                np.save(path + ".tmp.npy", a)
                os.replace(path + ".tmp.npy", path)
            if c % (of * 8) == part:
                print("[%s %d/%d] %dx%d chunk %d, %.0f s" % (args.variants, part, of, h, w, c, time.time() - t0), flush=True)


def same_set(a, b):
    """Per image: do two (n,K) anchor tables name the same anchors (as the hint mask sees them: a multiset per image)?"""
    return np.all(np.sort(a, 1) == np.sort(b, 1), 1)


NAMES = {"A": "oracle (CPU fp32, oneDNN)", "B": "CPU fp32, oneDNN off (im2col + sgemm)", "C": "CPU fp64", "G": "torch-ROCm fp32 (im2col + rocBLAS)",
         "D": "torch-ROCm fp64", "H": "HIP path (f16x3 anchor stacks)",
         "X": "the oracle ITSELF on another host (the GPU box's CPU, 32 threads, instead of the build container's 8)"}


def stage_report(args):
    import glob
    import bench
    report = {"source_hash": bench.source_hash(), "precision": "default (f16x3 on SpixelNet + ColorProbNet)", "k": K, "variants": NAMES, "sets": {}}
    lines = []
    for (h, w, n) in parse_sets(args.sets):
        tab = {}
        for v in NAMES:
            t = load_all([args.store, args.dir], h, w, v)
            if t:
                tab[v] = t
        if "A" not in tab or "H" not in tab:
            continue
        rec = {}
        chunks_ah = sorted(set(tab["A"]) & set(tab["H"]))
        A = np.concatenate([tab["A"][c] for c in chunks_ah]); H = np.concatenate([tab["H"][c] for c in chunks_ah])
        hA = ~same_set(H, A)
        rec["n"] = int(len(A)); rec["hip_vs_oracle"] = int(hA.sum()); rec["rate_hip_vs_oracle"] = round(float(hA.mean()), 5)
        rec["hip_mismatch_images"] = np.nonzero(hA)[0].tolist()[:64]
        line = "%4dx%-4d | HIP != oracle: %d of %d (%.3f %%)" % (h, w, hA.sum(), len(A), 100 * hA.mean())
        near_any = np.zeros(len(A), bool); covered = np.ones(len(A), bool)
        for v in "XBCGD":
            if v not in tab:
                continue
            cs = sorted(set(tab[v]) & set(chunks_ah))
            sel = np.concatenate([np.arange(chunks_ah.index(c) * CHUNK, chunks_ah.index(c) * CHUNK + CHUNK) for c in cs])
            V = np.concatenate([tab[v][c] for c in cs])
            vA = ~same_set(V, A[sel]); vH = ~same_set(V, H[sel])
            both = vA & hA[sel]
            rec[v] = {"name": NAMES[v], "n": int(len(V)), "vs_oracle": int(vA.sum()), "rate_vs_oracle": round(float(vA.mean()), 5),
                      "vs_hip": int(vH.sum()), "hip_mismatches_in_these_images": int(hA[sel].sum()), "of_which_this_variant_also_differs_from_oracle": int(both.sum())}
            line += " | %s != oracle: %d of %d (%.3f %%), != HIP: %d; of HIP's %d mismatches here it flips %d" % (
                NAMES[v], vA.sum(), len(V), 100 * vA.mean(), vH.sum(), hA[sel].sum(), both.sum())
            if v in "GD" and len(V) == len(A):
                near_any |= vA
        if all(v in tab and len(tab[v]) == len(chunks_ah) for v in "GD"):
            rec["hip_mismatches_also_flipped_by_an_exact_evaluation"] = int((hA & near_any).sum())
            rec["hip_mismatches_flipped_by_hip_alone"] = int((hA & ~near_any).sum())
            line += " | HIP mismatches that an exact evaluation (fp32 other order or fp64) flips too: %d, HIP alone: %d" % ((hA & near_any).sum(), (hA & ~near_any).sum())
        report["sets"]["%dx%d" % (h, w)] = rec
        lines.append(line)
        print(line, flush=True)
    report["table"] = lines
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("stage", choices=["run", "report", "pack"])
    ap.add_argument("--sets", default="256x256:2048,512x512:512,768x512:512")
    ap.add_argument("--variants", default="A", help="letters of A B C (CPU) G D (torch on the GPU) H (the HIP path)")
    ap.add_argument("--limit", default="", help="e.g. B:16,C:4 - at most that many chunks (of %d images) per set for a variant" % CHUNK)
    ap.add_argument("--part", default="0/1", help="k/n: this worker takes the chunks c with c %% n == k")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--dir", default=os.path.join(REPO, "gpurun_out", "anchor_study"))
    ap.add_argument("--store", default=os.path.join(REPO, "profiles", "r06_anchor_study"), help="packed results (tracked); `run` skips the chunks it holds")
    ap.add_argument("--redo", default="", help="variants to compute again although the store has them (H after a kernel change)")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "anchor_study", "anchor_mismatch.json"))
    a = ap.parse_args()
    a.limit = {p.split(":")[0]: int(p.split(":")[1]) for p in a.limit.split(",") if p}
    {"run": stage_run, "report": stage_report, "pack": stage_pack}[a.stage](a)
