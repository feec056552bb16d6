#!/usr/bin/env python
"""Which stage's output depends on what else runs on the GPU?  (GPU box only; debugging aid.)

Runs the 64-image batch as M micro-batches on M streams (runner.py's pipelining) with disco_set_debug_checksums on: every
forward leaves a checksum per stage (every conv launch in order, tokens, encoder output, pal_logit).  A serialised pass gives
the reference rows; concurrent passes are compared with them per micro-batch and the first diverging stage is reported.

    python tools/stagger_probe.py [--micro 8] [--steps 12]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disentangledcolorization_amd import _ffi, synth  # noqa: E402
from disentangledcolorization_amd.model import AnchorColorProb  # noqa: E402
from disentangledcolorization_amd.runner import global_draws, peek_randint, shard_bounds  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--micro", type=int, default=8)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--fixed-out", type=int, default=0, help="1: every micro-batch writes into the same preallocated output tensors in every step")
args = ap.parse_args()
M, COLS = args.micro, 96
m = AnchorColorProb(n_clusters=8, enhanced=True).cuda().eval()
m.sync_kmeans_events = False
gray, ab = synth.synth_inputs(64, 256, 256, seed=5)
gray, ab = gray.cuda(), ab.cuda()
for _ in range(4):      # use up the range checks (they synchronise)
    m(gray[:8], ab[:8], True, 0)
torch.cuda.synchronize()
table = torch.zeros(M, COLS, dtype=torch.int64, device="cuda")
ctx = m._context(torch.device("cuda", 0))
NI = 64 // M
TB = NI * 256 * 64 * 4                      # bytes of the tokens (n,L,64) fp32
AB = NI * 9 * 256 * 256 * 4                 # affinity
FB = NI * 64 * 256 * 256 * 4                # feats: hi + lo planes, fp16
ROW = 5 * TB + AB + FB                      # [tokens after pool][tokens at the first GEMM][q|k|v][affinity][feats]
dump = torch.zeros(M, ROW // 4, dtype=torch.float32, device="cuda")
_ffi.check(_ffi.lib().disco_set_debug_dump(ctx, C.c_void_p(dump.data_ptr()), ROW))
streams = [torch.cuda.Stream() for _ in range(M)]
f32 = dict(device="cuda", dtype=torch.float32)
fixed = [(torch.empty(64 // M, 313, 16, 16, **f32), torch.empty(64 // M, 313, 16, 16, **f32), torch.empty(64 // M, 2, 256, 256, **f32),
          torch.empty(64 // M, 9, 256, 256, **f32), torch.empty(64 // M, 2, 16, 16, **f32), torch.empty(64 // M, 1, 16, 16, **f32)) for _ in range(M)]


def step(serial):
    _ffi.check(_ffi.lib().disco_set_debug_checksums(ctx, C.c_void_p(table.data_ptr()), M, COLS))
    np.random.seed(130)
    idx, _ = global_draws(64, 256, 8, False)
    fs = peek_randint(256, m.max_fallback())
    main = torch.cuda.current_stream()
    prev = None
    keep = []
    for i, st in enumerate(streams):
        lo, hi = shard_bounds(64, M, i)
        st.wait_stream(main)
        if serial and prev is not None:
            st.wait_stream(prev)
        with torch.cuda.stream(st):
            keep.append(m.forward_once(gray[lo:hi], ab[lo:hi], True, 0, idx[lo:hi], None, fs, None, False, fixed[i] if args.fixed_out else None)[0])
        prev = st
    torch.cuda.synchronize()
    step.dump = dump.clone()
    return table.cpu().numpy().copy()


step(True)
ref = step(True)
assert (step(True) == ref).all(), "the serialised pass is not reproducible"
ref_dump = step.dump.clone()
ncols = int((ref != 0).any(0).sum())
print("stages with a checksum:", ncols)
first_bad = {}
for s in range(args.steps):
    got = step(False)
    for i in range(M):
        bad = np.nonzero(got[i] != ref[i])[0]
        if len(bad):
            first_bad[int(bad[0])] = first_bad.get(int(bad[0]), 0) + 1
            print(f"step {s} micro-batch {i}: first diverging stage {bad[0]}, {len(bad)} stages differ: {bad[:12].tolist()}", flush=True)
            if bad[0] == 47:
                T = NI * 256
                dgot, want = step.dump[i].view(torch.uint8), ref_dump[i].view(torch.uint8)
                def part(t, lo, nb): return t[lo:lo + nb]
                names = (("tokens after pool", 0, TB), ("tokens at the GEMM", TB, TB), ("qkv", 2 * TB, 3 * TB), ("affinity", 5 * TB, AB), ("feats hi+lo", 5 * TB + AB, FB))
                for nm, lo, nb in names:
                    a_, b_ = part(dgot, lo, nb), part(want, lo, nb)
                    nd = int((a_ != b_).sum())
                    print(f"   {nm:20s}: {nd} of {nb} bytes differ from the serialised pass")
                ta, tg = part(dgot, 0, TB).view(torch.float32).view(T, 64), part(dgot, TB, TB).view(torch.float32).view(T, 64)
                tw = part(want, 0, TB).view(torch.float32).view(T, 64)
                rows = torch.nonzero((tg != tw).any(1)).flatten().tolist()
                print(f"   token rows that differ at the GEMM: {rows[:16]}; after pool: {torch.nonzero((ta != tw).any(1)).flatten().tolist()[:16]}")
                for r_ in rows[:6]:
                    cols = torch.nonzero(ta[r_] != tw[r_]).flatten().tolist()
                    print(f"     row {r_}: columns {cols}: got {[round(float(ta[r_, c_]), 6) for c_ in cols[:4]]} want {[round(float(tw[r_, c_]), 6) for c_ in cols[:4]]}")
                af_g, af_w = part(dgot, 5 * TB, AB).view(torch.float32).view(NI, 9, 256, 256), part(want, 5 * TB, AB).view(torch.float32).view(NI, 9, 256, 256)
                d = torch.nonzero((af_g != af_w).any(1))
                if d.numel(): print(f"   affinity pixels that differ: {d.shape[0]}, first {d[:4].tolist()}, max |d| {(af_g - af_w).abs().max().item():.3e}")
print("first diverging stage -> count:", dict(sorted(first_bad.items())))
print("(stage numbering: 0 segnet conv0a (direct), 1-17 segnet convs, 18 repnet conv1_2.0 (direct), 19-45 repnet convs, 46 tokens, 47-64 wild-path encoder (qkv, attention, post-attention) x 6, 65 encoder output, 66 pal_logit, 67.. HourGlass2)")
_ffi.check(_ffi.lib().disco_set_debug_checksums(ctx, None, 0, 0))
