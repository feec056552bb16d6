#!/usr/bin/env python
"""Where does a small-batch forward spend its time: kernels or the gaps between them?
  1) rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/trace_gaps.py --run [--batch 1]
  2) python tools/trace_gaps.py --report DIR
--run issues forwards separated by idle pauses; --report splits the kernel trace at the pauses and prints, for the median
forward: kernel count, first-start -> last-end span, sum of kernel durations, sum of gaps, and the top kernels by time."""
import argparse
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--run", action="store_true")
ap.add_argument("--report")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--size", type=int, default=256)
args = ap.parse_args()

if args.run:
    import numpy as np
    import torch
    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.model import AnchorColorProb
    m = AnchorColorProb(n_clusters=8, enhanced=True).cuda().eval()
    m.sync_kmeans_events = False
    gray, ab = synth.synth_inputs(args.batch, args.size, args.size, seed=5)
    gray, ab = gray.cuda(), ab.cuda()
    for it in range(30):
        np.random.seed(130)
        t0 = time.perf_counter()
        m(gray, ab, True, 0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it >= 25:
            print("forward %d: host issue %.3f ms, until idle %.3f ms" % (it, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
        time.sleep(0.02)
    sys.exit(0)

files = glob.glob(os.path.join(args.report, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
groups, cur = [], []
for r in rows:
    if cur and r[0] - cur[-1][1] > 5_000_000:      # > 5 ms idle = the pause between forwards
        groups.append(cur); cur = []
    cur.append(r)
if cur:
    groups.append(cur)
groups = [g for g in groups if len(g) > 50][-8:]
stats = []
for g in groups:
    span = g[-1][1] - g[0][0]
    busy = sum(e - s for s, e, _ in g)
    stats.append((span, busy, len(g), g))
stats.sort(key=lambda t: t[0])
span, busy, n, g = stats[len(stats) // 2]
print("median forward: %d kernels, span %.3f ms, sum of kernel durations %.3f ms, gaps %.3f ms (%.1f us per boundary)" %
      (n, span / 1e6, busy / 1e6, (span - busy) / 1e6, (span - busy) / 1e3 / max(1, n - 1)))
acc = {}
for s, e, k in g:
    k = k.replace("disco::", "").replace("(anonymous namespace)::", "").replace("_GLOBAL__N_1", "")
    k = (k[:k.index(">(") + 1] if ">(" in k else k.split("(")[0])[-90:]
    a = acc.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %4d x %8.1f us  = %8.1f us   %s" % (c, t / c / 1e3, t / 1e3, k))
