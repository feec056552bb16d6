#!/usr/bin/env python
"""Socket power and shader clock per kernel (GPU box only): is the f16+fp6x2 conv kernel stall-bound or power-bound?

    hipcc --offload-arch=gfx950 -O3 tools/mfma_mix.hip -o tools/build/mfma_mix      (in the build container; the binary travels)
    python tools/power_per_kernel.py [--seconds 4] > gpurun_out/r04_power_per_kernel.txt

Runs sustained loops (back-to-back launches for --seconds each, 1.5 s idle in between) of
  * the conv kernel on the HourGlass2's layer shapes in the f16+fp6x2 arithmetic and on ColorProbNet's in f16x3 (random ReLU data),
  * the registers-only MFMA mixes of tools/mfma_mix.hip --sustain (the same instruction mix without LDS, memory or epilogue),
while tools/power_sampler.py records socket power and sclk at 50 Hz, and prints per segment: achieved rate, mean / max power,
mean reported sclk.  Reading: a conv loop that draws clearly LESS power than its registers-only mix while running slower is
stall-bound (the pipe idles); one that sits at the same power is at the cap and only fewer joules per result help.
"""
import argparse
import ctypes as C
import os
import signal
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as H  # noqa: E402
from disentangledcolorization_amd import _ffi  # noqa: E402

MX6 = [  # name, cin0, cin1, cout, h_in, stride, up0
    ("mx6 256->256 @64", 256, 0, 256, 64, 1, 0),
    ("mx6 128->128 @128", 128, 0, 128, 128, 1, 0),
    ("mx6 64->64 @256", 64, 0, 64, 256, 1, 0),
    ("mx6 cat 64+64->64 @256", 64, 64, 64, 256, 1, 1),
]
X3 = [
    ("f16x3 512->512 @32", 512, 512, 32),
    ("f16x3 256->256 @64", 256, 256, 64),
]


def sustained(run, seconds, flop_per_launch):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    time.sleep(1.5)
    t0 = time.time()
    launches = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        launches += 20
    t1 = time.time()
    return t0, t1, flop_per_launch * launches / (t1 - t0) * 1e-12


def conv_mx6(L, n, c0, c1, co, hin, stride, up0):
    hs = hin // 2 if up0 else hin
    planes = _ffi.PLANE_Q6
    x0 = H.to_act_mx(torch.relu(torch.randn(n, c0, hs, hs, device="cuda")), planes=planes, sexp=2)
    x1 = H.to_act_mx(torch.relu(torch.randn(n, c1, hin, hin, device="cuda")), planes=planes, sexp=2) if c1 else None
    w = torch.randn(co, c0 + c1, 3, 3) * 0.05
    packed, wexp = H.pack_conv_mx(w, 2)
    ho = (hin - 1) // stride + 1
    out = H.MxAct(n, co, ho, ho, planes, 0)
    bias = torch.zeros(co, device="cuda")
    d = _ffi.ConvMxDesc(n, hin, hin, c0, c1, up0, 0, x0.sexp, x1.sexp if x1 else 0, co, stride, _ffi.ACT_RELU, 0.0, planes, 0, 0, 0, 0, 0, 1, 0)
    keep = (x0, x1, packed, wexp, out, bias, d)

    def run():
        _ffi.check(L.disco_op_conv3x3_mx(C.byref(d), _ffi.ptr(x0.buf), _ffi.ptr(x1.buf) if x1 else None, _ffi.ptr(packed), _ffi.ptr(wexp),
                                        _ffi.ptr(bias), None, None, None, _ffi.ptr(out.buf), None, None, H.stream()))
    return run, 2.0 * 9 * (c0 + c1) * co * ho * ho * n, keep


def conv_x3(L, n, c0, co, hin):
    src0 = torch.relu(torch.randn(2, n, hin, hin, c0, device="cuda")).half()
    src0[1] *= 4.8e-4
    w = torch.randn(co, c0, 3, 3) * 0.05
    packed = H.pack_conv(w)
    out = torch.empty(2, n, hin, hin, co, device="cuda", dtype=torch.float16)
    bias = torch.zeros(co, device="cuda")
    d = _ffi.ConvDesc(n, hin, hin, c0, 0, 0, 0, co, 1, _ffi.ACT_RELU, 0.0, 0)
    keep = (src0, packed, out, bias, d)

    def run():
        _ffi.check(L.disco_op_conv3x3(C.byref(d), _ffi.ptr(src0), None, _ffi.ptr(packed), _ffi.ptr(bias), None, None, None, _ffi.ptr(out), H.stream()))
    return run, 2.0 * 9 * c0 * co * hin * hin * n, keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--samples", default=os.path.join(ROOT, "gpurun_out", "power_samples.csv"))
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.samples), exist_ok=True)
    sampler = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "power_sampler.py"), "--out", args.samples, "--hz", "50"])
    segs = []          # name, t0, t1, rate text
    try:
        L = _ffi.lib()
        torch.zeros(1, device="cuda")
        time.sleep(2.0)
        t0 = time.time(); time.sleep(2.0); segs.append(("idle", t0, time.time(), ""))
        for name, c0, c1, co, hin, stride, up0 in MX6:
            run, fl, keep = conv_mx6(L, args.n, c0, c1, co, hin, stride, up0)
            a, b, tf = sustained(run, args.seconds, fl)
            segs.append((name, a, b, "%7.1f alg TF = %7.1f executed" % (tf, 3 * tf)))
            del keep
        for name, c0, co, hin in X3:
            run, fl, keep = conv_x3(L, args.n, c0, co, hin)
            a, b, tf = sustained(run, args.seconds, fl)
            segs.append((name, a, b, "%7.1f alg TF = %7.1f executed" % (tf, 3 * tf)))
            del keep
        torch.cuda.synchronize()
        mix = os.path.join(ROOT, "tools", "build", "mfma_mix")
        if os.path.exists(mix):
            r = subprocess.run([mix, "--sustain", str(args.seconds)], capture_output=True, text=True, timeout=600)
            for line in r.stdout.splitlines():
                if line.startswith("SEG "):
                    nm, a, b, units, tf, clk = line[4:].split("|")
                    segs.append((nm, float(a), float(b), "%7.1f alg TF = %7.1f executed (%s G units/s, s_memtime clk %s GHz)" % (float(tf) / 3, float(tf), units, clk)))
        else:
            print("(tools/build/mfma_mix missing: registers-only mixes skipped)")
    finally:
        time.sleep(0.5)
        sampler.send_signal(signal.SIGTERM)
        sampler.wait(timeout=10)
    rows = []
    with open(args.samples) as f:
        head = f.readline().strip()
        for line in f:
            if line.startswith("#"):
                head += " " + line.strip()
                continue
            t, p, c = line.strip().split(",")
            rows.append((float(t), float(p), float(c)))
    print(head)
    print("%d samples, %.1f Hz" % (len(rows), len(rows) / max(rows[-1][0] - rows[0][0], 1e-9) if rows else 0))
    print("%-52s %8s %8s %8s %9s   %s" % ("segment (sustained %.0f s each)" % args.seconds, "mean W", "max W", "min W", "sclk MHz", "rate"))
    for name, a, b, rate in segs:
        # drop the first 0.5 s of a segment (the reading is a moving average)
        sel = [(p, c) for (t, p, c) in rows if a + 0.5 <= t <= b]
        if not sel:
            print("%-52s (no samples)   %s" % (name, rate))
            continue
        ps = [p for p, _ in sel]
        cs = [c for _, c in sel if c == c]
        print("%-52s %8.0f %8.0f %8.0f %9.0f   %s" % (name, sum(ps) / len(ps), max(ps), min(ps), sum(cs) / len(cs) if cs else float("nan"), rate))


if __name__ == "__main__":
    main()
