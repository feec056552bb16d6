#!/usr/bin/env python
"""Where does a tile's time go?  s_memtime stamps of workgroup (0, 0) of the f16+fp6x2 conv kernel (GPU box only).

    # build container:  hipcc ... -DMX_TIMELINE=1 -c csrc/conv_mx_ar3.hip -o tools/build/conv_mx_ar3_tl.o ; link -> tools/build/libdisco_tl.so
    DISCO_HIP_LIB=tools/build/libdisco_tl.so python tools/conv_timeline.py > gpurun_out/r04_conv_timeline.txt

The diagnostic build stamps, for the first and the last wave of that workgroup: tile start, per chunk (DMA landed, barrier passed,
taps done), epilogue math done, epilogue wait done, stores issued.  Printed per layer shape: cycles per phase averaged over the
tiles the persistent workgroup walks, next to the MFMA-pipe time the tile needs (2 waves per SIMD x their MFMAs x 32 cycles).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_helpers as H  # noqa: E402
from disentangledcolorization_amd import _ffi  # noqa: E402

SHAPES = [  # name, cin0, cin1, cout, h_in, stride, up0
    ("256->256 @64", 256, 0, 256, 64, 1, 0),
    ("128->128 @128", 128, 0, 128, 128, 1, 0),
    ("64->64 @256", 64, 0, 64, 256, 1, 0),
    ("cat 64+64->64 @256", 64, 64, 64, 256, 1, 1),
    ("outConv 64->2 @256 (tanh, fp32 NCHW out)", 64, 0, 2, 256, 1, 0),
]
EVENTS = 8192


def run_shape_x3(L, name, c0, c1, co, hin, stride, up0, n=64):
    """The f16x3 arithmetic (ColorProbNet / SpixelNet): plain hi + lo act tensors through disco_op_conv3x3."""
    hs = hin // 2 if up0 else hin
    src0 = torch.relu(torch.randn(2, n, hs, hs, c0, device="cuda")).half()
    w = torch.randn(co, c0, 3, 3) * 0.05
    packed = H.pack_conv(w)
    ho = (hin - 1) // stride + 1
    out = torch.empty(2, n, ho, ho, co, device="cuda", dtype=torch.float16)
    bias = torch.zeros(co, device="cuda")
    d = _ffi.ConvDesc(n, hin, hin, c0, 0, up0, 0, co, stride, _ffi.ACT_RELU, 0.0, 0)

    def run():
        _ffi.check(L.disco_op_conv3x3(C.byref(d), _ffi.ptr(src0), None, _ffi.ptr(packed), _ffi.ptr(bias), None, None, None, _ffi.ptr(out), H.stream()))
    return measure(L, L.disco_diag_conv_timeline_x3, run, name, c0, co, ho, n, (c0 // 16) * 27 * 2 * 32)


def run_shape(L, name, c0, c1, co, hin, stride, up0, n=64):
    hs = hin // 2 if up0 else hin
    planes = _ffi.PLANE_Q6
    x0 = H.to_act_mx(torch.relu(torch.randn(n, c0, hs, hs, device="cuda")), planes=planes, sexp=2)
    x1 = H.to_act_mx(torch.relu(torch.randn(n, c1, hin, hin, device="cuda")), planes=planes, sexp=2) if c1 else None
    w = torch.randn(co, c0 + c1, 3, 3) * 0.05
    packed, wexp = H.pack_conv_mx(w, 2)
    ho = (hin - 1) // stride + 1
    f32 = co < 32                                    # the network's last layer: 2 channels, tanh, fp32 NCHW
    out = torch.empty(n, co, ho, ho, device="cuda") if f32 else H.MxAct(n, co, ho, ho, planes, 0)
    bias = torch.zeros(co, device="cuda")
    d = _ffi.ConvMxDesc(n, hin, hin, c0, c1, up0, 0, x0.sexp, x1.sexp if x1 else 0, co, stride, _ffi.ACT_TANH if f32 else _ffi.ACT_RELU, 0.0,
                        0 if f32 else planes, 0, 1 if f32 else 0, 0, 0, 0, 1, 0)

    def run():
        _ffi.check(L.disco_op_conv3x3_mx(C.byref(d), _ffi.ptr(x0.buf), _ffi.ptr(x1.buf) if x1 else None, _ffi.ptr(packed), _ffi.ptr(wexp),
                                        _ffi.ptr(bias), None, None, None, _ffi.ptr(out if f32 else out.buf), None, None, H.stream()))
    return measure(L, L.disco_diag_conv_timeline, run, name, c0 + c1, co, ho, n, (c0 + c1) // 32 * 6912)


def measure(L, diag, run, name, cin, co, ho, n, pipe_cycles):
    for _ in range(20):                      # warm: clocks at their sustained level
        run()
    torch.cuda.synchronize()
    buf = np.zeros((2, EVENTS), np.uint64)
    assert diag(None, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    assert diag(buf.ctypes.data_as(C.c_void_p), 0) == 0
    fl = 2.0 * 9 * cin * co * ho * ho * n
    nch = cin // 16
    print("==== %s, n = %d: %.4f ms (instrumented), %.0f TF alg; %d chunks per tile; MFMA pipe time per tile and SIMD at 2 waves per SIMD on the 8-wave tiles: %d cycles"
          % (name, n, ms, fl / ms / 1e9, nch, pipe_cycles))
    if RAW:
        # every stamp of the first tile of the last wave: tag, cycles since the previous stamp (s_memtime runs at 100 MHz x ... see the header: the
        # unit is the constant-rate counter, NOT shader clocks; the tile total against the launch time above gives the conversion)
        cnt = int(buf[1, 0])
        ev = [(int(v) >> 4, int(v) & 15) for v in buf[1, 1:min(cnt, EVENTS)]]
        names = {1: "tile", 2: "chunk", 3: "dma-landed", 4: "barrier(H)", 5: "barrier(Q)", 6: "taps-done", 7: "epi-math", 8: "epi-wait", 9: "stores", 10: "tap8", 11: "barrier-in-tap8", 12: "entry", 13: "prologue", 14: "p14", 15: "p15"}
        line = []
        for i in range(1, len(ev)):
            line.append("%s+%d" % (names.get(ev[i][1], str(ev[i][1])), ev[i][0] - ev[i - 1][0]))
            if ev[i][1] == 9:
                break
        print("  raw (last wave, first tile): total %d ticks | %s" % (ev[min(len(line), len(ev) - 1)][0] - ev[0][0], " ".join(line)))
    for w in range(2):
        cnt = int(buf[w, 0])
        ev = [(int(v) >> 4, int(v) & 15) for v in buf[w, 1:min(cnt, EVENTS)]]
        tiles = []
        cur = None
        for t, tag in ev:
            if tag == 1:
                cur = {"start": t, "wait": 0, "barrier": 0, "tapsH": 0, "tapsQ": 0, "nH": 0, "nQ": 0, "last": t, "kind": None}
                tiles.append(cur)
                continue
            if cur is None:
                continue
            dt = t - cur["last"]
            if tag == 2:      # chunk start: time since the previous mark = taps of the previous chunk (or prologue)
                if cur["kind"] == 4: cur["tapsH"] += dt; cur["nH"] += 1
                elif cur["kind"] == 5: cur["tapsQ"] += dt; cur["nQ"] += 1
                else: cur["pro"] = dt
            elif tag == 10:   # (pipelined loop) taps 0-7 issued; the wait + barrier inside the last tap begins
                if cur["kind"] == 4: cur["tapsH"] += dt
                elif cur["kind"] == 5: cur["tapsQ"] += dt
            elif tag == 11: cur["barrier"] += dt
            elif tag == 3: cur["wait"] += dt
            elif tag in (4, 5): cur["barrier"] += dt; cur["kind"] = tag
            elif tag == 6:
                if cur["kind"] == 4: cur["tapsH"] += dt; cur["nH"] += 1
                elif cur["kind"] == 5: cur["tapsQ"] += dt; cur["nQ"] += 1
            elif tag == 7: cur["epi_math"] = dt
            elif tag == 8: cur["epi_wait"] = dt
            elif tag == 9: cur["epi_store"] = dt; cur["total"] = t - cur["start"]
            cur["last"] = t
        done = [t for t in tiles if "total" in t]
        if not done:
            print("  wave %s: no complete tile recorded (%d events)" % ("first" if w == 0 else "last", cnt)); continue
        body = done[1:] if len(done) > 2 else done        # the first tile carries the prologue
        avg = lambda k: sum(t.get(k, 0) for t in body) / len(body)
        tot = avg("total")
        print("  wave %-5s %2d tiles | tile %7.0f cyc | taps H %7.0f (%5.0f per chunk)  taps Q %7.0f (%5.0f per chunk) | dma wait %6.0f  barrier %6.0f | epilogue: math %6.0f  wait %6.0f  stores %6.0f"
              % ("first" if w == 0 else "last", len(done), tot, avg("tapsH"), avg("tapsH") / max(avg("nH"), 1), avg("tapsQ"), avg("tapsQ") / max(avg("nQ"), 1),
                 avg("wait"), avg("barrier"), avg("epi_math"), avg("epi_wait"), avg("epi_store")))
        print("             share of the tile: taps %4.1f %%  dma wait %4.1f %%  barrier %4.1f %%  epilogue %4.1f %%   (MFMA pipe need: %4.1f %%)"
              % (100 * (avg("tapsH") + avg("tapsQ")) / tot, 100 * avg("wait") / tot, 100 * avg("barrier") / tot,
                 100 * (avg("epi_math") + avg("epi_wait") + avg("epi_store")) / tot, 100 * (nch // 2 * 6912) / tot))


RAW = False
X3_SHAPES = [("f16x3 512->512 @32", 512, 0, 512, 32, 1, 0), ("f16x3 256->256 @64", 256, 0, 256, 64, 1, 0), ("f16x3 128->128 @128", 128, 0, 128, 128, 1, 0)]


def main():
    global RAW
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("only", nargs="?", default="")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--raw", action="store_true", help="also print every stamp of the first tile (small n: where does a launch's latency go?)")
    ap.add_argument("--x3", action="store_true", help="the f16x3 shapes (needs conv_mx_ar2.hip built with -DMX_TIMELINE=1 as well)")
    args = ap.parse_args()
    RAW = args.raw
    L = _ffi.lib()
    if not hasattr(L, "disco_diag_conv_timeline"):
        raise SystemExit("this library has no timeline probe: build with -DMX_TIMELINE=1 and point DISCO_HIP_LIB at it")
    for f in ("disco_diag_conv_timeline", "disco_diag_conv_timeline_x3"):
        if hasattr(L, f):
            getattr(L, f).restype = C.c_int
            getattr(L, f).argtypes = [C.c_void_p, C.c_int]
    if args.x3:
        for s in X3_SHAPES:
            if args.only in s[0]:
                run_shape_x3(L, *s, n=args.n)
        return
    for s in SHAPES:
        if args.only in s[0]:
            run_shape(L, *s, n=args.n)


if __name__ == "__main__":
    main()
