#!/usr/bin/env python
"""Time of one encoder stack (6 layers: attention + tail) at the --no_resize token counts (GPU box):
    DISCO_ATTN_MFMA=0 python tools/attn_ab.py ; python tools/attn_ab.py          (attention_kernel / attention_mfma_kernel from 1 024 tokens on)
Under `rocprofv3 --kernel-trace --stats` the per-kernel table separates the attention launches from the tails."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import gpu_helpers as H  # noqa: E402
from disentangledcolorization_amd import _ffi, synth  # noqa: E402
from test_gpu_ops import _encoder_weights  # noqa: E402

sd = synth.synth_state_dict(130)
wts = _encoder_weights(sd, "wildpath").to(H.DEV)
L = _ffi.lib()
for n, l in [(1, 1024), (1, 1536), (8, 1536), (16, 1024), (1, 4096), (4, 4096), (1, 16384)]:
    x = torch.randn(n, l, 64, device=H.DEV)
    pos = torch.randn(l, 64, device=H.DEV)
    out = torch.empty_like(x)
    ws = torch.empty(n * l * 384 * 4 + 256 + (4 << 20), device=H.DEV, dtype=torch.uint8)

    def run():
        _ffi.check(L.disco_op_encoder_stack(_ffi.ptr(x), _ffi.ptr(pos), _ffi.ptr(wts), _ffi.ptr(out), n, l, _ffi.ptr(ws), ws.numel(), H.stream()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20 if n * l <= 16384 else 5
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%3d x %6d tokens   %8.3f ms per stack   %7.1f us per layer   checksum %.6f" % (n, l, ms, ms / 6 * 1e3, out.double().abs().mean().item()), flush=True)
