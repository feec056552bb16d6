#!/usr/bin/env python
"""Headline benchmark: colorized 256x256 images/s of the DISCO hot path on N MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one forward of AnchorColorProb (test mode, K=8 clustering anchors, all six outputs produced) over a
batch of 64 synthetic 256x256 L-channel images per GPU (BASELINE config 2), inputs resident in HBM, followed —
for N>1 — by the RCCL all-gather of pred_colors and hint_mask.  Weak scaling: 64 images per GPU.
Weights: the deterministic synthetic checkpoint of the real layout (disentangledcolorization_amd/synth.py).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_IMAGE = 255.470          # algorithmic work of one 256x256 image
FP16_MFMA_PEAK = 2.5e15            # dense, MI355X_MICROARCH.md
FP32_MFMA_PEAK = 157.3e12


def cpu_baseline(sd, seconds_budget=25.0):
    """The CPU oracle (port of the reference arithmetic) timed on this box's host cores, bounded sample."""
    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.gamut import gamut_points
    from oracle.disco_ref import DiscoOracle

    cores = min(os.cpu_count() or 1, 32)      # oneDNN convs of this size stop scaling (and thrash) beyond ~32 threads
    torch.set_num_threads(cores)
    oracle = DiscoOracle(sd, gamut_points(), n_clusters=8)
    n = 1
    gray, ab = synth.synth_inputs(n, 256, 256, seed=5)
    np.random.seed(130)
    t0 = time.time(); oracle.forward(gray, ab); warm = time.time() - t0
    best, reps = float("inf"), 0
    t_start = time.time()
    while reps < 5 and (time.time() - t_start) + warm < seconds_budget:
        np.random.seed(130)
        t0 = time.time(); oracle.forward(gray, ab); best = min(best, time.time() - t0); reps += 1
    if reps == 0:
        best = warm
    return {"value": round(n / best, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle/disco_ref.py forward, N=%d 256x256, best of %d after 1 warm-up, torch %d threads"
                      % (n, max(reps, 1), cores)}


def pmc_traffic():
    """HBM bytes per conv launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE; profiles/r01_pmc_traffic.json).  PMC counters cannot be read from inside a timed run."""
    try:
        with open(os.path.join(REPO, "profiles", "r01_pmc_traffic.json")) as f:
            return json.load(f)["hbm_bytes_per_launch"]
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--precision", default="mx8", choices=["mx8", "mx8all", "f16x3", "f16x1"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--micro", type=int, default=1, help="micro-batches per GPU, each on its own HIP stream")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: native libraries (RCCL's version banner, HIP runtime notices) write to fd 1
    # directly, so fd 1 points at stderr until the result is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.model import AnchorColorProb
    from disentangledcolorization_amd.runner import ShardedColorizer, shard_bounds

    sd = synth.synth_state_dict(130)
    model = AnchorColorProb(inChannel=1, outChannel=313, sp_size=16, d_model=64, use_dense_pos=True, n_clusters=8,
                            enhanced=True, precision=args.precision, init_weights=False)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    model.sync_kmeans_events = False          # no host sync inside the timed region
    model.set_profiling(2)                    # hipEvent pairs around every conv3x3_mfma launch (and stage marks)
    n_global = args.batch * world
    lo, hi = shard_bounds(n_global, world, rank)
    gray_all, ab_all = synth.synth_inputs(n_global, 256, 256, seed=5)
    gray, ab = gray_all[lo:hi].cuda(), ab_all[lo:hi].cuda()    # inputs resident in HBM before timing
    runner = ShardedColorizer.from_model(model, micro_batches=args.micro)

    def step():
        # batch k's all-gather is enqueued behind its forward on the communication stream and overlaps with batch k+1's
        # convolutions; everything is complete at the synchronize() that closes the timed region
        np.random.seed(130)
        return runner.colorize(gray, ab, n_global, 0, gather=True, async_gather=True)

    # initialisation (untimed, not counted as warm-up): the first forward creates the native context (weight fold / pack /
    # upload) and sizes the workspace; a second one lets clocks and the caching allocator settle on a fresh box
    for _ in range(2):
        step()
    runner.wait()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    runner.wait()
    conv_ms = conv_fl = 0.0
    conv_launches = 0
    stage_ms = {}
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()          # asynchronous: no host sync inside the timed region; every conv launch is event-bracketed
    runner.wait()       # the outstanding all-gathers (N > 1)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # per-launch hipEvent timings of the LAST timed step (the context keeps the events of its latest forward)
    nl, ms, fl = model.conv_profile()
    conv_bytes = model.conv_profile_bytes() / max(nl, 1)
    conv_launches, conv_ms, conv_fl = nl * args.steps, ms * args.steps, fl * args.steps
    for name, sms, _ in model.profile():
        stage_ms[name] = sms * args.steps
    if use_dist:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # registers-only MFMA loop on the same operand mix (after the timed region): the rate the power-managed clock
    # sustains for this data, i.e. the practical ceiling under the 2.5 PFLOP/s datasheet peak
    sustained = None
    if rank == 0:
        import ctypes
        from disentangledcolorization_amd import _ffi
        tf = ctypes.c_double(0.0)
        _ffi.check(_ffi.lib().disco_diag_mfma_rate(2 if args.precision == "f16x3" else 1, 8000, ctypes.byref(tf)))
        sustained = tf.value
        _ffi.check(_ffi.lib().disco_diag_mfma_rate(3, 8000, ctypes.byref(tf)))     # pixel operands non-negative, half zeros
        sustained_relu = tf.value

    if rank == 0:
        ips = n_global * args.steps / elapsed
        achieved = conv_fl / (conv_ms * 1e-3) if conv_ms > 0 else 0.0
        executed = (3 if args.precision == "f16x3" else 1) * achieved / 1e12
        out = {
            "metric": "colorized 256x256 images/sec", "value": round(ips, 2), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16x3 (fp16 hi/lo split operands, fp32 accumulate)" if args.precision == "f16x3" else "f16",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 2: batch=64/GPU synthetic 256x256 L-channel, K=8 clustering anchors, "
                                   "forward only, synthetic checkpoint of the DISCO layout", "images_per_gpu": args.batch,
                       "global_batch": n_global, "parallelism": "batch-sharded x%d, all-gather of pred_colors+hint_mask" % world},
            "roofline": {
                "bound": "mfma", "kernel": "conv3x3_mfma_kernel (all instantiations)",
                "achieved": round(achieved / 1e12, 2), "peak": FP16_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": round(achieved / FP16_MFMA_PEAK, 4), "traffic": pmc_traffic(),
                "launches_per_step": conv_launches // max(args.steps, 1),
                "avg_launch_ms": round(conv_ms / max(conv_launches, 1), 4),
                "algorithmic_gflop_per_launch": round(conv_fl / max(conv_launches, 1) / 1e9, 3),
                "algorithmic_hbm_bytes_per_launch": int(conv_bytes),
                "frac_of_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK, 4),
                "executed_mfma_tflops": round(executed, 2),
                "sustained_mfma_tflops_same_operand_mix": round(sustained, 1),
                "executed_frac_of_sustained": round(executed / sustained, 4) if sustained else None,
                "sustained_mfma_tflops_relu_like_operands": round(sustained_relu, 1),
                "end_to_end_frac_of_fp16_conv_roofline": round(ips * GFLOP_PER_IMAGE * 1e9 / world / FP16_MFMA_PEAK, 4),
            },
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage_ms.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
