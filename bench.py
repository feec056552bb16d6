#!/usr/bin/env python
"""Headline benchmark: colorized 256x256 images/s of the DISCO hot path on N MI355X (BASELINE.json).

    python bench.py --gpus N --steps 10 --warmup 3 [--config 2|3|5a|5b]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Either launch works: started WITHOUT a launcher's environment (no WORLD_SIZE) and with --gpus N > 1, bench.py starts its own N ranks
(one process per GPU, LOCAL_RANK -> device, RCCL rendezvous on 127.0.0.1 and a free port - the launch convention of the reference's
main/utils_train.py:221-241 `init_dist`, which reads RANK / WORLD_SIZE the same way), passes rank 0's single JSON line through and
exits with the first failing rank's code (the other ranks are stopped by PID, so a dead rank cannot leave the job hanging in a collective).

One step = one forward of AnchorColorProb (test mode, K=8 clustering anchors, all six outputs produced) over a
batch of 64 synthetic 256x256 L-channel images per GPU (BASELINE config 2), inputs resident in HBM, followed —
for N>1 — by the ONE packed RCCL all-gather of pred_colors + hint_mask.  Weak scaling: 64 images per GPU.
Weights: the deterministic synthetic checkpoint of the real layout (disentangledcolorization_amd/synth.py).
Prints ONE JSON line on rank 0.

The timed loop runs without host synchronisation (k-means empty-cluster bookkeeping off); afterwards the same batch is
run once more with the bookkeeping on and the line reports `kmeans_events` - the run FAILS unless it is 0 and the two
results are identical, i.e. unless the timed forwards were the reference-exact ones.

DISCO_BENCH_FAKE=1 (tests/test_dist_gloo.py): the same distributed scaffolding on the gloo backend with a cheap CPU
stand-in for the forward, so that the N>1 code path is exercised without GPUs.
"""
import argparse
import hashlib
import json
import os
import sys
import time
import zlib

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_IMAGE = 255.470          # algorithmic work of one 256x256 image
FP16_MFMA_PEAK = 2.5e15            # dense, MI355X_MICROARCH.md
PMC_TRAFFIC_FILE = "r06_pmc_traffic.json"
ANCHOR_STUDY_FILE = "r06_anchor_mismatch.json"
SOCKET_POWER_CAP_W = 1400.0        # MI355X board power limit (rocm-smi; profiles/r04_power_per_kernel.txt sits on it)
MAX_SCLK_MHZ = 2400.0


def cpu_baseline(sd, seconds_budget=30.0, all_cores=False):
    """The CPU oracle (port of the reference arithmetic) timed on this box's host cores, N in {1, 8}, bounded to
    `seconds_budget`.  Thread count: min(32, cores) - oneDNN's 3x3 convs of this size stop scaling and start thrashing
    beyond a few dozen threads: with every one of the GPU box's 256 cores ONE 256x256 forward takes 90 s (0.011 img/s,
    profiles/r02_final_bench.json) where 32 threads give 4.7 img/s.  `--cpu-all-cores` adds that point (and its 1.5 minutes)."""
    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.gamut import gamut_points
    from oracle.disco_ref import DiscoOracle

    cores = os.cpu_count() or 1
    oracle = DiscoOracle(sd, gamut_points(), n_clusters=8)
    t_start = time.time()
    points = {}
    for threads in sorted({min(cores, 32)} | ({cores} if all_cores else set())):
        torch.set_num_threads(threads)
        for n in (1, 8):
            if time.time() - t_start > seconds_budget * 0.8:
                break
            gray, ab = synth.synth_inputs(n, 256, 256, seed=5)
            np.random.seed(130)
            t0 = time.time(); oracle.forward(gray, ab); best = time.time() - t0          # first call doubles as warm-up
            reps = 0
            while reps < 2 and (time.time() - t_start) + best < seconds_budget * (0.5 if n == 1 else 1.0):
                np.random.seed(130)
                t0 = time.time(); oracle.forward(gray, ab); best = min(best, time.time() - t0); reps += 1
            points["N=%d,threads=%d" % (n, threads)] = (round(n / best, 3), threads)
            if best > seconds_budget / 2:      # this thread count is hopeless on this box: do not burn the budget on N=8
                break
    key = max(points, key=lambda k: points[k][0])
    return {"value": points[key][0], "unit": "images/s", "cores": points[key][1], "host_cores": cores, "kind": "port",
            "points": {k: v[0] for k, v in points.items()},
            "sample": "oracle/disco_ref.py forward (torch CPU), 256x256, N in {1, 8}, %s threads of %d host cores, best of <=3 runs each "
                      "inside a %ds budget; value = the best point (%s)" % ("32 and all" if all_cores else str(min(cores, 32)), cores, int(seconds_budget), key)}


def source_hash():
    """Hash of everything libdisco_hip.so is built from (build.SOURCES + HEADERS: every translation unit incl. the per-arithmetic
    instantiation lists, the packers and the pooling kernels - round 3 hashed four files only)."""
    from disentangledcolorization_amd import build as B
    h = hashlib.sha256()
    for f in sorted(B.SOURCES) + sorted(B.HEADERS):
        with open(os.path.join(B.CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_record(key):
    """A number from the committed rocprofv3 PMC passes (tools/pmc_traffic.py): `hbm_bytes_per_launch` (FETCH_SIZE x2 gfx950 correction +
    WRITE_SIZE) or `mfma_busy` (SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) over the conv launches).  PMC counters cannot be read
    from inside a timed run, so the file carries the hash of the kernel sources it was measured on; a stale file is reported as null
    rather than as a number."""
    try:
        with open(os.path.join(REPO, "profiles", PMC_TRAFFIC_FILE)) as f:
            d = json.load(f)
        return d.get(key) if d.get("source_hash") == source_hash() else None
    except Exception:
        return None


def pmc_traffic():
    return pmc_record("hbm_bytes_per_launch")


def anchor_mismatch():
    """Anchor exactness as a rate (tools/anchor_study.py -> profiles/r06_anchor_mismatch.json, keyed by the kernel sources): per image size,
    how many images the HIP path decides differently from the fp32 CPU oracle, next to the same count for an exact-fp32 evaluation of
    the reference in ANOTHER summation order (torch-ROCm im2col + rocBLAS) and for an fp64 evaluation.  null when the file is stale."""
    try:
        with open(os.path.join(REPO, "profiles", ANCHOR_STUDY_FILE)) as f:
            d = json.load(f)
        if d.get("source_hash") != source_hash():
            return None
        out = {}
        for size, r in d["sets"].items():
            e = {"n": r["n"], "hip_vs_oracle": r["hip_vs_oracle"], "rate": r["rate_hip_vs_oracle"]}
            for v, name in (("G", "fp32_other_order_vs_oracle"), ("D", "fp64_vs_oracle"), ("B", "cpu_fp32_no_onednn_vs_oracle"), ("X", "oracle_on_another_host_vs_oracle")):
                if v in r:
                    e[name] = {"n": r[v]["n"], "count": r[v]["vs_oracle"], "rate": r[v]["rate_vs_oracle"]}
            for k in ("hip_mismatches_also_flipped_by_an_exact_evaluation", "hip_mismatches_flipped_by_hip_alone"):
                if k in r:
                    e[k] = r[k]
            out[size] = e
        return out
    except Exception:
        return None


class PowerProbe:
    """Socket power and shader clock DURING the timed loop: tools/power_sampler.py (a separate process reading librocm_smi64 at 100 Hz; it
    touches neither the HIP runtime nor the GPU's queues) is started before the warm-up and stopped after the timed region; the mean over
    the samples inside [t0, t1] goes into the roofline block - the reason the matrix pipe runs below its datasheet clock is on the line."""

    def __init__(self):
        import subprocess
        import tempfile
        self.path = os.path.join(tempfile.gettempdir(), "disco_bench_power_%d.csv" % os.getpid())
        try:
            self.proc = subprocess.Popen([sys.executable, os.path.join(REPO, "tools", "power_sampler.py"), "--out", self.path, "--hz", "100",
                                          "--dev", str(int(os.environ.get("LOCAL_RANK", "0")))], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            self.proc = None

    def read(self, t0, t1):
        import signal
        if self.proc is None:
            return None
        try:
            self.proc.send_signal(signal.SIGTERM)
            self.proc.wait(timeout=10)
            w, c, src = [], [], ""
            with open(self.path) as f:
                for line in f:
                    if line.startswith("#"):
                        src += line[1:].strip() + " "
                        continue
                    t, p, k = line.strip().split(",")
                    if t0 <= float(t) <= t1:
                        w.append(float(p)); c.append(float(k))
            os.unlink(self.path)
            if not w:
                return None
            return {"socket_w": round(sum(w) / len(w), 1), "sclk_mhz": round(sum(c) / len(c), 0), "samples": len(w), "source": src.strip()}
        except Exception:
            return None


def fake_forward(gray, ab, T, idx, pos, fstream, fbases, want, out=None):
    """DISCO_BENCH_FAKE: CPU stand-in with the model's output contract (depends on the per-image k-means rows); out: preallocated
    result tensors to fill (the pipelined path hands them in, as AnchorColorProb.forward_once does)."""
    n, _, H, W = gray.shape
    h, w = H // 16, W // 16
    d = torch.as_tensor(idx if idx is not None else pos, dtype=torch.float32)
    pred = torch.tanh(gray.repeat(1, 2, 1, 1) * 0.5 + d.sum(1).reshape(n, 1, 1, 1) * 1e-3)
    mask = torch.zeros(n, h * w)
    mask.scatter_add_(1, torch.as_tensor(idx if idx is not None else pos, dtype=torch.long), torch.ones(n, d.shape[1]))
    mask = mask.reshape(n, 1, h, w)
    if int(T) > 0:         # --diverse: three colorizations per image, image-major (model.py:148-159)
        pred = torch.stack([pred * (1.0 - 0.25 * t) for t in range(3)], 1).flatten(0, 1)
        mask = mask.repeat_interleave(3, 0)
    if out is not None:
        out[2].copy_(pred); out[5].copy_(mask)
        return out, (np.zeros(n, np.int32) if want else None)
    return (None, None, pred, None, None, mask), (np.zeros(n, np.int32) if want else None)


def measure_alt(precision, sd, gray, ab, n_global, args, sync):
    """Images/s of another precision mode on the same batch, same warm-up and step count (single GPU)."""
    from disentangledcolorization_amd.model import AnchorColorProb
    from disentangledcolorization_amd.runner import ShardedColorizer
    m = AnchorColorProb(inChannel=1, outChannel=313, sp_size=16, d_model=64, use_dense_pos=True, n_clusters=8, enhanced=True,
                        precision=precision, init_weights=False)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    r = ShardedColorizer.from_model(m, micro_batches=1 if args.pipeline else args.micro, exact_fallback=False)
    r.pipeline = bool(args.pipeline)

    def step():
        np.random.seed(130); torch.manual_seed(130)
        return r.colorize(gray, ab, n_global, 0, gather=True, async_gather=True)
    for _ in range(2 + args.warmup):
        step()
    r.wait(); sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    r.wait(); sync()
    dt = time.perf_counter() - t0
    return {"precision": precision, "value": round(n_global * args.steps / dt, 2), "unit": "images/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
            "note": "ColorProbNet on f16x2+fp8; passes the same parity suite; anchors differ from the fp32 reference in 0.66 % of 1 960 images (default: 0.10 %)"}


def single_image_latency(model, gray1, ab1, sync, reps=30):
    """Median wall time of one synchronised forward of ONE 256x256 image (issue + GPU + sync), after 5 warm-ups."""
    import numpy as np
    idx = np.stack([np.random.RandomState(7).choice((gray1.shape[2] // 16) * (gray1.shape[3] // 16), model.hint_num, replace=False)]).astype(np.int32)
    ts = []
    for it in range(5 + reps):
        sync()
        t0 = time.perf_counter()
        model.forward_once(gray1, ab1, True, 0, idx, None, None, None, False)
        sync()
        if it >= 5:
            ts.append(time.perf_counter() - t0)
    return round(sorted(ts)[len(ts) // 2] * 1e3, 3)


def other_single_gpu_configs(model, sd, precision, sync):
    """The other BASELINE configurations one GPU can run, each timed the plain way (3 warm-ups, then `reps` asynchronous calls closed by one
    synchronize), reported NEXT TO the headline and never as `value`:
      config 4  --no_resize: eight 512x512 and eight 768x512 L images through runner.colorize_mixed (grouped by shape: two forwards), K = 8
      config 5a --diverse, K = 16, clustering: the 32-image share one GPU has of the 8-GPU batch of 256 -> 96 colorizations per forward
      small     8 x 256x256, K = 8: the largest forward that still takes the small-batch path (SpixelNet on a side stream)"""
    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.model import AnchorColorProb
    from disentangledcolorization_amd.runner import colorize_mixed
    out = {}

    def timed(fn, reps):
        for _ in range(3):
            np.random.seed(130); fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            np.random.seed(130); fn()
        sync()
        return (time.perf_counter() - t0) / reps

    keep = model.sync_kmeans_events
    model.sync_kmeans_events = False
    try:
        grays = [synth.synth_inputs(1, 512, 512, seed=40 + i)[0].cuda() for i in range(8)] + [synth.synth_inputs(1, 768, 512, seed=60 + i)[0].cuda() for i in range(8)]
        dt = timed(lambda: colorize_mixed(model, grays), 5)
        px = 8 * 512 * 512 + 8 * 768 * 512
        out["config4_no_resize_mixed"] = {"workload": "8 x 512x512 + 8 x 768x512 (H x W), K=8, grouped by shape", "ms_per_batch": round(dt * 1e3, 2),
                                          "images_per_s": round(16 / dt, 1), "ns_per_pixel": round(dt * 1e9 / px, 2)}
        g8, a8 = synth.synth_inputs(8, 256, 256, seed=5)
        g8, a8 = g8.cuda(), a8.cuda()
        dt = timed(lambda: model(g8, a8, True, 0), 20)
        out["small_batch_8x256"] = {"workload": "8 x 256x256, K=8, forwards back to back on one stream", "ms_per_forward": round(dt * 1e3, 3), "images_per_s": round(8 / dt, 1)}
        # ... and issued the way the headline is: successive forwards alternating between two staggered streams (runner.ShardedColorizer.pipeline),
        # so that one forward's token path / k-means (a few CUs busy) runs under the other's convolutions
        from disentangledcolorization_amd.runner import ShardedColorizer
        r8 = ShardedColorizer.from_model(model, micro_batches=1, exact_fallback=False)
        r8.pipeline = True

        def step8():
            torch.manual_seed(130)
            r8.colorize(g8, a8, 8, 0, gather=False)
        for _ in range(6):
            np.random.seed(130); step8()
        r8.wait(); sync()
        t0 = time.perf_counter()
        for _ in range(40):
            np.random.seed(130); step8()
        r8.wait(); sync()
        dt = (time.perf_counter() - t0) / 40
        out["small_batch_8x256_pipelined"] = {"workload": "8 x 256x256, K=8, successive forwards pipelined over 2 HIP streams (the headline's issue mode)",
                                              "ms_per_forward": round(dt * 1e3, 3), "images_per_s": round(8 / dt, 1)}
    finally:
        model.sync_kmeans_events = keep
    m16 = AnchorColorProb(inChannel=1, outChannel=313, sp_size=16, d_model=64, use_dense_pos=True, n_clusters=16, enhanced=True,
                          precision=precision, init_weights=False)
    m16.load_state_dict(sd)
    m16 = m16.cuda().eval()
    m16.sync_kmeans_events = False
    g32, a32 = synth.synth_inputs(32, 256, 256, seed=5)
    g32, a32 = g32.cuda(), a32.cuda()
    dt = timed(lambda: m16(g32, a32, True, 1), 5)                 # sampled_T > 0: the three diverse colorizations of every image
    out["config5a_diverse_k16_shard"] = {"workload": "32 x 256x256 (one GPU's share of batch 256 on 8), --diverse, K=16 clustering: 96 colorizations",
                                         "ms_per_forward": round(dt * 1e3, 2), "colorizations_per_s": round(96 / dt, 1), "images_per_s": round(32 / dt, 1)}
    del m16
    return out


# The BASELINE.json configurations bench.py can time by name (--config).  "2" is the headline (the metric is quoted on it: weak scaling,
# 64 images per GPU); the others fix the GLOBAL batch BASELINE names and shard it over however many GPUs run (strong scaling in --gpus).
#   per_gpu / global_batch: images; k: anchors; T: sampled_T (> 0 = --diverse: three colorizations per image); random_hint
CONFIGS = {
    "2": dict(per_gpu=64, global_batch=0, k=8, T=0, random_hint=False, scaling="weak",
              workload="BASELINE config 2: batch=64/GPU synthetic 256x256 L-channel, K=8 clustering anchors, forward only, synthetic checkpoint of the DISCO layout"),
    "3": dict(per_gpu=0, global_batch=512, k=8, T=0, random_hint=False, scaling="strong",
              workload="BASELINE config 3: global batch=512 synthetic 256x256 L-channel sharded over the GPUs (64 per GPU on 8), K=8 clustering anchors, "
                       "synthetic checkpoint of the DISCO layout (the real one is not available offline), packed all-gather inside the timed region"),
    "4": dict(per_gpu=0, global_batch=16, k=8, T=0, random_hint=False, scaling="strong", mixed=True,
              workload="BASELINE config 4: --no_resize, a mixed list of eight 512x512 and eight 768x512 (H x W) L images on ONE GPU, K=8 clustering anchors, "
                       "grouped by shape (two forwards per step: 1 024 and 1 536 tokens per image); `value` counts IMAGES of these sizes, not 256x256 ones"),
    "5a": dict(per_gpu=0, global_batch=256, k=16, T=1, random_hint=False, scaling="strong",
               workload="BASELINE config 5 (a): global batch=256 synthetic 256x256 sharded over the GPUs, --diverse (three colorizations per image = 768 outputs), "
                        "K=16 clustering anchors"),
    "5b": dict(per_gpu=0, global_batch=256, k=16, T=0, random_hint=True, scaling="strong",
               workload="BASELINE config 5 (b): global batch=256 synthetic 256x256 sharded over the GPUs, random_hint with K=16 host-drawn anchor positions per image "
                        "(Python `random`, seed 130, global image order)"),
}


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, fake):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves - one process per GPU with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in its environment (what torch.distributed.run would set; the reference's init_dist, main/utils_train.py:221-241,
    reads the same variables), rendezvous on 127.0.0.1 (the container hostname may not resolve) and a free port.  Every rank inherits this
    process's stdout: only rank 0 writes to it (the one JSON line).  Returns the exit code: 0, or the first failing rank's - the remaining
    ranks are terminated by PID when one dies, so a failure never turns into a hang inside a collective."""
    import subprocess
    if not fake and os.environ.get("DISCO_DIST_BACKEND", "nccl") == "nccl":
        have = torch.cuda.device_count()
        if have < n:
            print("bench: --gpus %d asks for %d devices, this node shows %d (RCCL wants one device per rank)" % (n, n, have), file=sys.stderr)
            return 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT") or str(free_port()), WORLD_SIZE=str(n),
               LOCAL_WORLD_SIZE=str(n), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # N ranks share the host: without a cap every rank's NumPy / torch would start one thread per core for the host-side set-up (the synthetic
    # checkpoint's power iterations, the inputs), and N x cores threads thrash for minutes on a box whose container owns a fraction of the cores
    # it shows (seen in round 6: eight 28-thread workers needed 15 minutes for work that takes one of them 30 s).  torch.distributed.run does the
    # same for its children (OMP_NUM_THREADS=1); the GPU work does not depend on it.
    threads = str(max(1, min(8, (os.cpu_count() or 8) // (4 * n))))
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        env.setdefault(var, threads)
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r))) for r in range(n)]
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print("bench: rank %d exited with code %d; stopping the other ranks" % (procs.index(p), code), file=sys.stderr)
                    for q in live:
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def bench_mixed(args, cfg):
    """--config 4: one step = runner.colorize_mixed over BASELINE's mixed --no_resize list (inputs resident in HBM), W warm-up steps, K timed
    steps closed by one synchronize; afterwards one more pass with the k-means bookkeeping on must report no empty-cluster event and the same
    result.  One JSON line; `roofline` from hipEvent pairs around the conv launches of a profiled pass, like the headline's."""
    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.model import AnchorColorProb
    from disentangledcolorization_amd.runner import colorize_mixed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(0)
    sd = synth.synth_state_dict(130)
    model = AnchorColorProb(inChannel=1, outChannel=313, sp_size=16, d_model=64, use_dense_pos=True, n_clusters=cfg["k"], enhanced=True,
                            precision=args.precision, init_weights=False)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    model.set_profiling(0)
    grays = [synth.synth_inputs(1, 512, 512, seed=40 + i)[0].cuda() for i in range(8)] + [synth.synth_inputs(1, 768, 512, seed=60 + i)[0].cuda() for i in range(8)]
    px = sum(g.shape[2] * g.shape[3] for g in grays)

    def step():
        np.random.seed(130); torch.manual_seed(130)
        return colorize_mixed(model, grays)
    model.sync_kmeans_events = False
    for _ in range(2 + args.warmup):
        step()
    torch.cuda.synchronize()
    probe = PowerProbe()
    time.sleep(0.05)
    wall0 = time.time()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    power = probe.read(wall0, time.time())
    model.sync_kmeans_events = True
    exact = step()
    torch.cuda.synchronize()
    ev = model.last_kmeans_events()
    events = 0 if ev is None else int(np.asarray(ev).sum())
    same = all(torch.equal(a[2], b[2]) and torch.equal(a[5], b[5]) for a, b in zip(last, exact))
    if events != 0 or not same:
        raise SystemExit("bench: the timed (unsynchronised) forwards are not the reference-exact ones: kmeans_events=%d, identical: %s" % (events, same))
    checksum = 0
    for o in last:
        checksum = zlib.crc32(o[5].cpu().numpy().tobytes(), zlib.crc32(o[2].cpu().numpy().tobytes(), checksum))
    model.set_profiling(2)
    model.sync_kmeans_events = False
    conv_ms = conv_fl = 0.0
    conv_n = 0
    for shape_group in (grays[:8], grays[8:]):                 # one forward per shape: the profile of each
        np.random.seed(130)
        colorize_mixed(model, shape_group)
        torch.cuda.synchronize()
        nl, ms, fl = model.conv_profile()
        conv_n += nl; conv_ms += ms; conv_fl += fl
    model.set_profiling(0)
    achieved = conv_fl / (conv_ms * 1e-3) if conv_ms > 0 else 0.0
    out = {"metric": "colorized images/sec (--no_resize sizes)", "value": round(len(grays) * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": 1,
           "world_size_seen_by_backend": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16x3 / f16+fp6x2 conv stacks, fp32 token path (as the headline)",
           "data": "synthetic",
           "config": {"workload": cfg["workload"], "name": args.config, "global_batch": len(grays), "n_clusters": cfg["k"], "pixels_per_step": px,
                      "ns_per_pixel": round(elapsed / args.steps * 1e9 / px, 3), "equivalent_256x256_images_per_s": round(px / 65536.0 * args.steps / elapsed, 1)},
           "kmeans_events": events, "kmeans_fallback_images": model.kmeans_fallback_count(), "result_checksum": "%08x" % checksum,
           "roofline": {"bound": "mfma", "kernel": "conv3x3_mx_kernel", "achieved": round(achieved / 1e12, 2), "peak": FP16_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                        "frac": round(achieved / FP16_MFMA_PEAK, 4), "traffic": None, "conv_launches_per_step": conv_n,
                        "products_per_mac": 3, "executed_tflops": round(3 * achieved / 1e12, 1),
                        "socket_w": power and power["socket_w"], "sclk_mhz": power and power["sclk_mhz"],
                        "end_to_end_frac_of_fp16_conv_roofline": round(px / 65536.0 * args.steps / elapsed * GFLOP_PER_IMAGE * 1e9 / FP16_MFMA_PEAK, 4)}}
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2", choices=sorted(CONFIGS), help="the BASELINE.json configuration to time (default 2: the one the metric is quoted on); "
                    "3 / 5a / 5b fix the global batch (512 / 256 / 256) and shard it over --gpus")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU (weak scaling)")
    ap.add_argument("--global-batch", type=int, default=0, help="fix the TOTAL batch instead (tests; ragged shards allowed)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--precision", default=None, choices=["mx6", "mx8", "x2q", "mx8all", "f16x3"],
                    help="conv arithmetic per stack (disentangledcolorization_amd/model.py); default: the package default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-image latency measurement (the PMC passes of tools/collect_profiles.sh count conv launches by position)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra timed lines for BASELINE configs 4 and 5a (reported next to the headline, never as `value`)")
    ap.add_argument("--pipeline", type=int, default=1, help="1 (default): successive steps alternate between two HIP streams, each a full-size forward, the second "
                                                              "staggered behind the first (runner.py), everything joined by the synchronize() that closes the timed region; "
                                                              "0: every step is issued as --micro micro-batches and joined before the next one")
    ap.add_argument("--alt", action="store_true", help="also time the opt-in precision mode (x2q) and report it next to `value` as `opt_in_precision` (it runs second, on a warm "
                                                      "power-limited GPU: 5 % below its own stand-alone run, profiles/r03_bench_x2q.json - off by default since round 3)")
    ap.add_argument("--no-alt", action="store_true", help="(accepted for older command lines: the default now)")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU baseline with one thread per host core (minutes on a 256-core box)")
    ap.add_argument("--micro", type=int, default=2, help="micro-batches per GPU, each on its own HIP stream: the token path / k-means of one (a few CUs busy) "
                    "overlaps with the conv stacks of the other (+1.7%% over 1).  The per-launch profile behind `roofline` is taken on ONE stream after the timed loop")
    ap.add_argument("--profile-steps", type=int, default=3, help="single-stream, event-bracketed forwards after the timed loop (roofline / stage table)")
    args = ap.parse_args()
    if args.precision is None:
        from disentangledcolorization_amd.model import default_precision
        args.precision = default_precision()
    fake = os.environ.get("DISCO_BENCH_FAKE") == "1"
    cfg = CONFIGS[args.config]
    if cfg.get("mixed"):
        if args.gpus != 1 or fake:
            raise SystemExit("bench: --config 4 is BASELINE's single-GPU --no_resize configuration (--gpus 1, real device)")
        return bench_mixed(args, cfg)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, fake))

    # stdout carries exactly one JSON line: native libraries (RCCL's version banner, HIP runtime notices) write to fd 1
    # directly, so fd 1 points at stderr until the result is printed
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but the launcher's WORLD_SIZE is %d (run `python bench.py --gpus N` without a launcher, or "
                         "torch.distributed.run --nproc-per-node N with the same N)" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # DISCO_DIST_BACKEND=gloo (tests/test_gpu_dist.py): several ranks on ONE GPU - real forwards, streams and events on device tensors, the
    # collectives on gloo (RCCL wants a device per rank): the multi-rank code path on a single-GPU test box
    backend = os.environ.get("DISCO_DIST_BACKEND", "nccl")
    if not fake and backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    dev = torch.device("cpu") if fake else torch.device("cuda", local_rank)
    if not fake:
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if fake or backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev)
    sync = (lambda: None) if fake else torch.cuda.synchronize
    if fake and os.environ.get("DISCO_BENCH_FAIL_RANK") == str(rank):      # tests/test_dist_gloo.py: a rank that dies after the rendezvous
        os._exit(7)

    from disentangledcolorization_amd import synth
    from disentangledcolorization_amd.runner import ShardedColorizer, shard_bounds

    K, T = cfg["k"], cfg["T"]
    rep_out = 3 if T > 0 else 1
    n_global = args.global_batch if args.global_batch > 0 else (cfg["global_batch"] or args.batch * world)
    lo, hi = shard_bounds(n_global, world, rank)
    gray_all, ab_all = synth.synth_inputs(n_global, args.size, args.size, seed=5)
    gray, ab = gray_all[lo:hi].to(dev), ab_all[lo:hi].to(dev)    # inputs resident in HBM before timing
    model = sd = None
    if fake:
        runner = ShardedColorizer(fake_forward, n_clusters=K, random_hint=cfg["random_hint"], micro_batches=1 if args.pipeline else args.micro, exact_fallback=False)
        # the timed loop's real configuration - steps pipelined over two (host stand-in) streams, results written in place, the packed
        # all-gather enqueued asynchronously behind each forward - so that the gloo tests run the code an 8-GPU node will run
        runner.out_capable = True
        runner.progress_fn = lambda ev, k: None
        runner.pipeline = bool(args.pipeline)
    else:
        from disentangledcolorization_amd.model import AnchorColorProb
        sd = synth.synth_state_dict(130)
        model = AnchorColorProb(inChannel=1, outChannel=313, sp_size=16, d_model=64, use_dense_pos=True, n_clusters=K,
                                random_hint=cfg["random_hint"], enhanced=True, precision=args.precision, init_weights=False)
        model.load_state_dict(sd)
        model = model.cuda().eval()
        model.set_profiling(0)                    # the timed loop is the product: no event pairs around the launches
        # DISCO_FORCE_GATHER=1 (tests/test_gpu_dist.py): run the collectives at world size 1 too
        force = os.environ.get("DISCO_FORCE_GATHER") == "1"
        runner = ShardedColorizer.from_model(model, micro_batches=1 if args.pipeline else args.micro, exact_fallback=False, force_gather=force)   # no host sync while timing
        runner.pipeline = bool(args.pipeline)

    def seed():
        import random
        np.random.seed(130); torch.manual_seed(130); random.seed(130)

    def step():
        # batch k's all-gather is enqueued behind its forward and overlaps with batch k+1's convolutions; everything is
        # complete at the synchronize() that closes the timed region
        seed()
        return runner.colorize(gray, ab, n_global, T, gather=True, async_gather=True)

    # initialisation (untimed, not counted as warm-up): the first forward creates the native context (weight fold / pack /
    # upload / calibration) and sizes the workspace; a second one lets clocks and the caching allocator settle
    for _ in range(2):
        step()
    runner.wait()
    sync()
    probe = PowerProbe() if (model is not None and rank == 0) else None
    for _ in range(args.warmup):
        step()
    runner.wait()
    if use_dist:
        dist.barrier()
    sync()
    wall0 = time.time()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()   # asynchronous: no host sync and no instrumentation inside the timed region
    runner.wait()       # the outstanding all-gathers (N > 1)
    if use_dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    power = probe.read(wall0, time.time()) if probe is not None else None
    checksum = zlib.crc32(last[1].cpu().numpy().tobytes(), zlib.crc32(last[0].cpu().numpy().tobytes()))
    conv_ms = conv_fl = conv_bytes = 0.0
    conv_launches = 0
    stage_ms = {}
    if use_dist:
        t = torch.tensor([elapsed], device="cpu" if dist.get_backend() == "gloo" else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- after the timed region: was the timed path the reference-exact one?  One synchronised forward of the same batch:
    # the empty-cluster event counts must all be zero (then reading every image's fallback rows from the start of the draw
    # stream, as the timed loop does, changes nothing) and no fp8 activation may have been clamped.
    events = sat = 0
    if model is not None:
        exact = ShardedColorizer.from_model(model, exact_fallback=True, force_gather=force)
        seed()
        p_exact, m_exact = exact.colorize(gray, ab, n_global, T, gather=True)
        sync()
        events = 0 if exact.last_events is None else int(exact.last_events.sum())      # (random hints: no k-means, no draws)
        same = bool(torch.equal(p_exact, last[0]) and torch.equal(m_exact, last[1]))
        from disentangledcolorization_amd import _ffi
        import ctypes
        c64 = ctypes.c_uint64(0)
        _ffi.check(_ffi.lib().disco_saturation_count(model._ctx, _ffi.current_stream(), ctypes.byref(c64)))
        sat = int(c64.value)
        if events != 0 or not same:
            raise SystemExit("bench: the timed (unsynchronised) forward is not the reference-exact one: kmeans_events=%d, "
                             "identical to the synchronised forward: %s" % (events, same))

    # ---- the per-launch profile, separately from the timed loop: `--profile-steps` forwards of the same batch on ONE stream with a
    # hipEvent pair around every MFMA conv launch (recorded on the stream the kernels run on) and the stage marks
    prof_steps = max(1, args.profile_steps)
    if model is not None:
        model.set_profiling(2)
        single = ShardedColorizer.from_model(model, micro_batches=1, exact_fallback=False)
        for it in range(1 + prof_steps):           # the first one settles the clocks after the synchronised check above
            seed()
            p_prof, m_prof = single.colorize(gray, ab, n_global, T, gather=False)
            sync()
            if it == 0:
                continue
            nl, ms, fl = model.conv_profile()
            conv_bytes = model.conv_profile_bytes() / max(nl, 1)
            conv_launches += nl; conv_ms += ms; conv_fl += fl
            for name, sms, _ in model.profile():
                stage_ms[name] = stage_ms.get(name, 0.0) + sms
        model.set_profiling(0)
        lo_r, hi_r = shard_bounds(n_global, world, rank)
        if not (torch.equal(p_prof, last[0][lo_r * rep_out:hi_r * rep_out]) and torch.equal(m_prof, last[1][lo_r * rep_out:hi_r * rep_out])):
            raise SystemExit("bench: the profiled single-stream forward differs from the timed one")

    if rank == 0:
        ips = n_global * args.steps / elapsed
        out = {
            "metric": "colorized 256x256 images/sec", "value": round(ips, 2), "unit": "images/s",
            "n_gpus": world, "world_size_seen_by_backend": dist.get_world_size() if use_dist else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak" if args.global_batch == 0 and cfg["scaling"] == "weak" else "strong",
            "vs_baseline": None,
            "dtype": {"mx6": "f16x3 (fp16 hi/lo split, 3 MFMA products) for SpixelNet+ColorProbNet; f16+fp6x2 (fp16 main product + two fp6 e2m3 "
                             "correction products in one K=64 MFMA) for HourGlass2; fp32 accumulate",
                      "mx8": "f16x3 (fp16 hi/lo split, 3 MFMA products) for SpixelNet+ColorProbNet; f16+fp8x2 (fp16 main product + two fp8 e4m3 "
                             "correction products in one K=64 MFMA) for HourGlass2; fp32 accumulate",
                      "x2q": "f16x3 for SpixelNet; f16x2+fp8 (w_h a_h + w_l a_h in fp16, fp8(w) fp8(a_l) in one K=64 MFMA per 64 channels) for "
                             "ColorProbNet; f16+fp6x2 for HourGlass2; fp32 accumulate",
                      "mx8all": "f16+fp8x2 (fp16 main product + two fp8 e4m3 correction products), fp32 accumulate - not anchor-safe",
                      "f16x3": "f16x3 (fp16 hi/lo split operands, fp32 accumulate)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "name": args.config, "images_per_gpu": hi - lo,
                       "global_batch": n_global, "colorizations_per_step": n_global * rep_out, "n_clusters": K, "image_size": args.size,
                       "parallelism": "batch-sharded x%d, one packed all-gather of pred_colors+hint_mask" % world,
                       "issue": "steps pipelined over 2 HIP streams, staggered" if args.pipeline else "%d staggered micro-batches per step" % args.micro},
            "kmeans_events": events, "fp8_saturated_elements": sat, "result_checksum": "%08x" % checksum,
        }
        if model is not None:
            out["anchor_mismatch_rate"] = anchor_mismatch()
            out["kmeans_fallback_images"] = model.kmeans_fallback_count()
        if model is not None:
            # what the HourGlass2 really ran on: the channel-disparity guard of disco_finalize may have moved it from fp6 to fp8 corrections
            ar, disp = model.enhance_arithmetic()
            out["hourglass2_arithmetic"] = ar
            out["mx6_channel_disparity"] = round(disp, 1)
            out["mx6_channel_disparity_before_equalisation"] = round(model.equalised_from(), 1)
        if model is not None:
            achieved = conv_fl / (conv_ms * 1e-3) if conv_ms > 0 else 0.0
            out["roofline"] = {
                "bound": "mfma", "kernel": "conv3x3_mx_kernel (all instantiations: the f16x3, f16+fp6x2, f16+fp8x2 and f16x2+fp8 arithmetics share one skeleton)",
                "achieved": round(achieved / 1e12, 2), "peak": FP16_MFMA_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": round(achieved / FP16_MFMA_PEAK, 4), "traffic": pmc_traffic(),
                "launches_per_step": conv_launches // prof_steps,
                "measured": "hipEvent pairs around every conv launch of %d single-stream forwards of the same batch, after the timed loop "
                            "(the timed loop itself runs un-instrumented: %s)" % (prof_steps, "successive steps pipelined over 2 streams, full-size launches" if args.pipeline else "%d micro-batches per step on %d streams" % (args.micro, args.micro)),
                "avg_launch_ms": round(conv_ms / max(conv_launches, 1), 4),
                "algorithmic_gflop_per_launch": round(conv_fl / max(conv_launches, 1) / 1e9, 3),
                "algorithmic_hbm_bytes_per_launch": int(conv_bytes),
                # why the fraction is what it is, on the line itself: every algorithmic MAC costs THREE matrix-pipe products in both arithmetics
                # (f16x3: w_lo a_hi + w_hi a_lo + w_hi a_hi; f16+fp6x2: one fp16 product + two fp6 corrections in one K=64 MFMA), the pipe is busy
                # most of the time, and the socket sits on its power cap with the shader clock pulled below its 2.4 GHz maximum
                "products_per_mac": 3, "executed_tflops": round(3 * achieved / 1e12, 1),
                "executed_frac_of_peak": round(3 * achieved / FP16_MFMA_PEAK, 4),
                "mfma_busy": pmc_record("mfma_busy"),
                "socket_w": power and power["socket_w"], "socket_cap_w": SOCKET_POWER_CAP_W,
                "sclk_mhz": power and power["sclk_mhz"], "sclk_max_mhz": MAX_SCLK_MHZ,
                "power_samples_in_timed_loop": power and power["samples"],
                "end_to_end_frac_of_fp16_conv_roofline": round(ips * GFLOP_PER_IMAGE * 1e9 / world / FP16_MFMA_PEAK, 4),
            }
            out["stage_ms_per_step"] = {k: round(v / prof_steps, 3) for k, v in stage_ms.items()}        # of the profiled single-stream forwards
            if world == 1 and not args.no_latency and args.config == "2":
                # the reference's own call pattern is one image per forward (main/colorizer/inference.py:93-109): its latency, NOT `value`
                out["single_image_latency_ms"] = single_image_latency(model, gray[:1].contiguous(), ab[:1].contiguous(), sync)
            if world == 1 and not args.no_other_configs and args.config == "2" and args.batch == 64 and args.size == 256 and args.global_batch == 0:
                out["other_configs"] = other_single_gpu_configs(model, sd, args.precision, sync)
            if world == 1 and args.alt and not args.no_alt and args.precision == "mx6" and args.config == "2":
                # the opt-in arithmetic on the same inputs, timed the same way (NOT `value`: DESIGN.md section 2 says why it is opt-in)
                out["opt_in_precision"] = measure_alt("x2q", sd, gray, ab, n_global, args, sync)
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(sd, all_cores=args.cpu_all_cores)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
