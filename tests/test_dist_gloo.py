"""The N>1 path on CPU: world_size-2 (and 3) gloo processes shard a batch, run a forward, exchange k-means event counts,
all-gather the packed results - everything must equal the single-process run bit for bit (anchors included).
The forward here is a cheap CPU stand-in with the model's contract (the HIP forward needs a GPU) whose results depend on
the per-image k-means rows AND on the empty-cluster fallback rows it is handed; what is under test is runner.py: sharding,
global draws, the rank-invariant fallback stream, the packed collective, ragged and empty shards, and bench.py's
distributed scaffolding (the same code path the 8-GPU run takes, on the gloo backend)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from disentangledcolorization_amd.runner import ShardedColorizer, colorize_mixed, global_draws, peek_randint, shard_bounds  # noqa: E402


def _n_events(gray_i):
    """Images whose mean is positive 'hit empty clusters': 1..3 fallback draws, decided by the data alone."""
    m = float(gray_i.mean())
    return 0 if m <= 0 else 1 + int(abs(m) * 1e4) % 3


def _fake_forward(gray, ab, T, idx, pos, fstream, fbases, want):
    """Deterministic stand-in with the model's output contract; depends on the per-image draws and on the fallback rows."""
    n, _, H, W = gray.shape
    h, w = H // 16, W // 16
    d = torch.as_tensor(idx if idx is not None else pos, dtype=torch.float32)
    extra = torch.zeros(n)
    events = np.zeros(n, np.int32)
    if fstream is not None:
        bases = np.zeros(n, np.int64) if fbases is None else np.asarray(fbases)
        for i in range(n):
            events[i] = _n_events(gray[i])
            extra[i] = float(np.sum(fstream[bases[i]: bases[i] + events[i]].astype(np.float64) * (1 + np.arange(events[i]))))
    rep = 3 if T > 0 else 1
    pred = torch.tanh(gray.repeat(1, 2, 1, 1) * 0.5 + d.sum(1).reshape(n, 1, 1, 1) * 1e-3 + extra.reshape(n, 1, 1, 1) * 1e-4)
    mask = torch.zeros(n, h * w)
    mask.scatter_add_(1, torch.as_tensor(idx if idx is not None else pos, dtype=torch.long), torch.ones(n, d.shape[1]))
    pred, mask = pred.repeat_interleave(rep, 0), mask.repeat_interleave(rep, 0)
    return (None, None, pred, None, None, mask.reshape(n * rep, 1, h, w)), (events if want else None)


def _inputs(n_global):
    g = torch.Generator().manual_seed(1)
    gray = torch.rand(n_global, 1, 64, 64, generator=g) * 2 - 1
    gray += torch.linspace(-0.02, 0.02, n_global).reshape(-1, 1, 1, 1)       # a mix of images with and without "events"
    return gray, torch.zeros(n_global, 2, 64, 64)


def _seed():
    import random
    np.random.seed(130); random.seed(130); torch.manual_seed(130)


def _expected(n_global, random_hint, T=0):
    """The reference semantics computed directly: one sequential pass over the global batch."""
    gray, ab = _inputs(n_global)
    _seed()
    idx, pos = global_draws(n_global, 16, 4, random_hint)
    stream = peek_randint(16, 20 * 4 * 4)
    ev = np.array([0 if random_hint else _n_events(gray[i]) for i in range(n_global)])
    bases = np.concatenate(([0], np.cumsum(ev)[:-1]))
    (_, _, pred, _, _, mask), _ = _fake_forward(gray, ab, T, idx, pos, None if random_hint else stream, bases, True)
    return pred, mask, int(ev.sum())


def _worker(rank, world, port, n_global, random_hint, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gray, ab = _inputs(n_global)
    lo, hi = shard_bounds(n_global, world, rank)
    sc = ShardedColorizer(_fake_forward, n_clusters=4, random_hint=random_hint, max_fallback=16)
    _seed()
    stream = peek_randint(16, 64)
    pred, mask = sc.colorize(gray[lo:hi], ab[lo:hi], n_global, T)
    consumed = 0 if random_hint else int(sc.last_events.sum())
    nxt = int(torch.randint(16, (1,)))                 # the torch generator advanced by exactly the global number of draws
    assert nxt == int(stream[consumed]), (nxt, consumed)
    # pipelined form used by bench.py: two batches in flight, collectives only enqueued, completed by wait()
    _seed()
    p1, m1 = sc.colorize(gray[lo:hi], ab[lo:hi], n_global, T, async_gather=True)
    _seed()
    p2, m2 = sc.colorize(gray[lo:hi], ab[lo:hi], n_global, T, async_gather=True)
    sc.wait()
    assert not sc._pending
    for a, b in ((p1, pred), (p2, pred), (m1, mask), (m2, mask)):
        assert torch.equal(a, b)
    counts = [shard_bounds(n_global, world, r)[1] - shard_bounds(n_global, world, r)[0] for r in range(world)]
    if min(counts) == max(counts):      # equal shards: gathered straight into the result (views of ONE receive buffer, no unpack pass)
        assert pred._base is not None and mask._base is not None and pred._base is mask._base
    q.put((rank, pred.contiguous().numpy(), mask.contiguous().numpy(), consumed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global,world,random_hint,T", [(6, 2, False, 0), (5, 2, False, 0), (4, 2, True, 0), (7, 3, False, 0),
                                                          (2, 3, False, 0), (3, 2, False, 2)])
def test_multi_rank_gloo_equals_reference_semantics(n_global, world, random_hint, T):
    """(5,2) and (7,3): ragged shards; (2,3): one rank holds no image at all; (3,2,T=2): diverse (3 outputs per image)."""
    want_pred, want_mask, want_events = _expected(n_global, random_hint, T)
    assert random_hint or want_events > 0, "the case must exercise the fallback stream"
    # a single process through the runner must agree with the direct computation as well
    gray, ab = _inputs(n_global)
    _seed()
    sp, sm = ShardedColorizer(_fake_forward, 4, random_hint, max_fallback=16).colorize(gray, ab, n_global, T)
    assert torch.equal(sp, want_pred) and torch.equal(sm, want_mask)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + n_global * 13 + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, random_hint, T, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    for rank, pred, mask, consumed in got:
        assert np.array_equal(pred, want_pred.numpy()), rank
        assert np.array_equal(mask, want_mask.numpy()), rank
        assert consumed == want_events


def _fake_forward_out(gray, ab, T, idx, pos, fstream, fbases, want, out=None):
    """_fake_forward with AnchorColorProb.forward_once's out= contract (the pipelined path hands preallocated result tensors in)."""
    o, ev = _fake_forward(gray, ab, T, idx, pos, fstream, fbases, want)
    if out is not None:
        out[2].copy_(o[2]); out[5].copy_(o[5])
        return out, ev
    return o, ev


def _pipelined_worker(rank, world, port, n_global, q):
    """bench.py's timed configuration: pipeline = True (successive colorize() calls alternate between two streams - host stand-ins on
    CPU tensors, the same code otherwise), results written into preallocated tensors, async_gather = True."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gray, ab = _inputs(n_global)
    lo, hi = shard_bounds(n_global, world, rank)
    sc = ShardedColorizer(_fake_forward_out, n_clusters=4, max_fallback=16, exact_fallback=False)
    sc.out_capable, sc.progress_fn, sc.pipeline, sc.stagger_convs = True, (lambda ev, k: None), True, 26
    entered = []
    orig = sc._forward_pipelined
    sc._forward_pipelined = lambda *a: (entered.append(1), orig(*a))[1]
    results = []
    for step in range(5):
        _seed()
        r = sc.colorize(gray[lo:hi], ab[lo:hi], n_global, 0, gather=True, async_gather=True)
        if step != 2:              # step 2's result is DROPPED before wait(): its buffers must outlive the collective all the same
            results.append(r)
        del r
    assert len(sc._pending) == 5
    sc.wait()
    assert not sc._pending and not sc._pipe_busy
    assert len(entered) == (5 if hi > lo else 0), "the pipelined path must be the one that ran"
    q.put((rank, [(p.contiguous().numpy(), m.contiguous().numpy()) for p, m in results]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global,world", [(6, 2), (5, 2), (2, 3)])
def test_pipelined_async_gather_on_gloo(n_global, world):
    """The configuration `bench.py --gpus N` times - pipeline = True with the packed all-gather only ENQUEUED behind each forward - had
    never run with more than one rank (round 3: it required CUDA tensors).  World 2 with equal and ragged shards, world 3 with an
    empty shard: five batches in flight, one result dropped before wait(); every kept result equals the single-process one."""
    gray, ab = _inputs(n_global)
    _seed()
    sc = ShardedColorizer(_fake_forward, 4, False, max_fallback=16, exact_fallback=False)
    want_pred, want_mask = sc.colorize(gray, ab, n_global, 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + (os.getpid() * 5 + n_global * 17 + world) % 2000
    procs = [ctx.Process(target=_pipelined_worker, args=(r, world, port, n_global, q)) for r in range(world)]
    for p in procs: p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs: p.join(timeout=60)
    for rank, results in got:
        assert len(results) == 4
        for pred, mask in results:
            assert np.array_equal(pred, want_pred.numpy()) and np.array_equal(mask, want_mask.numpy()), rank


def test_ranks_with_different_seeds_are_detected():
    """The exchange carries a checksum of the draws: a rank that seeded differently must fail loudly, not diverge."""
    code = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import test_dist_gloo as T
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
gray, ab = T._inputs(4)
lo, hi = T.shard_bounds(4, 2, rank)
np.random.seed(130 + rank); torch.manual_seed(130)
try:
    T.ShardedColorizer(T._fake_forward, 4, False, max_fallback=16).colorize(gray[lo:hi], ab[lo:hi], 4)
    print("NOT DETECTED")
except RuntimeError as e:
    print("DETECTED" if "seed" in str(e) else "OTHER " + str(e))
dist.destroy_process_group()
''' % (REPO, REPO)
    port = 31500 + os.getpid() % 1000
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, "-c", code], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all("DETECTED" in o and "NOT DETECTED" not in o for o in outs), outs


def test_bench_distributed_scaffolding_on_gloo():
    """bench.py's own N>1 code path (process group, shard, async packed all-gather, closing barrier, max-over-ranks time)
    with an injected CPU forward on the gloo backend: world 2 must report the same result checksum as world 1 on the same
    global batch."""
    lines = {}
    for world in (1, 2):
        port = 33500 + (os.getpid() + world) % 1000
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), DISCO_BENCH_FAKE="1")
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
               "--global-batch", "6", "--size", "64", "--no-cpu-baseline"]
        procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                 for r in range(world)]
        outs = [p.communicate(timeout=300) for p in procs]
        assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
        line = [l for l in outs[0][0].splitlines() if l.startswith("{")]
        assert len(line) == 1, outs[0]
        assert all(not [l for l in o[0].splitlines() if l.startswith("{")] for o in outs[1:]), "only rank 0 prints"
        lines[world] = json.loads(line[0])
    assert lines[1]["n_gpus"] == 1 and lines[2]["n_gpus"] == 2
    assert lines[2]["world_size_seen_by_backend"] == 2
    assert "pipelined" in lines[2]["config"]["issue"]         # the default timed loop: pipeline = 1 + async gather, in fake mode too
    assert lines[1]["result_checksum"] == lines[2]["result_checksum"]
    assert lines[2]["config"]["global_batch"] == 6 and lines[2]["value"] > 0


class _RangeModel:
    """The slice of AnchorColorProb that ShardedColorizer.from_model touches, with a calibration STATE: `limit` = the largest |gray| the
    'context' covers.  A forward clamps (counts) every image beyond it and - like the fp8 planes - returns something else for it;
    calibrate(images) widens the limit to their maximum.  range_checks = 3 like the product."""
    sp_size, hint_num, random_hint, sync_kmeans_events, hint2regress = 16, 4, False, False, False

    def __init__(self):
        self.range_checks, self.limit, self.sat, self.calibrated_on, self.own_checks = 3, 1.0, 0, [], 0

    def max_fallback(self):
        return 16

    def forward_once(self, g, a, test_mode, T, idx, pos, fs, fb, want, out=None, range_check=True):
        self.own_checks += int(range_check)
        (o, ev) = _fake_forward(g, a, T, idx, pos, fs, fb, want)
        over = g.abs().flatten(1).amax(1) > self.limit
        self.sat += int(over.sum())
        pred = torch.where(over.reshape(-1, 1, 1, 1), torch.full_like(o[2], -7.0), o[2])
        return (None, None, pred, None, None, o[5]), ev

    def saturation_count(self):
        v, self.sat = self.sat, 0
        return v

    def calibrate(self, images):
        self.calibrated_on.append(images.clone())
        self.limit = max(self.limit, float(images.abs().max()))


def _range_worker(rank, world, port, q):
    import warnings
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_global = 5
    gray, ab = _inputs(n_global)
    gray = gray.clamp(-1, 1)
    gray[4] *= 3.0                     # the LAST image (rank 1's shard) lies outside the load-time range
    lo, hi = shard_bounds(n_global, world, rank)
    m = _RangeModel()
    sc = ShardedColorizer.from_model(m, exact_fallback=False)
    assert m.range_checks == 3, "from_model must not modify the caller's model"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _seed()
        pred, mask = sc.colorize(gray[lo:hi], ab[lo:hi], n_global, 0)
    q.put((rank, pred.contiguous().numpy(), m.limit, [c.numpy() for c in m.calibrated_on], len(w), m.own_checks, sc._range_left))
    dist.barrier()
    dist.destroy_process_group()


def test_range_check_is_collective_under_a_process_group():
    """Advisor finding of round 5: from_model used to switch the model's range check OFF under a process group (and mutate the caller's
    model); an out-of-range batch then clamped silently.  Now the first batches are checked TOGETHER: rank 1's shard clamps, rank 0's
    does not - both ranks must re-calibrate on the SAME gathered images (identical contexts afterwards), run their shards again, and return
    the result an all-covering calibration gives; the model's own per-rank check stays out of it."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30700 + os.getpid() % 1000
    procs = [ctx.Process(target=_range_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    got = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    gray, ab = _inputs(5)
    gray = gray.clamp(-1, 1); gray[4] *= 3.0
    _seed()
    idx, _ = global_draws(5, 16, 4, False)
    want = _fake_forward(gray, ab, 0, idx, None, peek_randint(16, 16), None, False)[0][2].numpy()
    for rank, pred, limit, cal, nwarn, own, left in got:
        assert np.array_equal(pred, want), "rank %d returned a clamped result" % rank
        assert abs(limit - float(gray.abs().max())) < 1e-6 and nwarn == 1 and own == 0 and left == 2
        assert len(cal) == 1 and cal[0].shape[0] == 64 and np.array_equal(cal[0], got[0][3][0]), "every rank calibrates on the same images"


def _plain_bench(args, extra_env=None):
    """`python bench.py ...` with NO launcher environment (what the driver runs): returns (parsed JSON line, stderr)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(DISCO_BENCH_FAKE="1", **(extra_env or {}))
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(args), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    return r


@pytest.mark.parametrize("config", ["2", "3", "5a", "5b"])
def test_bench_starts_its_own_ranks(config):
    """The driver's command is plain `python bench.py --gpus N ...` (no torch.distributed.run): bench.py must start its N ranks itself,
    print exactly ONE JSON line, report the world size the BACKEND saw, and - for the named BASELINE configurations with a fixed global
    batch (3: 512 images K=8; 5a: 256 images --diverse K=16; 5b: 256 images random_hint K=16) - give the same result checksum on 2 ranks
    as on 1 (shards of the same global batch, draws in global image order)."""
    common = ["--steps", "2", "--warmup", "1", "--size", "64", "--batch", "3", "--no-cpu-baseline", "--config", config]
    two = _plain_bench(["--gpus", "2"] + common)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [l for l in two.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, two.stdout
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["world_size_seen_by_backend"] == 2 and d2["steps"] == 2 and d2["warmup"] == 1
    assert d2["config"]["name"] == config and ("config %s" % config[0]) in d2["config"]["workload"]
    want_global = {"2": 6, "3": 512, "5a": 256, "5b": 256}[config]
    assert d2["config"]["global_batch"] == want_global
    assert d2["config"]["colorizations_per_step"] == want_global * (3 if config == "5a" else 1)
    assert d2["config"]["n_clusters"] == (8 if config in "23" else 16)
    assert d2["scaling"] == ("weak" if config == "2" else "strong")
    if config != "2":
        one = _plain_bench(["--gpus", "1"] + common)
        assert one.returncode == 0, one.stderr[-3000:]
        d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
        assert d1["n_gpus"] == 1 and d1["result_checksum"] == d2["result_checksum"]


def test_bench_self_launch_reports_a_failing_rank():
    """A rank that dies must end the job with its exit code instead of leaving the others in a collective: world 2 with a launcher
    environment that disagrees (WORLD_SIZE given by the parent = bench.py does not launch; --gpus 2 vs WORLD_SIZE 1 is refused)."""
    env = dict(os.environ, DISCO_BENCH_FAKE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="31999")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
    # ... and a rank failing inside a self-launched job: an impossible shard request (DISCO_BENCH_FAIL_RANK makes that rank exit 7 after init)
    r = _plain_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "32", "--batch", "2", "--no-cpu-baseline"], {"DISCO_BENCH_FAIL_RANK": "1"})
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_shard_bounds_cover_batch():
    for n in (1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


def test_global_draws_follow_reference_stream():
    np.random.seed(130)
    idx, _ = global_draws(3, 256, 8, False)
    np.random.seed(130)
    want = np.stack([np.random.choice(256, 8, replace=False) for _ in range(3)])
    assert np.array_equal(idx, want)


class _FakeModel:
    """The slice of AnchorColorProb that colorize_mixed touches, on the CPU stand-in forward."""
    sp_size, hint_num, random_hint = 16, 4, False

    def forward_with_draws(self, gray, ab, test_mode, T, init_idx, hint_pos):
        assert init_idx.shape == (gray.shape[0], 4)
        return _fake_forward(gray, ab, T, init_idx, hint_pos, None, None, False)[0]


def test_colorize_mixed_groups_by_shape_and_keeps_the_per_file_draw_order():
    """BASELINE config 4: a list of images of different sizes -> per-image results in input order, equal shapes batched, the
    k-means rows of image i being the i-th draw from NumPy's global state exactly as in the reference's per-file loop."""
    g = torch.Generator().manual_seed(3)
    shapes = [(64, 64), (32, 64), (64, 64), (64, 32), (32, 64), (64, 64)]
    grays = [torch.rand(1, 1, h, w, generator=g) for h, w in shapes]
    _seed()
    want = []
    for gr in grays:            # the reference's loop: one file at a time
        l = (gr.shape[2] // 16) * (gr.shape[3] // 16)
        idx, _ = global_draws(1, l, 4, False)
        want.append(_fake_forward(gr, torch.zeros(1, 2, *gr.shape[2:]), 0, idx, None, None, None, False)[0])
    _seed()
    got = colorize_mixed(_FakeModel(), grays, max_batch=2)
    assert len(got) == len(grays)
    for w_, g_ in zip(want, got):
        assert torch.equal(w_[2], g_[2]) and torch.equal(w_[5], g_[5]) and g_[2].shape[0] == 1
