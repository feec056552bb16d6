"""Batch-sharded multi-GPU colorization: one process per GPU, ONE RCCL all-gather of the results.

The reference has no batched/multi-GPU inference (inference.py:93 loops over files with batch 1;
nn.DataParallel :76-82).  Every image is independent end to end, so the global batch
is cut into contiguous shards, each rank runs the HIP forward on its shard with the full weights,
and `pred_colors` and `hint_mask` travel together in one packed all-gather over xGMI
(torch.distributed backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

Results do not depend on the number of GPUs.  Everything random on the path is host-side and is
drawn for the GLOBAL batch in image order from the generators the reference consumes:
  * k-means initial rows (NumPy legacy RandomState, clusterkit.py:107) and random hints (Python
    `random`, basic.py:42-47): drawn once per batch, sliced per rank;
  * empty-cluster fallback rows (torch's CPU generator, clusterkit.py:181-182): image i reads the
    global draw stream at offset sum(events of all earlier images of the GLOBAL batch).  The ranks
    exchange their per-image event counts (one tiny all-gather) and re-run when an image that drew
    rows started at the wrong offset; afterwards every rank consumes the same number of draws.
Every rank must therefore seed NumPy / random / torch identically (checked: the exchange carries a
checksum of the draws).
"""
import os
import random
import zlib

import numpy as np
import torch
import torch.distributed as dist

# micro-batch i + 1 starts behind this many MFMA conv launches of micro-batch i (of 69 per forward: SpixelNet 18, ColorProbNet 27,
# HourGlass2 24); $DISCO_STAGGER_CONVS overrides (0 = off: the micro-batches start together), profiles/r03_stagger_sweep.txt
STAGGER_CONVS = int(os.environ.get("DISCO_STAGGER_CONVS", "26"))
# images a collective re-calibration gathers over all ranks (AnchorColorProb.calibrate takes at most 64)
CAL_IMAGES = 64


class _HostStream:
    """Stand-ins for torch.cuda.Stream / Event when the tensors live on the host (the gloo tests and bench.py's fake mode): the
    pipelined issue order, the result bookkeeping and the asynchronous collectives of _forward_pipelined / colorize are the SAME code
    on either device - only the stream operations are no-ops (host work is complete when the call returns)."""

    def wait_event(self, ev): pass
    def wait_stream(self, st): pass
    def synchronize(self): pass
    def __enter__(self): return self
    def __exit__(self, *a): return False


class _HostEvent:
    def record(self, st=None): pass
    def synchronize(self): pass


class _Sx:
    """Stream operations for tensors on `dev` (HIP streams, or the host stand-ins above)."""

    def __init__(self, dev):
        self.dev = torch.device(dev)
        self.cuda = self.dev.type == "cuda"

    def stream(self): return torch.cuda.Stream(device=self.dev) if self.cuda else _HostStream()
    def event(self): return torch.cuda.Event() if self.cuda else _HostEvent()
    def current(self): return torch.cuda.current_stream(self.dev) if self.cuda else _HostStream()
    def on(self, st): return torch.cuda.stream(st) if self.cuda else st

    def keep_for(self, t, st):
        """`t` was allocated on another stream than `st`, which reads or writes it: no reuse of its block before st is done."""
        if self.cuda and t is not None:
            t.record_stream(st)


def shard_bounds(n_global, world, rank):
    """Contiguous [lo, hi) of `rank`; the first n_global % world ranks hold one extra image."""
    base, extra = divmod(n_global, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def global_draws(n_global, n_tokens, k, random_hint):
    """(init_idx, hint_pos) for the whole batch, image order — clusterkit.py:107 / basic.py:42-47."""
    if random_hint:
        pos = np.stack([np.asarray(random.sample(range(0, n_tokens), random.randint(k, k))) for _ in range(n_global)])
        return None, pos.astype(np.int32)
    idx = np.stack([np.random.choice(n_tokens, k, replace=False) for _ in range(n_global)])
    return idx.astype(np.int32), None


def peek_randint(l, count):
    """The next `count` values torch.randint(l,(1,)) would return, without consuming them."""
    g = torch.Generator()
    g.set_state(torch.get_rng_state())
    return np.asarray(torch.randint(l, (count,), generator=g).tolist(), dtype=np.int32)


class ShardedColorizer:
    """forward_fn(gray, ab, sampled_T, init_idx, hint_pos, fallback_stream, fallback_bases, want_events)
         -> (6-tuple like the model, events (n,) int32 or None)

    For the product path forward_fn = AnchorColorProb.forward_once (HIP); tests inject a CPU function.
    exact_fallback: exchange event counts so that empty-cluster draws follow the reference's global stream (one host
    synchronisation per forward); False = no synchronisation, every image reads the stream from its start - identical
    as long as no empty-cluster event occurs (bench.py checks that after its timed loop)."""

    def __init__(self, forward_fn, n_clusters=8, random_hint=False, sp_size=16, group=None, micro_batches=1,
                 exact_fallback=True, max_fallback=None, force_gather=False, virtual_rank=None):
        self.forward_fn = forward_fn
        # virtual_rank = (world, rank): compute the share rank `rank` of `world` has of a global batch WITHOUT a process group - the global
        # draws, the slice at its global offset, the local forward - on the one GPU a test box has (tests/test_gpu_forward.py checks rank 7
        # of 8 of BASELINE configs 3 and 5 against the oracle that way).  No collective exists in this mode: gather=False and
        # exact_fallback=False only (the caller checks the event counts with forward_once).
        self.virtual_rank = None if virtual_rank is None else (int(virtual_rank[0]), int(virtual_rank[1]))
        if self.virtual_rank is not None and not (0 <= self.virtual_rank[1] < self.virtual_rank[0]):
            raise ValueError("virtual_rank = (world, rank) with 0 <= rank < world")
        # the collective range check of the first forwards (from_model): the model whose clamp counter is read, and how many checks are left
        self._range_model = None
        self._range_left = 0
        self.k, self.random_hint, self.sp = n_clusters, random_hint, sp_size
        self.group = group
        # micro_batches > 1: the local shard is cut into that many slices, each issued on its own HIP stream, so the
        # latency-bound token path / k-means of one slice (a few CUs busy) overlaps with the conv stacks of another
        self.micro = max(1, int(micro_batches))
        self.exact_fallback = bool(exact_fallback)
        self.max_fallback = int(max_fallback) if max_fallback else 20 * n_clusters
        # force_gather: run the collectives even at world size 1 when a process group exists (the only way to exercise the RCCL
        # path - packed send buffer, async work handle, result views - on a single-GPU box: tests/test_gpu_dist.py via bench.py)
        self.force_gather = bool(force_gather)
        self._streams = None
        # stagger: micro-batch i + 1 starts behind the `stagger_convs`-th conv launch of micro-batch i (progress_fn(event, k) arms
        # the forward that is issued next: AnchorColorProb.set_progress_event), so that the token path / k-means of either runs under
        # the other's convolutions; out_capable: forward_fn takes out= (preallocated result slices: no concatenation pass)
        self.progress_fn = None
        self.stagger_convs = 0
        self.out_capable = False
        self._out_channels = (313, 313)             # channels of pal_logit / ref_logit (2 with hint2regress)
        self._stagger_events = []
        # pipeline: successive colorize() calls alternate between two streams and are NOT joined with the caller's stream until wait():
        # call k + 1 starts behind the `stagger_convs`-th conv launch of call k, so that whole batches overlap the way staggered
        # micro-batches do - with full-size launches (a 32-image launch is ~2 % less efficient than a 64-image one) - at the price of an
        # asynchronous contract: results (and the inputs!) must stay untouched until wait().  bench.py's timed loop runs this way.
        self.pipeline = False
        self._pipe_streams = None
        self._pipe_events = None
        self._pipe_count = 0
        self._pipe_prev = None
        self._pipe_done = [None, None]
        self._pipe_busy = []
        self._last_stream = None
        self._pending = []          # outstanding asynchronous all-gathers (async_gather=True): (work, finish callback)
        self.last_events = None     # per-image empty-cluster draws of the GLOBAL batch of the latest exact forward

    @classmethod
    def from_model(cls, model, group=None, micro_batches=1, exact_fallback=None, force_gather=False, virtual_rank=None):
        exact = model.sync_kmeans_events if exact_fallback is None else exact_fallback
        r = cls(None, model.hint_num, model.random_hint, model.sp_size, group, micro_batches, exact, model.max_fallback(), force_gather, virtual_rank)
        # Under a process group the model's OWN range check stays out of the way (range_check=False per call - the caller's model object is
        # not modified): it would re-calibrate a rank on ITS shard (activation exponents are a per-context property, and an exponent moves
        # pred_colors at the 1e-5 level), so the result of an image could depend on how many ranks share the batch.  The check is made
        # COLLECTIVELY instead (_collective_range_check): the clamp counts of the first `model.range_checks` batches are summed over the
        # ranks, and when any rank clamped, every rank calibrates on the SAME images (an all-gather of a few from every shard) and runs its
        # shard again - all contexts stay identical.
        r.forward_fn = lambda g, a, T, idx, pos, fs, fb, want, out=None: model.forward_once(g, a, True, T, idx, pos, fs, fb, want, out,
                                                                                            range_check=not r._collective())
        r._range_model = model if hasattr(model, "saturation_count") and hasattr(model, "calibrate") else None
        r._range_left = int(getattr(model, "range_checks", 0))
        r.out_capable = True
        r._out_channels = (313, 2 if getattr(model, "hint2regress", False) else 313)
        if hasattr(model, "set_progress_event"):
            r.progress_fn = model.set_progress_event
            r.stagger_convs = STAGGER_CONVS
        return r

    def _collective(self):
        """More than one rank shares the batch: range checks and calibration are collective."""
        return self.virtual_rank is None and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _collective_range_check(self, gray_local):
        """One of the first batches of a model under a process group: sum the ranks' fp8 clamp counts; if any rank clamped, every rank
        calibrates on the same images - CAL_IMAGES // world from every shard (a rank with fewer repeats its first, an empty shard sends
        zeros: ranges only widen, so padding is harmless), all-gathered - and the caller runs its shard again.  Returns True then.
        Costs one host synchronisation and one 8-byte all-reduce per checked batch (the model's own check costs the same sync at world 1)."""
        model = self._range_model
        if model is None or self._range_left <= 0 or not self._collective():
            return False
        self._range_left -= 1
        world = dist.get_world_size(self.group)
        sx = _Sx(gray_local.device)
        with sx.on(self._last_stream if self._last_stream is not None else sx.current()):
            mine = int(model.saturation_count()) if gray_local.shape[0] else 0         # (synchronises the forward's stream)
        host = dist.get_backend(self.group) == "gloo"
        t = torch.tensor([mine], dtype=torch.int64, device="cpu" if host else gray_local.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        total = int(t.item())
        if total == 0:
            return False
        per = max(1, CAL_IMAGES // world)
        H, W = gray_local.shape[2:]
        send = gray_local.new_zeros((per, 1, H, W))
        m = min(per, gray_local.shape[0])
        if m:
            send[:m] = gray_local[:m]
            send[m:] = gray_local[:1]
        recv = gray_local.new_empty((world * per, 1, H, W))
        all_gather_into(recv, send, group=self.group)
        import warnings
        warnings.warn("%d fp8 activation values were clamped across the %d ranks: this batch is outside the ranges the contexts were calibrated on; "
                      "every rank re-calibrates on the same %d images and runs its shard again" % (total, world, world * per))
        model.calibrate(recv)
        return True

    # ---- local forward (optionally as micro-batches on separate streams) ----------------------------------------
    def _forward_local(self, gray, ab, sampled_T, idx, pos, fstream, fbases, want):
        n = gray.shape[0]
        self._last_stream = None
        if (self.pipeline and not want and self.out_capable and self.progress_fn is not None and int(sampled_T) == 0
                and self.micro <= 1 and n > 0):
            return self._forward_pipelined(gray, ab, idx, pos, fstream, fbases)
        # a forward that reports its empty-cluster events synchronises the host before it returns, so micro-batches would
        # run one after the other anyway: the exact mode issues the shard as one batch
        m = 1 if want else min(self.micro, n)
        if m <= 1 or not gray.is_cuda:
            return self.forward_fn(gray, ab, sampled_T, idx, pos, fstream, fbases, want)
        if self._streams is None or len(self._streams) < m:
            self._streams = [torch.cuda.Stream(device=gray.device) for _ in range(m)]
        main = torch.cuda.current_stream(gray.device)
        stagger = self.progress_fn is not None and self.stagger_convs > 0
        while stagger and len(self._stagger_events) < m - 1:
            ev = torch.cuda.Event()
            ev.record(main)                            # (torch creates the hipEvent on the first record)
            self._stagger_events.append(ev)
        full = None
        if self.out_capable and int(sampled_T) == 0:
            # the whole batch's results, allocated once on the caller's stream; every micro-batch writes its slice
            H, W = gray.shape[2:]
            h, w = H // self.sp, W // self.sp
            probe = self._out_channels
            f32 = dict(device=gray.device, dtype=torch.float32)
            full = (torch.empty(n, probe[0], h, w, **f32), torch.empty(n, probe[1], h, w, **f32), torch.empty(n, 2, H, W, **f32),
                    torch.empty(n, 9, H, W, **f32), torch.empty(n, 2, h, w, **f32), torch.empty(n, 1, h, w, **f32))
        parts, evs = [], []
        for i in range(m):
            lo, hi = shard_bounds(n, m, i)
            st = self._streams[i]
            st.wait_stream(main)                       # inputs were produced on the caller's stream
            if stagger and i > 0:
                st.wait_event(self._stagger_events[i - 1])
            with torch.cuda.stream(st):
                if stagger and i + 1 < m:
                    self.progress_fn(self._stagger_events[i], self.stagger_convs)
                args = (gray[lo:hi], ab[lo:hi], sampled_T, None if idx is None else idx[lo:hi],
                        None if pos is None else pos[lo:hi], fstream, None if fbases is None else fbases[lo:hi], want)
                try:
                    o, e = self.forward_fn(*args, tuple(t[lo:hi] for t in full)) if full is not None else self.forward_fn(*args)
                except BaseException:
                    if stagger and i + 1 < m:
                        self.progress_fn(None, 0)          # a forward that never reached the native call must not leave the event armed
                    raise
                parts.append(o); evs.append(e)
        for i in range(m):
            main.wait_stream(self._streams[i])
            if full is None:
                for t in parts[i]:
                    t.record_stream(main)              # allocated on the side stream, consumed on the caller's
        out = full if full is not None else tuple(torch.cat([p[k] for p in parts], 0) for k in range(6))
        return out, (np.concatenate(evs) if want else None)

    def _forward_pipelined(self, gray, ab, idx, pos, fstream, fbases):
        dev = gray.device
        sx = _Sx(dev)
        if self._pipe_streams is None:
            self._pipe_streams = [sx.stream() for _ in range(2)]
            self._pipe_events = [sx.event() for _ in range(2)]
            for ev in self._pipe_events:
                ev.record(sx.current())                        # (torch creates the hipEvent on the first record)
        k = self._pipe_count
        self._pipe_count += 1
        st = self._pipe_streams[k & 1]
        # back-pressure: the host never runs more than two batches ahead (the results of a batch in flight cannot be recycled by the
        # caching allocator, so an unbounded run-ahead would hold one set of result tensors per enqueued batch)
        if self._pipe_done[k & 1] is not None:
            self._pipe_done[k & 1].synchronize()
        main = sx.current()
        ready = sx.event()
        ready.record(main)
        st.wait_event(ready)                                   # the inputs were produced on the caller's stream
        if self._pipe_prev is not None and self.stagger_convs > 0:
            st.wait_event(self._pipe_prev)                     # ... and the previous batch is `stagger_convs` conv launches ahead
        n = gray.shape[0]
        H, W = gray.shape[2:]
        h, w = H // self.sp, W // self.sp
        f32 = dict(device=dev, dtype=torch.float32)
        full = (torch.empty(n, self._out_channels[0], h, w, **f32), torch.empty(n, self._out_channels[1], h, w, **f32), torch.empty(n, 2, H, W, **f32),
                torch.empty(n, 9, H, W, **f32), torch.empty(n, 2, h, w, **f32), torch.empty(n, 1, h, w, **f32))
        for t in full:
            sx.keep_for(t, st)                                 # allocated on the caller's stream, written on st: no reuse before st is done
        with sx.on(st):
            self.progress_fn(self._pipe_events[k & 1], max(1, self.stagger_convs))
            try:
                self.forward_fn(gray, ab, 0, idx, pos, fstream, fbases, False, full)
            except BaseException:
                self.progress_fn(None, 0)                  # a forward that never reached the native call must not leave the event armed
                raise
            done = sx.event()
            done.record(st)
        self._pipe_done[k & 1] = done
        self._pipe_prev = self._pipe_events[k & 1]
        if st not in self._pipe_busy:
            self._pipe_busy.append(st)
        self._last_stream = st
        return full, None

    def world(self):
        if self.virtual_rank is not None:
            return self.virtual_rank
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def _exchange_events(self, ev_local, n_global, world, rank, checksum, device):
        """Per-image event counts of the global batch on every rank (+ a check that all ranks made the same draws)."""
        if world == 1 and not (self.force_gather and dist.is_available() and dist.is_initialized()):
            return ev_local.astype(np.int64)
        counts = [shard_bounds(n_global, world, r)[1] - shard_bounds(n_global, world, r)[0] for r in range(world)]
        mx = max(counts)
        buf = torch.zeros(mx + 1, dtype=torch.int32, device=device)
        if len(ev_local):
            buf[: len(ev_local)] = torch.as_tensor(ev_local, dtype=torch.int32)
        buf[mx] = int(checksum & 0x7fffffff)
        out = torch.empty(world * (mx + 1), dtype=torch.int32, device=device)
        all_gather_into(out, buf, group=self.group)
        out = out.cpu().numpy().reshape(world, mx + 1)
        if len(set(int(v) for v in out[:, mx])) != 1:
            raise RuntimeError("the ranks drew different k-means rows / fallback streams: seed NumPy, random and torch identically "
                               "on every rank before each colorize() (runner.py module docstring)")
        return np.concatenate([out[r, : counts[r]] for r in range(world)]).astype(np.int64)

    def colorize(self, gray_local, ab_local, n_global, sampled_T=0, gather=True, async_gather=False):
        """gray_local/ab_local: this rank's shard (shard_bounds order; may be empty when n_global < world).  Returns
        (pred_colors, hint_mask) of the GLOBAL batch on every rank when gather=True (ONE packed all-gather), else the
        local shard.  async_gather=True: the collective is only ENQUEUED (after this forward); the returned tensors are
        complete after wait() - a pipelined caller issues the next batch's forward meanwhile, so the xGMI transfer of
        batch k hides under the convolutions of batch k+1."""
        world, rank = self.world()
        if self.virtual_rank is not None and (gather or self.exact_fallback):
            raise ValueError("a virtual rank has no peers: gather=False and exact_fallback=False only")
        lo, hi = shard_bounds(n_global, world, rank)
        if gray_local.shape[0] != hi - lo:
            raise ValueError("rank %d expects %d images, got %d" % (rank, hi - lo, gray_local.shape[0]))
        H, W = gray_local.shape[2], gray_local.shape[3]
        h, w = H // self.sp, W // self.sp
        l = h * w
        rep = 3 if sampled_T > 0 else 1
        idx, pos = global_draws(n_global, l, self.k, self.random_hint)
        n_loc = hi - lo

        def run(fstream, fbases, want):
            if n_loc == 0:      # empty shard: nothing to compute, zero rows to contribute
                z = lambda *s: gray_local.new_zeros((0,) + s)
                return (None, None, z(2, H, W), None, None, z(1, h, w)), np.zeros(0, np.int32)
            return self._forward_local(gray_local, ab_local, sampled_T, None if idx is None else idx[lo:hi],
                                       None if pos is None else pos[lo:hi], fstream, fbases, want)

        MF = self.max_fallback

        def compute():
            """The local forward(s) of this batch -> (outputs, per-image event counts of the GLOBAL batch or None).  Consumes no generator
            state (the draws above are made once; the fallback stream is peeked), so it can run again after a re-calibration."""
            if self.random_hint:
                return run(None, None, False)[0], None
            if not self.exact_fallback:
                return run(peek_randint(l, MF), None, False)[0], None
            stream = peek_randint(l, 2 * MF)
            check = zlib.crc32(idx.tobytes()) ^ zlib.crc32(stream[:MF].tobytes())
            bases = np.zeros(n_global, np.int64)
            out, ev = run(stream, bases[lo:hi], True)
            for _ in range(n_global + 1):
                events = self._exchange_events(ev, n_global, world, rank, check, gray_local.device)
                new_bases = np.concatenate(([0], np.cumsum(events)[:-1])).astype(np.int64)
                redo = (events > 0) & (new_bases != bases)      # images that drew fallback rows from the wrong offset of the stream
                if not np.any(redo):
                    break
                bases = new_bases
                while len(stream) < int(bases.max()) + MF:
                    stream = peek_randint(l, 2 * len(stream))
                # only those images run again (an image's result does not depend on the batch it is part of); every rank takes
                # part in the next exchange, whether it had anything to redo or not
                sel = np.nonzero(redo[lo:hi])[0]
                if len(sel):
                    ts = torch.as_tensor(sel, device=gray_local.device)
                    o2, e2 = self._forward_local(gray_local[ts], ab_local[ts], sampled_T, idx[lo:hi][sel], None, stream, bases[lo:hi][sel], True)
                    rows = (ts[:, None] * rep + torch.arange(rep, device=ts.device)[None, :]).reshape(-1)
                    fixed = []
                    for k in range(6):      # outputs 1, 2, 4 carry n*rep rows (image-major); 0 has n rows; 3, 5 may be expanded views
                        t = out[k]
                        if t is None or o2[k] is None:
                            fixed.append(t); continue
                        t = t.clone() if t._base is not None or not t.is_contiguous() else t
                        t[rows if t.shape[0] == n_loc * rep else ts] = o2[k]
                        fixed.append(t)
                    out = tuple(fixed)
                    ev = ev.copy(); ev[sel] = e2
            return out, events

        self._last_stream = None
        out, events = compute()
        if self._collective_range_check(gray_local):
            out, events = compute()
        if events is not None:
            for _ in range(int(events.sum())):      # every rank consumes what the reference's single process would have
                torch.randint(l, (1,))
            self.last_events = events
        pred, mask = out[2], out[5]
        force = self.force_gather and dist.is_available() and dist.is_initialized()    # world size 1 normally skips the collective
        if not gather or (world == 1 and not force):
            return pred, mask
        if self._last_stream is not None:                    # pipelined forward: pack and gather behind it, on its stream
            sx = _Sx(pred.device)
            sx.current_outer = sx.current()
            with sx.on(self._last_stream):
                # every tensor the collective allocates here (send, recv, results) lives in the SIDE stream's pool but is read on the
                # caller's stream after wait() - the ragged unpack reads `recv` there: none may be recycled before the caller's stream
                # is done with it (round 3 kept only the two results: advisor finding)
                res = self._all_gather_packed(pred, mask, n_global, world, rank, rep, async_gather, keep=lambda t: sx.keep_for(t, sx.current_outer))
            return res
        return self._all_gather_packed(pred, mask, n_global, world, rank, rep, async_gather)

    def wait(self):
        """Complete every all-gather issued with async_gather=True and every pipelined forward (the current stream then waits for them)."""
        for work, finish in self._pending:
            work.wait()
            if finish is not None:
                finish()
        self._pending = []
        if self._pipe_busy:
            for st in self._pipe_busy:
                if not isinstance(st, _HostStream):
                    torch.cuda.current_stream(st.device).wait_stream(st)
            self._pipe_busy = []

    def _all_gather_packed(self, pred, mask, n_global, world, rank, rep, async_op=False, keep=None):
        """One collective for both results: per output row [pred (2HW) | hint_mask (hw)].  Equal shards (the bench, any batch
        that divides by the world size): the collective gathers straight into the result - pred_colors and hint_mask are
        returned as strided VIEWS of the receive buffer (row stride 2HW + hw), no unpack pass; call .contiguous() where a
        dense tensor is needed.  Ragged shards are padded to the largest and unpacked."""
        counts = [(shard_bounds(n_global, world, r)[1] - shard_bounds(n_global, world, r)[0]) * rep for r in range(world)]
        mx = max(counts)
        ps, ms = tuple(pred.shape[1:]), tuple(mask.shape[1:])
        np_, nm = int(np.prod(ps)), int(np.prod(ms))
        rows = pred.shape[0]
        equal = min(counts) == mx
        send = pred.new_empty((mx, np_ + nm)) if equal else pred.new_zeros((mx, np_ + nm))
        if rows:
            send[:rows, :np_] = pred.reshape(rows, np_)
            send[:rows, np_:] = mask.reshape(rows, nm)
        recv = pred.new_empty((world * mx, np_ + nm))
        total = sum(counts)
        if equal:
            pred_g = recv[:, :np_].unflatten(1, ps)
            mask_g = recv[:, np_:].unflatten(1, ms)
            finish = None
        else:
            pred_g = pred.new_empty((total,) + ps)
            mask_g = mask.new_empty((total,) + ms)

            def finish():
                o = 0
                for r in range(world):
                    blk = recv[r * mx: r * mx + counts[r]]
                    pred_g[o: o + counts[r]] = blk[:, :np_].reshape((counts[r],) + ps)
                    mask_g[o: o + counts[r]] = blk[:, np_:].reshape((counts[r],) + ms)
                    o += counts[r]

        if keep is not None:
            for t in (send, recv, pred_g, mask_g):
                keep(t)
        work = all_gather_into(recv, send, group=self.group, async_op=async_op)
        if async_op:
            self._pending.append((work, finish))
        elif finish is not None:
            finish()
        return pred_g, mask_g


class _DoneWork:
    """Stand-in for the work handle of a collective that has already completed."""
    def wait(self):
        return True


def all_gather_into(recv, send, group=None, async_op=False):
    """dist.all_gather_into_tensor, plus ONE special case: device tensors on the gloo backend - two or more ranks sharing one GPU, which
    is how tests/test_gpu_dist.py runs world size 2 on a single-GPU box (RCCL wants a device per rank).  gloo's all_gather_into_tensor
    takes host tensors only, so the rows are staged through pinned host buffers; the device-side ordering is the caller's: the copy out
    follows everything enqueued on the current stream, the copy back is complete when this returns."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        st = torch.cuda.current_stream(send.device)
        h_send = torch.empty(send.shape, dtype=send.dtype, pin_memory=True)
        h_send.copy_(send, non_blocking=True)
        st.synchronize()
        h_recv = torch.empty(recv.shape, dtype=recv.dtype, pin_memory=True)
        dist.all_gather_into_tensor(h_recv, h_send, group=group)
        recv.copy_(h_recv, non_blocking=True)
        st.synchronize()
        return _DoneWork() if async_op else None
    return dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)


def colorize_mixed(model, grays, abs_=None, sampled_T=0, max_batch=64):
    """BASELINE config 4 (`--no_resize`, mixed 512x512 / 768x512 ...): a LIST of (1,1,H,W) / (1,H,W) gray images of different
    sizes (H, W multiples of 16: fetch_data pads, inference.py:27-31) -> list of the model's 6-tuples, one per image, in input
    order.  The reference loops over files one at a time (inference.py:93-109); here images of equal shape run as one batch
    (at most `max_batch` at a time).  Host-side draws follow the reference's order: the k-means rows of image i are drawn i-th
    from NumPy's global state, so the result equals the per-file loop's as long as no empty-cluster fallback draw occurs
    (those come from torch's global stream in file order; the grouped batches consume them in group order)."""
    n = len(grays)
    g4 = [g.reshape(1, 1, g.shape[-2], g.shape[-1]) for g in grays]
    a4 = [None if abs_ is None else abs_[i].reshape(1, 2, g4[i].shape[2], g4[i].shape[3]) for i in range(n)]
    sp, K = model.sp_size, model.hint_num
    # draws in input order, exactly as the per-file loop would make them
    idx, pos = [], []
    for g in g4:
        l = (g.shape[2] // sp) * (g.shape[3] // sp)
        i1, p1 = global_draws(1, l, K, model.random_hint)
        idx.append(i1); pos.append(p1)
    groups = {}
    for i, g in enumerate(g4):
        groups.setdefault((g.shape[2], g.shape[3]), []).append(i)
    results = [None] * n
    rep = 3 if sampled_T > 0 else 1
    for (H, W), members in groups.items():
        for c0 in range(0, len(members), max_batch):
            sel = members[c0: c0 + max_batch]
            gray = torch.cat([g4[i] for i in sel], 0)
            ab = torch.zeros(len(sel), 2, H, W, device=gray.device) if abs_ is None else torch.cat([a4[i] for i in sel], 0)
            out = model.forward_with_draws(gray, ab, True, sampled_T,
                                           None if model.random_hint else np.concatenate([idx[i] for i in sel]),
                                           np.concatenate([pos[i] for i in sel]) if model.random_hint else None)
            for j, i in enumerate(sel):
                results[i] = tuple(None if t is None else (t[j * rep:(j + 1) * rep] if t.shape[0] == len(sel) * rep else t[j:j + 1]) for t in out)
    return results
