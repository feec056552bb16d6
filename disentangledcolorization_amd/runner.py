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
                 exact_fallback=True, max_fallback=None):
        self.forward_fn = forward_fn
        self.k, self.random_hint, self.sp = n_clusters, random_hint, sp_size
        self.group = group
        # micro_batches > 1: the local shard is cut into that many slices, each issued on its own HIP stream, so the
        # latency-bound token path / k-means of one slice (a few CUs busy) overlaps with the conv stacks of another
        self.micro = max(1, int(micro_batches))
        self.exact_fallback = bool(exact_fallback)
        self.max_fallback = int(max_fallback) if max_fallback else 20 * n_clusters
        self._streams = None
        self._pending = []          # outstanding asynchronous all-gathers (async_gather=True): (work, finish callback)
        self.last_events = None     # per-image empty-cluster draws of the GLOBAL batch of the latest exact forward

    @classmethod
    def from_model(cls, model, group=None, micro_batches=1, exact_fallback=None):
        fn = lambda g, a, T, idx, pos, fs, fb, want: model.forward_once(g, a, True, T, idx, pos, fs, fb, want)
        exact = model.sync_kmeans_events if exact_fallback is None else exact_fallback
        return cls(fn, model.hint_num, model.random_hint, model.sp_size, group, micro_batches, exact, model.max_fallback())

    # ---- local forward (optionally as micro-batches on separate streams) ----------------------------------------
    def _forward_local(self, gray, ab, sampled_T, idx, pos, fstream, fbases, want):
        n = gray.shape[0]
        m = min(self.micro, n)
        if m <= 1 or not gray.is_cuda:
            return self.forward_fn(gray, ab, sampled_T, idx, pos, fstream, fbases, want)
        if self._streams is None or len(self._streams) < m:
            self._streams = [torch.cuda.Stream(device=gray.device) for _ in range(m)]
        main = torch.cuda.current_stream(gray.device)
        parts, evs = [], []
        for i in range(m):
            lo, hi = shard_bounds(n, m, i)
            st = self._streams[i]
            st.wait_stream(main)                       # inputs were produced on the caller's stream
            with torch.cuda.stream(st):
                o, e = self.forward_fn(gray[lo:hi], ab[lo:hi], sampled_T, None if idx is None else idx[lo:hi],
                                       None if pos is None else pos[lo:hi], fstream, None if fbases is None else fbases[lo:hi], want)
                parts.append(o); evs.append(e)
        for i in range(m):
            main.wait_stream(self._streams[i])
            for t in parts[i]:
                t.record_stream(main)                  # allocated on the side stream, consumed on the caller's
        out = tuple(torch.cat([p[k] for p in parts], 0) for k in range(6))
        return out, (np.concatenate(evs) if want else None)

    def world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def _exchange_events(self, ev_local, n_global, world, rank, checksum, device):
        """Per-image event counts of the global batch on every rank (+ a check that all ranks made the same draws)."""
        if world == 1 and not (os.environ.get("DISCO_FORCE_GATHER") == "1" and dist.is_available() and dist.is_initialized()):
            return ev_local.astype(np.int64)
        counts = [shard_bounds(n_global, world, r)[1] - shard_bounds(n_global, world, r)[0] for r in range(world)]
        mx = max(counts)
        buf = torch.zeros(mx + 1, dtype=torch.int32, device=device)
        if len(ev_local):
            buf[: len(ev_local)] = torch.as_tensor(ev_local, dtype=torch.int32)
        buf[mx] = int(checksum & 0x7fffffff)
        out = torch.empty(world * (mx + 1), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(out, buf, group=self.group)
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
        if self.random_hint:
            out, _ = run(None, None, False)
        elif not self.exact_fallback:
            out, _ = run(peek_randint(l, MF), None, False)
        else:
            stream = peek_randint(l, 2 * MF)
            check = zlib.crc32(idx.tobytes()) ^ zlib.crc32(stream[:MF].tobytes())
            bases = np.zeros(n_global, np.int64)
            for _ in range(n_global + 1):
                while len(stream) < int(bases.max()) + MF:
                    stream = peek_randint(l, 2 * len(stream))
                out, ev = run(stream, bases[lo:hi], True)
                events = self._exchange_events(ev, n_global, world, rank, check, gray_local.device)
                new_bases = np.concatenate(([0], np.cumsum(events)[:-1])).astype(np.int64)
                if not np.any((events > 0) & (new_bases != bases)):
                    break
                bases = new_bases
            for _ in range(int(events.sum())):      # every rank consumes what the reference's single process would have
                torch.randint(l, (1,))
            self.last_events = events
        pred, mask = out[2], out[5]
        # world size 1 normally skips the collective; DISCO_FORCE_GATHER=1 runs it anyway when a process group exists (the only
        # way to exercise the RCCL path - packed send buffer, async work handle, unpack - on a single-GPU box: tests/test_gpu_dist.py)
        force = os.environ.get("DISCO_FORCE_GATHER") == "1" and dist.is_available() and dist.is_initialized()
        if not gather or (world == 1 and not force):
            return pred, mask
        return self._all_gather_packed(pred, mask, n_global, world, rank, rep, async_gather)

    def wait(self):
        """Complete every all-gather issued with async_gather=True (the current stream then waits for them)."""
        for work, finish in self._pending:
            work.wait()
            if finish is not None:
                finish()
        self._pending = []

    def _all_gather_packed(self, pred, mask, n_global, world, rank, rep, async_op=False):
        """One collective for both results: per output row [pred (2HW) | hint_mask (hw)], shards padded to the largest."""
        counts = [(shard_bounds(n_global, world, r)[1] - shard_bounds(n_global, world, r)[0]) * rep for r in range(world)]
        mx = max(counts)
        ps, ms = tuple(pred.shape[1:]), tuple(mask.shape[1:])
        np_, nm = int(np.prod(ps)), int(np.prod(ms))
        send = pred.new_zeros((mx, np_ + nm))
        rows = pred.shape[0]
        if rows:
            send[:rows, :np_] = pred.reshape(rows, np_)
            send[:rows, np_:] = mask.reshape(rows, nm)
        recv = pred.new_empty((world * mx, np_ + nm))
        total = sum(counts)
        pred_g = pred.new_empty((total,) + ps)
        mask_g = mask.new_empty((total,) + ms)

        def finish():
            o = 0
            for r in range(world):
                blk = recv[r * mx: r * mx + counts[r]]
                pred_g[o: o + counts[r]] = blk[:, :np_].reshape((counts[r],) + ps)
                mask_g[o: o + counts[r]] = blk[:, np_:].reshape((counts[r],) + ms)
                o += counts[r]

        work = dist.all_gather_into_tensor(recv, send, group=self.group, async_op=async_op)
        if async_op:
            self._pending.append((work, finish))
        else:
            finish()
        return pred_g, mask_g
