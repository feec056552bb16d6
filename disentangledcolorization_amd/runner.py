"""Batch-sharded multi-GPU colorization: one process per GPU, one RCCL all-gather of the results.

The reference has no batched/multi-GPU inference (inference.py:93 loops over files with batch 1;
nn.DataParallel :76-82).  Every image is independent end to end, so the global batch
is cut into contiguous shards, each rank runs the HIP forward on its shard with the full weights,
and `pred_colors` (+ `hint_mask`) are all-gathered over xGMI (torch.distributed backend "nccl" =
RCCL on ROCm; "gloo" in the CPU tests).  Host-side draws (k-means initial rows, random hints) are
made once for the GLOBAL batch in image order from the same generators the reference consumes, and
sliced per rank, so results do not depend on the number of GPUs.
"""
import random

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


class ShardedColorizer:
    """forward_fn(gray_local, ab_local, sampled_T, init_idx_local, hint_pos_local) -> 6-tuple like the model.

    For the product path forward_fn = AnchorColorProb.forward_with_draws (HIP); tests inject a CPU function.
    """

    def __init__(self, forward_fn, n_clusters=8, random_hint=False, sp_size=16, group=None, micro_batches=1):
        self.forward_fn = forward_fn
        self.k, self.random_hint, self.sp = n_clusters, random_hint, sp_size
        self.group = group
        # micro_batches > 1: the local shard is cut into that many slices, each issued on its own HIP stream, so the
        # latency-bound token path / k-means of one slice (a few CUs busy) overlaps with the conv stacks of another
        self.micro = max(1, int(micro_batches))
        self._streams = None
        self._pending = []          # outstanding asynchronous all-gathers (async_gather=True)

    @classmethod
    def from_model(cls, model, group=None, micro_batches=1):
        fn = lambda g, a, T, idx, pos: model.forward_with_draws(g, a, True, T, init_idx=idx, hint_pos=pos)
        return cls(fn, model.hint_num, model.random_hint, model.sp_size, group, micro_batches)

    def _forward_local(self, gray, ab, sampled_T, idx, pos):
        n = gray.shape[0]
        m = min(self.micro, n)
        if m <= 1 or not gray.is_cuda:
            return self.forward_fn(gray, ab, sampled_T, idx, pos)
        if self._streams is None or len(self._streams) < m:
            self._streams = [torch.cuda.Stream(device=gray.device) for _ in range(m)]
        main = torch.cuda.current_stream(gray.device)
        parts = []
        for i in range(m):
            lo, hi = shard_bounds(n, m, i)
            st = self._streams[i]
            st.wait_stream(main)                       # inputs were produced on the caller's stream
            with torch.cuda.stream(st):
                parts.append(self.forward_fn(gray[lo:hi], ab[lo:hi], sampled_T,
                                             None if idx is None else idx[lo:hi], None if pos is None else pos[lo:hi]))
        for i in range(m):
            main.wait_stream(self._streams[i])
            for t in parts[i]:
                t.record_stream(main)                  # allocated on the side stream, consumed on the caller's
        return tuple(torch.cat([p[k] for p in parts], 0) for k in range(6))

    def world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def colorize(self, gray_local, ab_local, n_global, sampled_T=0, gather=True, async_gather=False):
        """gray_local/ab_local: this rank's shard (shard_bounds order).  Returns (pred_colors, hint_mask) of the
        GLOBAL batch on every rank when gather=True (one all-gather each), else the local shard.
        async_gather=True: the collectives are only ENQUEUED (on the communication stream, after this forward); the
        returned tensors are complete after wait() - a pipelined caller issues the next batch's forward meanwhile, so the
        xGMI transfer of batch k hides under the convolutions of batch k+1."""
        world, rank = self.world()
        lo, hi = shard_bounds(n_global, world, rank)
        if gray_local.shape[0] != hi - lo:
            raise ValueError("rank %d expects %d images, got %d" % (rank, hi - lo, gray_local.shape[0]))
        h, w = gray_local.shape[2] // self.sp, gray_local.shape[3] // self.sp
        idx, pos = global_draws(n_global, h * w, self.k, self.random_hint)
        out = self._forward_local(gray_local, ab_local, sampled_T,
                                  None if idx is None else idx[lo:hi], None if pos is None else pos[lo:hi])
        pred, mask = out[2], out[5]
        if not gather or not (dist.is_available() and dist.is_initialized()):
            return pred, mask
        return self._all_gather(pred, n_global, world, async_gather), self._all_gather(mask, n_global, world, async_gather)

    def wait(self):
        """Complete every all-gather issued with async_gather=True (the current stream then waits for them)."""
        for w in self._pending:
            w.wait()
        self._pending = []

    def _all_gather(self, local, n_global, world, async_op=False):
        counts = [shard_bounds(n_global, world, r)[1] - shard_bounds(n_global, world, r)[0] for r in range(world)]
        mine = counts[self.world()[1]]
        rep = local.shape[0] // mine if mine else 1          # 3 outputs per image in diverse mode
        per = [c * rep for c in counts]
        if len(set(per)) == 1:   # equal shards: one fused collective
            out = local.new_empty((per[0] * world,) + tuple(local.shape[1:]))
            work = dist.all_gather_into_tensor(out, local.contiguous(), group=self.group, async_op=async_op)
            if async_op:
                self._pending.append(work)
            return out
        mx = max(per)            # ragged: pad to the largest shard
        pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
        pad[: local.shape[0]] = local
        out = local.new_empty((mx * world,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, pad, group=self.group)
        return torch.cat([out[r * mx: r * mx + per[r]] for r in range(world)], 0)
