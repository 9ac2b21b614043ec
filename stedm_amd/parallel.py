"""Data-parallel sharding of the sampling workload: latents are independent, so the global batch is
cut into contiguous per-rank shards with NO collective inside the denoising loop; one all-gather of the final latents
(RCCL over xGMI with backend "nccl", gloo on CPU) after the loop. Noise is drawn from per-SAMPLE streams keyed by the
global sample id, so a run with 1, 2, 4 or 8 ranks produces the same latents sample for sample (the reference's
batch-shaped global-RNG draw, ddim.py:122, cannot give that)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
import torch


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of samples owned by `rank`; sizes differ by at most one (first ranks take the remainder)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def per_sample_normal(seed: int, sample_ids, shape, stream: int = 0) -> torch.Tensor:
    """N(0,1) tensor [len(ids), *shape] whose row i depends only on (seed, stream, sample_ids[i]).
    Key word 1 = sample id (high 32 bits) | stream (low 32 bits): 2^32 streams per sample, so the step index of any DDIM schedule
    (ddim.py:139: up to 1000 iterations) never aliases x_T's stream 0."""
    assert 0 <= stream < (1 << 32)
    rows = []
    for sid in sample_ids:
        assert 0 <= int(sid) < (1 << 32)
        g = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, (int(sid) << 32) | stream]))
        rows.append(g.standard_normal(size=tuple(shape), dtype=np.float32))
    return torch.from_numpy(np.stack(rows)) if rows else torch.empty((0,) + tuple(shape))


def per_sample_normal_device(seed: int, first_id: int, count: int, shape, stream: int, device) -> torch.Tensor:
    """The same contract on the GPU (stedm_philox_normal: Philox4x32-10 keyed by (seed, global sample id), counter = (element group, stream),
    Box-Muller): row i of the result belongs to sample first_id + i whatever the world size. What predict_latents_sharded draws x_T and,
    for eta > 0, every step's noise from (ddim.py:122, 206) - one launch per tensor instead of a host loop over numpy generators per sample
    and step. (A different generator than per_sample_normal: a run uses one or the other, never a mix.)"""
    from . import ops
    assert 0 <= stream < (1 << 32) and 0 <= first_id and first_id + count <= (1 << 32)
    if count == 0:
        return torch.empty((0,) + tuple(shape), dtype=torch.float32, device=device)
    return ops.philox_normal(count, shape, seed, stream, device, None, first_id)


def all_gather_samples(local: torch.Tensor, global_batch: int, group=None) -> torch.Tensor:
    """Concatenate the per-rank shards (possibly of unequal length) in rank order -> [global_batch, ...] on every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(global_batch, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def all_reduce_buckets(flat: torch.Tensor, bucket_elems: int, group=None) -> int:
    """In-place SUM all-reduce of a flat gradient arena in buckets of `bucket_elems` elements (views, no copies); returns the
    world size (1 when torch.distributed is not initialised: nothing to do)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world == 1:
        return 1
    assert flat.dim() == 1 and flat.is_contiguous() and bucket_elems > 0
    works = [dist.all_reduce(flat[o:o + bucket_elems], op=dist.ReduceOp.SUM, group=group, async_op=True)
             for o in range(0, flat.numel(), bucket_elems)]
    for w in works:
        w.wait()
    return world


class BucketSchedule:
    """Gradient buckets of a flat arena for an all-reduce that overlaps the backward pass (what DistributedDataParallel's reducer does
    for the reference, train_diff.py:75): the arena is cut at PARAMETER boundaries into runs of about `bucket_elems` elements; parameters
    report completion in whatever order the backward produces them; a bucket fires — once — when its last parameter completes. The same
    cuts serve the non-overlapped path, so both issue identical collectives (bitwise-equal sums for any world size).
    `tail_from`: parameters from this index on (gradients filled after the backward, e.g. the cond stage's) form a last bucket of their own."""

    def __init__(self, offsets, sizes, bucket_elems: int, tail_from: Optional[int] = None, align: int = 1, total: Optional[int] = None):
        """align: a bucket may only end where the next parameter starts at a multiple of `align` elements (vector kernels work on the
        slices); total: the arena's padded length (the last bucket runs to it)."""
        assert len(offsets) == len(sizes) and bucket_elems > 0
        n = len(sizes)
        tail_from = n if tail_from is None else tail_from
        self.bounds: List[Tuple[int, int]] = []
        self.param_bucket: List[int] = [0] * n
        lo, acc, first = None, 0, 0
        for i in range(n):
            if lo is None:
                lo, first = offsets[i], i
            self.param_bucket[i] = len(self.bounds)
            acc += sizes[i]
            last = i == n - 1
            end = offsets[i] + sizes[i]
            aligned = last or offsets[i + 1] % align == 0
            if last or (aligned and ((acc >= bucket_elems and i < tail_from) or i + 1 == tail_from)):
                self.bounds.append((lo, (total if (last and total is not None) else (end if last else offsets[i + 1]))))
                lo, acc = None, 0
        self.count = [0] * len(self.bounds)
        for b in self.param_bucket:
            self.count[b] += 1
        self.reset()

    def reset(self) -> None:
        self.left = list(self.count)
        self.seen = [False] * len(self.param_bucket)
        self.fired = [False] * len(self.bounds)

    def done(self, param_index: int) -> Optional[int]:
        """mark one parameter complete; returns its bucket's index when that completes the bucket"""
        if self.seen[param_index]:
            return None
        self.seen[param_index] = True
        b = self.param_bucket[param_index]
        self.left[b] -= 1
        if self.left[b] == 0 and not self.fired[b]:
            self.fired[b] = True
            return b
        return None

    def pending(self) -> List[int]:
        """buckets that have not fired (to be flushed after the backward)"""
        out = [b for b, f in enumerate(self.fired) if not f]
        for b in out:
            self.fired[b] = True
        return out


def all_reduce_bounds(flat: torch.Tensor, bounds, group=None) -> int:
    """In-place SUM all-reduce of flat[lo:hi] for every (lo, hi) of `bounds` (views, no copies); returns the world size."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    world = dist.get_world_size(group)
    if world == 1:
        return 1
    works = [dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=True) for lo, hi in bounds]
    for w in works:
        w.wait()
    return world


class BatchPrefetcher:
    """Host -> HBM copy of the NEXT predict batch on a side stream while the current batch samples: `predict_step` of the reference receives
    its batch from the DataLoader on the host (modules/ldm_diffusion.py:76-79; pin_memory=True in data/dm.py:87) and Lightning copies it on the
    compute stream, in front of the step. Here the copy of batch i + 1 (889 MB at B = 64 with four 512^2 style images per sample: 16 ms over
    PCIe) runs beside the ~360 ms of sampling of batch i; the compute stream only waits for the copy's event.

        pf = BatchPrefetcher(device); h = pf.submit(first)
        for nxt in batches: cur = pf.get(h); h = pf.submit(nxt); work(cur)
    """

    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)

    def submit(self, host_batch):
        """host_batch: dict / list / tuple of (ideally pinned) host tensors. Returns a handle for get()."""
        with torch.cuda.stream(self.stream):
            if isinstance(host_batch, dict):
                dev = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in host_batch.items()}
            else:
                dev = type(host_batch)((v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for v in host_batch)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return dev, ev

    def get(self, handle):
        dev, ev = handle
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        for v in (dev.values() if isinstance(dev, dict) else dev):
            if torch.is_tensor(v):
                v.record_stream(cur)          # the caching allocator must not hand the block back while the compute stream still reads it
        return dev
