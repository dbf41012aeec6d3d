"""Multi-GPU harness: one process per GPU, batch sharded, one all-gather of the landmarks.

The reference is single-process (prediction.py:208 predicts a batch of one); this is new work
(SURVEY.md section 8 row E).  Every face crop is independent, so ranks process contiguous
slices of the batch with replicated weights and meet only in one all-gather of the
[B/R, C, 2] float64 landmark tensor (74 KB per rank at B/R = 64): latency-bound, one step.
Backend "nccl" is RCCL on ROCm (xGMI between the 8 GPUs of a node); "gloo" on CPU for tests.
"""
from __future__ import annotations

import os


def shard_range(total: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of `total` items owned by `rank` (first `total % world` ranks
    get one extra item)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process when unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """Initialise torch.distributed from the env; returns (rank, local_rank, world)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def all_gather_landmarks(local, total: int, group=None):
    """Gather per-rank landmark shards [b_r, C, 2] (contiguous slices by `shard_range`) into
    the full [total, C, 2] tensor on every rank.  Equal shards use one
    `all_gather_into_tensor`; ragged shards are padded to the largest shard first."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        if local.shape[0] != total:
            raise ValueError("single process: shard has %d rows, expected %d" % (local.shape[0], total))
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(total, r, world) for r in range(world)]
    counts = [hi - lo for lo, hi in sizes]
    mx = max(counts)
    rank = dist.get_rank(group)
    if local.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d rows, its shard has %d" % (rank, local.shape[0], counts[rank]))
    tail = tuple(local.shape[1:])
    if mx != local.shape[0]:
        pad = torch.zeros((mx - local.shape[0],) + tail, dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], 0)
    else:
        send = local.contiguous()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # dry runs of the multi-rank flow without RCCL (ranks sharing a GPU): gloo gathers host tensors
        host = torch.empty((world * mx,) + tail, dtype=local.dtype)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        out = host.to(local.device)
    else:
        out = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, send, group=group)
    if all(c == mx for c in counts):
        return out
    return torch.cat([out[r * mx:r * mx + counts[r]] for r in range(world)], 0)


class PendingGather:
    """A landmark all-gather in flight (`all_gather_landmarks_async`): `wait()` returns the full [total, C, 2] tensor.
    With RCCL the wait is a stream dependency, not a host block: kernels launched on the current stream BEFORE the call
    are not held up by the collective."""

    def __init__(self, work, out, counts, mx, device):
        self._work, self._out, self._counts, self._mx, self._device = work, out, counts, mx, device

    def wait(self):
        import torch
        if self._work is not None:
            self._work.wait()
            self._work = None
        out = self._out
        if self._device is not None:          # gloo dry run with CUDA shards: the gather ran on host tensors
            out = out.to(self._device)
        if all(c == self._mx for c in self._counts):
            return out
        return torch.cat([out[r * self._mx:r * self._mx + c] for r, c in enumerate(self._counts)], 0)


def all_gather_landmarks_async(local, total: int, group=None, single_rank_collective=False):
    """`all_gather_landmarks` without waiting for it: the collective is queued behind the work already on the current
    stream (RCCL runs it on its own stream) and a `PendingGather` comes back.  The shard is COPIED first, so the caller
    may overwrite `local` (the next batch's decode) while the exchange is in flight.  A pipeline waits for the gather of
    batch i after it has launched batch i + 1 (bench.py): the exchange then hides behind that batch's kernels instead
    of stalling the stream for its latency once per batch."""
    import torch
    import torch.distributed as dist
    alone = not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1
    if alone and not (single_rank_collective and dist.is_available() and dist.is_initialized()):
        # (single_rank_collective: run the exchange even with one rank -- tests/test_gpu_rccl.py drives the RCCL call
        #  sequence of the pipelined step on a one-GPU box that way)
        if local.shape[0] != total:
            raise ValueError("single process: shard has %d rows, expected %d" % (local.shape[0], total))
        return PendingGather(None, local, [total], total, None)
    world = dist.get_world_size(group)
    counts = [hi - lo for lo, hi in (shard_range(total, r, world) for r in range(world))]
    mx = max(counts)
    rank = dist.get_rank(group)
    if local.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d rows, its shard has %d" % (rank, local.shape[0], counts[rank]))
    tail = tuple(local.shape[1:])
    if mx != local.shape[0]:
        send = torch.cat([local, torch.zeros((mx - local.shape[0],) + tail, dtype=local.dtype, device=local.device)], 0)
    else:
        send = local.clone()
    if local.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty((world * mx,) + tail, dtype=local.dtype)
        work = dist.all_gather_into_tensor(host, send.cpu(), group=group, async_op=True)
        return PendingGather(work, host, counts, mx, local.device)
    out = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, send, group=group, async_op=True)
    return PendingGather(work, out, counts, mx, None)


def sharded_predict(predict_fn, batch, total: int, group=None):
    """Run `predict_fn(local_batch) -> [b_r, C, 2]` on this rank's slice and gather.
    `batch` is this rank's slice (already resident on its device)."""
    return all_gather_landmarks(predict_fn(batch), total, group)
