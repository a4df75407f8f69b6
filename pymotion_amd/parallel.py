"""Frame sharding across the GPUs of one node (one process per GPU, ``torch.distributed``;
backend ``"nccl"`` is RCCL over xGMI on ROCm, ``"gloo"`` on CPU for tests).

Every op on the hot path is independent per frame (the joint chain is inside a frame), so the
batch splits into contiguous frame blocks with NO data-path collective: rank r owns frames
``[r*F/W, (r+1)*F/W)`` (the first ``F % W`` ranks take one extra).  The only communication is the
optional reassembly of ``(positions, rotmats)`` with ONE all-gather per output
(``all_gather_into_tensor``; uneven shards are padded to the largest and trimmed).  At config-5
sizes the gather moves 2.2 GB per GPU and costs 20-40x the kernel, so callers
that consume the result data-parallel should pass ``gather=False`` and keep outputs sharded.
"""
from typing import Callable, Sequence, Tuple


def shard_bounds(F: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of rank ``rank`` out of ``F`` frames: sizes differ by at most one."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank/world_size {rank}/{world_size}")
    base, rem = divmod(int(F), world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(F: int, world_size: int):
    return [shard_bounds(F, world_size, r)[1] - shard_bounds(F, world_size, r)[0] for r in range(world_size)]


def all_gather_frames(local, F_total: int, group=None):
    """All-gather shards along dim 0 (frames) into the full ``[F_total, ...]`` tensor on every rank.

    One collective: shards are padded to the largest shard so ``all_gather_into_tensor`` applies,
    then the padding rows are dropped.  With even shards no copy besides the collective happens.
    """
    import torch
    import torch.distributed as dist

    W = dist.get_world_size(group)
    sizes = shard_sizes(F_total, W)
    assert local.shape[0] == sizes[dist.get_rank(group)], "local shard does not match shard_bounds()"
    mx = max(sizes)
    tail = tuple(local.shape[1:])
    if local.shape[0] != mx:
        pad = torch.zeros((mx - local.shape[0],) + tail, dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((W * mx,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(W)], dim=0)


def sharded_apply(fn: Callable, frame_args: Sequence, F_total: int, gather: bool = True, group=None):
    """Run ``fn(*local_frame_args)`` on this rank's frame block of each ``[F_total, ...]`` argument.

    ``fn`` returns a tensor or a tuple of tensors with frames on dim 0.  With ``gather`` the
    outputs are reassembled on every rank, otherwise the local shards are returned.
    """
    import torch.distributed as dist

    W, r = dist.get_world_size(group), dist.get_rank(group)
    s, e = shard_bounds(F_total, W, r)
    out = fn(*[a[s:e] for a in frame_args])
    outs = out if isinstance(out, tuple) else (out,)
    if gather:
        outs = tuple(all_gather_frames(o, F_total, group) for o in outs)
    return outs if isinstance(out, tuple) else outs[0]


def fk_sharded(rot, global_pos, offsets, parents, gather: bool = True, group=None, fk_fn=None):
    """``fk`` over a frame-sharded batch.  ``rot [F, J, 4]`` and ``global_pos [F, 3]`` are the FULL
    arrays (or views of them); each rank computes only its block with the HIP kernel.
    ``fk_fn`` defaults to ``pymotion_amd.ops.skeleton_torch.fk`` (injectable for CPU/gloo tests).
    Reference semantics: pymotion/ops/skeleton_torch.py:16-66 applied per block.
    """
    if fk_fn is None:
        from .ops.skeleton_torch import fk as fk_fn
    per_frame = offsets.dim() > 2
    if per_frame:
        return sharded_apply(lambda r_, g_, o_: fk_fn(r_, g_, o_, parents), [rot, global_pos, offsets], rot.shape[0], gather, group)
    return sharded_apply(lambda r_, g_: fk_fn(r_, g_, offsets, parents), [rot, global_pos], rot.shape[0], gather, group)
