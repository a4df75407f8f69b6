"""Frame sharding across the GPUs of one node (one process per GPU, ``torch.distributed``;
backend ``"nccl"`` is RCCL over xGMI on ROCm, ``"gloo"`` on CPU for tests).

Every op on the hot path is independent per frame (the joint chain is inside a frame), so the
batch splits into contiguous frame blocks with NO data-path collective: rank r owns frames
``[r*F/W, (r+1)*F/W)`` (the first ``F % W`` ranks take one extra).  The only communication is the
optional reassembly of ``(positions, rotmats)`` with ONE all-gather per output, in two interchangeable forms:
RCCL's ``all_gather_into_tensor`` (uneven shards padded to the largest and trimmed) and an explicit full-mesh
``send/recv`` group (7 concurrent peer transfers per GPU, one per xGMI link; ``bench.py --gpus N`` times both).  At config-5
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


GATHER_METHODS = ("all_gather_into_tensor", "mesh_send_recv")
_default_method = "all_gather_into_tensor"


def set_default_gather_method(method: str) -> None:
    """Pick what ``all_gather_frames`` uses when no ``method`` is given (``bench.py --gpus N`` measures both and
    reports which one won on the node it ran on)."""
    global _default_method
    if method not in GATHER_METHODS:
        raise ValueError(f"unknown gather method {method!r}; choose from {GATHER_METHODS}")
    _default_method = method


def _gather_collective(local, sizes, group):
    """ONE ``all_gather_into_tensor``: shards padded to the largest so the collective applies, padding dropped after."""
    import torch
    import torch.distributed as dist

    W, mx = len(sizes), max(sizes)
    tail = tuple(local.shape[1:])
    if local.shape[0] != mx:
        pad = torch.zeros((mx - local.shape[0],) + tail, dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    out = torch.empty((W * mx,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(W)], dim=0)


def _gather_mesh(local, sizes, group):
    """Direct full mesh: every rank sends its shard to each of the W-1 peers and receives theirs straight into its
    block of the result, all 2(W-1) transfers in ONE group (``batch_isend_irecv`` = ncclGroupStart ... ncclGroupEnd on
    RCCL), so on a fully connected xGMI node each of a GPU's 7 links carries exactly one shard in each direction --
    the pattern SURVEY 8(e) asks for; no padding for uneven shards, no staging copy besides our own block."""
    import torch
    import torch.distributed as dist

    W, r = dist.get_world_size(group), dist.get_rank(group)
    tail = tuple(local.shape[1:])
    starts = [sum(sizes[:k]) for k in range(W)]
    out = torch.empty((sum(sizes),) + tail, dtype=local.dtype, device=local.device)
    local = local.contiguous()
    ops = []
    for step in range(1, W):  # peer order staggered by rank: at every step each link has one sender and one receiver
        to, frm = (r + step) % W, (r - step) % W
        g_to = dist.get_global_rank(group, to) if group is not None else to
        g_frm = dist.get_global_rank(group, frm) if group is not None else frm
        if sizes[r]:
            ops.append(dist.P2POp(dist.isend, local, g_to, group))
        if sizes[frm]:
            ops.append(dist.P2POp(dist.irecv, out[starts[frm]: starts[frm] + sizes[frm]], g_frm, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    out[starts[r]: starts[r] + sizes[r]].copy_(local)  # our own block, overlapping the transfers
    for q in reqs:
        q.wait()
    return out


def all_gather_frames(local, F_total: int, group=None, method=None):
    """All-gather shards along dim 0 (frames) into the full ``[F_total, ...]`` tensor on every rank.

    ``method``: ``"all_gather_into_tensor"`` (one RCCL collective; RCCL picks ring / direct itself) or
    ``"mesh_send_recv"`` (explicit full-mesh point-to-point, one group); default = ``set_default_gather_method``.
    Both return identical tensors.
    """
    import torch.distributed as dist

    method = method or _default_method
    if method not in GATHER_METHODS:
        raise ValueError(f"unknown gather method {method!r}; choose from {GATHER_METHODS}")
    W = dist.get_world_size(group)
    sizes = shard_sizes(F_total, W)
    assert local.shape[0] == sizes[dist.get_rank(group)], "local shard does not match shard_bounds()"
    if W == 1:
        return local.contiguous().clone() if method == "mesh_send_recv" else _gather_collective(local, sizes, group)
    return _gather_mesh(local, sizes, group) if method == "mesh_send_recv" else _gather_collective(local, sizes, group)


def sharded_apply(fn: Callable, frame_args: Sequence, F_total: int, gather: bool = True, group=None, method=None,
                  local: bool = False):
    """Run ``fn(*local_frame_args)`` on this rank's frame block.

    ``local=False``: every ``[F_total, ...]`` argument is the FULL array (or a view of it) and is sliced here.
    ``local=True``: the arguments ARE this rank's block already (``shard_bounds(F_total, W, rank)`` frames) -- what a
    data-parallel producer hands over, and what keeps a rank's resident input at 1/W of the batch.
    ``fn`` returns a tensor or a tuple of tensors with frames on dim 0.  With ``gather`` the
    outputs are reassembled on every rank, otherwise the local shards are returned.
    """
    import torch.distributed as dist

    W, r = dist.get_world_size(group), dist.get_rank(group)
    s, e = shard_bounds(F_total, W, r)
    if local:
        for a in frame_args:
            if a.shape[0] != e - s:
                raise ValueError(f"rank {r}: local shard has {a.shape[0]} frames, shard_bounds({F_total}, {W}, {r}) says {e - s}")
        out = fn(*frame_args)
    else:
        out = fn(*[a[s:e] for a in frame_args])
    outs = out if isinstance(out, tuple) else (out,)
    if gather:
        outs = tuple(all_gather_frames(o, F_total, group, method) for o in outs)
    return outs if isinstance(out, tuple) else outs[0]


def fk_sharded(rot, global_pos, offsets, parents, gather: bool = True, group=None, fk_fn=None, method=None, F_total=None):
    """``fk`` over a frame-sharded batch; each rank computes only its block with the HIP kernel.

    ``F_total=None``: ``rot [F, J, 4]`` and ``global_pos [F, 3]`` (and per-frame ``offsets [F, J, 3]``) are the FULL arrays
    (or views of them) on every rank and are sliced here.
    ``F_total=F``: they are this rank's LOCAL block of an ``F``-frame batch (``shard_bounds(F, W, rank)``): nothing but the
    shard is ever resident on the GPU -- at config 5 (2^24 frames x 22 joints on 8 GPUs) 0.76 GB of inputs and 2.2 GB of
    outputs per GPU instead of 6.1 GB of inputs on every one of them.
    ``fk_fn`` defaults to ``pymotion_amd.ops.skeleton_torch.fk`` (injectable for CPU/gloo tests).
    Reference semantics: pymotion/ops/skeleton_torch.py:16-66 applied per block.
    """
    if fk_fn is None:
        from .ops.skeleton_torch import fk as fk_fn
    local = F_total is not None
    F = int(F_total) if local else rot.shape[0]
    per_frame = offsets.dim() > 2
    if per_frame:
        return sharded_apply(lambda r_, g_, o_: fk_fn(r_, g_, o_, parents), [rot, global_pos, offsets], F, gather, group, method, local)
    return sharded_apply(lambda r_, g_: fk_fn(r_, g_, offsets, parents), [rot, global_pos], F, gather, group, method, local)
