"""CPU-only, world sizes 2, 3 and 8 over gloo: the frame-sharding + all-gather path of pymotion_amd.parallel.
The per-rank compute is injected (the CPU oracle stands in for the HIP kernel, as the checker),
so what is tested here is exactly the N>1 logic: block bounds, uneven shards, reassembly order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from pymotion_amd import parallel
from pymotion_amd import synthetic as syn


def test_shard_bounds_cover_everything_once():
    for F in (0, 1, 7, 8, 9, 1000, 1 << 20):
        for W in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(F, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == F
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == parallel.shard_sizes(F, W)
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_fk(rot, gpos, off, parents):
    from oracle import c_oracle as co

    p, r = co.fk(rot.numpy(), gpos.numpy(), off.numpy(), parents.numpy())
    return torch.from_numpy(p), torch.from_numpy(r)


def _worker(rank, world, port, F, q, method="all_gather_into_tensor"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rot, root, off, parents = syn.fk_workload(F, seed=42)
        rot, root, off, parents = map(torch.from_numpy, (rot, root, off, parents))
        pos, rm = parallel.fk_sharded(rot, root, off, parents, gather=True, fk_fn=_oracle_fk, method=method)
        lp, lr = parallel.fk_sharded(rot, root, off, parents, gather=False, fk_fn=_oracle_fk)
        s, e = parallel.shard_bounds(F, world, rank)
        full_p, full_r = _oracle_fk(rot, root, off, parents)
        ok = (
            pos.shape == full_p.shape and torch.equal(pos, full_p) and torch.equal(rm, full_r)
            and lp.shape[0] == e - s and torch.equal(lp, full_p[s:e]) and torch.equal(lr, full_r[s:e])
        )
        # per-frame offsets take the 3-argument sharded path
        offs = off.unsqueeze(0).repeat(F, 1, 1) * torch.linspace(0.5, 1.5, F).view(F, 1, 1)
        parallel.set_default_gather_method(method)  # ... and the process-wide default is honoured
        p2, _ = parallel.fk_sharded(rot, root, offs, parents, gather=True, fk_fn=_oracle_fk)
        ok = ok and torch.equal(p2, _oracle_fk(rot, root, offs, parents)[0])
        # a rank that only ever holds ITS block (F_total given): same results, nothing but the shard handed over
        pl, rl = parallel.fk_sharded(rot[s:e].clone(), root[s:e].clone(), off, parents, gather=True, fk_fn=_oracle_fk, method=method, F_total=F)
        ok = ok and torch.equal(pl, full_p) and torch.equal(rl, full_r)
        pl2, _ = parallel.fk_sharded(rot[s:e].clone(), root[s:e].clone(), offs[s:e].clone(), parents, gather=False, fk_fn=_oracle_fk, F_total=F)
        ok = ok and torch.equal(pl2, p2[s:e])
        try:
            parallel.fk_sharded(rot[: (e - s) + 1], root[: (e - s) + 1], off, parents, gather=False, fk_fn=_oracle_fk, F_total=F)
            ok = False  # a block of the wrong size must be refused, not silently gathered out of place
        except ValueError:
            pass
        # both reassemblies give the same tensor, shard by shard
        a1 = parallel.all_gather_frames(lp, F, method="all_gather_into_tensor")
        a2 = parallel.all_gather_frames(lp, F, method="mesh_send_recv")
        ok = ok and torch.equal(a1, a2)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def _run(world, F, method):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, F, q, method)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = dict(q.get(timeout=5) for _ in range(world))
    assert res == {r: True for r in range(world)}


@pytest.mark.parametrize("method", parallel.GATHER_METHODS)
@pytest.mark.parametrize("F", [64, 101])  # even and uneven shards
def test_fk_sharded_gloo_world2(F, method):
    _run(2, F, method)


def test_fk_sharded_gloo_world3_with_an_empty_shard():
    """F < world: one rank owns no frames; the full-mesh gather must not post zero-size transfers"""
    _run(3, 2, "mesh_send_recv")
    _run(3, 2, "all_gather_into_tensor")


@pytest.mark.parametrize("method", parallel.GATHER_METHODS)
@pytest.mark.parametrize("F", [128, 203])  # even shards / five ranks with one frame more than the other three
def test_fk_sharded_gloo_world8(F, method):
    """the world size of the target node: the staggered peer order (r +- step) % 8 of the full-mesh gather, seven
    sends and seven receives per rank in one group, and the padded collective with uneven shards"""
    _run(8, F, method)


def test_fk_sharded_gloo_world8_with_empty_shards():
    """F = 5 < world 8: three ranks own nothing; neither method may post a zero-size transfer or wait for one"""
    _run(8, 5, "mesh_send_recv")
    _run(8, 5, "all_gather_into_tensor")


def test_unknown_gather_method_is_rejected():
    with pytest.raises(ValueError):
        parallel.set_default_gather_method("ring")
