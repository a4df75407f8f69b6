"""GPU, BASELINE.json configs[4] (8 x MI355X, 16 M frames x 22 joints, frame-sharded, one all-gather to reassemble)
as far as ONE GPU can exercise it: the kernel at the per-rank shard size (2 097 152 frames) with 2^16-frame oracle
slices, `fk_sharded(gather=True)` through a real RCCL process group (one rank: the collective and the
full-mesh path both run on the device), and bench.py's multi-rank launch paths (self-launch under
torch.distributed.run; two ranks rehearsed on one GPU)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from oracle import c_oracle as co
from pymotion_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ATOL = 1e-5
SHARD = 1 << 21  # 16 777 216 frames / 8 GPUs


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_workload(torch, F, rank, seed=0):
    """bench.py's generator: born on the device from (seed, rank), never crosses PCIe"""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed * 1000 + rank)
    rot = torch.randn((F, 22, 4), generator=g, device="cuda")
    root = torch.rand((F, 3), generator=g, device="cuda") * 4 - 2
    off_np = syn.make_offsets(22, np.random.default_rng(seed), 0.3)
    return rot, root, torch.from_numpy(off_np).cuda(), off_np


def test_fk_at_the_per_rank_shard_size_with_oracle_slices():
    import torch

    import pymotion_amd.ops.skeleton_torch as skt

    rot, root, off, off_np = _rank_workload(torch, SHARD, rank=5)
    par = torch.from_numpy(syn.PARENTS_22)
    pos, rm = skt.fk(rot, root, off, par)
    torch.cuda.synchronize()
    assert pos.shape == (SHARD, 22, 3) and rm.shape == (SHARD, 22, 3, 3)
    n = 1 << 16
    for sl in (slice(0, n), slice(SHARD // 2 - 3, SHARD // 2 - 3 + n), slice(SHARD - n, SHARD)):
        p_o, r_o = co.fk(rot[sl].cpu().numpy().astype(np.float64), root[sl].cpu().numpy().astype(np.float64),
                         off_np.astype(np.float64), syn.PARENTS_22)
        assert float(np.abs(pos[sl].cpu().numpy() - p_o).max()) <= ATOL
        assert float(np.abs(rm[sl].cpu().numpy() - r_o).max()) <= ATOL
    # size-independent properties over the WHOLE shard: root pinned, rotations orthonormal, no untouched output
    assert bool(torch.equal(pos[:, 0], root))
    worst = 0.0
    for i in range(3):
        for k in range(i, 3):
            d = (rm[..., i, :] * rm[..., k, :]).sum(-1) - (1.0 if i == k else 0.0)
            worst = max(worst, float(d.abs().max()))
    assert worst <= 1e-5
    assert bool(torch.isfinite(pos).all()) and bool(torch.isfinite(rm).all())
    # the last tile of the shard is a full one for 2^21 % 20 != 0? (it is ragged: 2^21 = 104857 * 20 + 12): covered above


def test_fk_sharded_through_a_real_rccl_group_one_rank():
    """`fk_sharded(gather=True)` with backend "nccl" (= RCCL): world size 1 is what one GPU allows, but the collective
    call, the full-mesh call and the device-side block placement are the real ones."""
    import torch
    import torch.distributed as dist

    from pymotion_amd import parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        F = 1 << 18
        rot, root, off, off_np = _rank_workload(torch, F, rank=0, seed=3)
        par = torch.from_numpy(syn.PARENTS_22)
        ref_p, ref_r = None, None
        for method in parallel.GATHER_METHODS:
            pos, rm = parallel.fk_sharded(rot, root, off, par, gather=True, method=method)
            torch.cuda.synchronize()
            assert pos.shape == (F, 22, 3) and rm.shape == (F, 22, 3, 3) and pos.is_cuda
            if ref_p is None:
                ref_p, ref_r = pos, rm
                n = 1 << 12
                p_o, r_o = co.fk(rot[:n].cpu().numpy().astype(np.float64), root[:n].cpu().numpy().astype(np.float64),
                                 off_np.astype(np.float64), syn.PARENTS_22)
                assert float(np.abs(pos[:n].cpu().numpy() - p_o).max()) <= ATOL
                assert float(np.abs(rm[:n].cpu().numpy() - r_o).max()) <= ATOL
            else:
                assert bool(torch.equal(pos, ref_p)) and bool(torch.equal(rm, ref_r))
        # one real RCCL collective on device memory at the shard's byte size per output (pos: 2^21 x 264 B = 554 MB)
        big = torch.empty((SHARD, 22, 3), device="cuda").normal_()
        out = parallel.all_gather_frames(big, SHARD, method="all_gather_into_tensor")
        assert bool(torch.equal(out, big))
    finally:
        dist.destroy_process_group()


def _bench(args, env_extra=None, launcher=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_plain_invocation_with_two_ranks_launches_itself():
    """`python bench.py --gpus 2` (no torchrun around it) must re-launch under torch.distributed.run instead of exiting;
    on this one-GPU box the two ranks share cuda:0 over gloo (--dry-run-shared-gpu), which rehearses everything but RCCL:
    rendezvous, per-rank seeds, oracle slices on every rank, both gather methods, the combined figure, the JSON line."""
    line = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm-ms", "0", "--frames-per-gpu", "65536",
                   "--oracle-slice-frames", "4096", "--dry-run-shared-gpu"])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3
    assert line["value"] > 0 and "dry_run_shared_gpu" in line
    assert line["max_abs_err_vs_oracle_slice"]["value"] <= ATOL
    g = line["gather"]
    assert set(g["methods"]) == {"all_gather_into_tensor", "mesh_send_recv"}
    for m in g["methods"].values():
        assert m.get("own_block_intact") is True, m
    assert g["combined_compute_plus_gather"]["ms"] >= g["compute_only_ms"]
    assert "fk_kernel" in line["roofline"]["kernel"]


def test_bench_eight_ranks_rehearsal_of_the_drivers_command():
    """the exact command the driver will issue on an 8-GPU node -- `torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8
    --steps K --warmup W` -- rehearsed end to end on this one-GPU box: eight ranks share cuda:0 over gloo (--dry-run-shared-gpu),
    32 768 frames per rank.  Without --frames-per-gpu the same command runs BASELINE configs[4] (16 777 216 // 8 frames per GPU)."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    line = _bench(["--gpus", "8", "--steps", "3", "--warmup", "1", "--prewarm-ms", "0", "--frames-per-gpu", "32768",
                   "--oracle-slice-frames", "2048", "--dry-run-shared-gpu"], launcher=launcher)
    assert line["n_gpus"] == 8 and line["steps"] == 3 and line["scaling"] == "weak"  # (an explicit per-GPU size: weak scaling)
    assert line["config"]["frames_total"] == 8 * 32768
    assert line["max_abs_err_vs_oracle_slice"]["value"] <= ATOL
    g = line["gather"]
    assert set(g["methods"]) == {"all_gather_into_tensor", "mesh_send_recv"}
    for m in g["methods"].values():
        assert m.get("own_block_intact") is True, m
    assert g["received_GB_per_gpu"] == pytest.approx(7 * 32768 * 22 * 48 / 1e9)


def test_bench_defaults_to_config5_when_more_than_one_gpu_is_asked_for():
    """`--gpus N` with N > 1 and no --frames-per-gpu is BASELINE configs[4]: 16 777 216 // N frames per GPU, labelled strong
    scaling (two ranks sharing this one GPU: 2^23 frames each, 17.7 GB of outputs in all -- the gather is skipped)."""
    line = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--prewarm-ms", "0", "--oracle-slice-frames", "2048",
                   "--dry-run-shared-gpu", "--no-gather"])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["frames_per_gpu"] == (1 << 24) // 2 and line["config"]["frames_total"] == 1 << 24
    assert "configs[4]" in line["config"]["workload"]
    assert line["max_abs_err_vs_oracle_slice"]["value"] <= ATOL


def test_bench_under_torchrun_one_rank_rccl():
    """the driver's launch line with --nproc-per-node 1: process group "nccl" on the device, oracle slice of 2^16"""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    line = _bench(["--gpus", "1", "--steps", "5", "--warmup", "2", "--prewarm-ms", "50", "--frames-per-gpu", str(1 << 18),
                   "--no-cpu-baseline", "--no-secondary"], launcher=launcher)
    assert line["n_gpus"] == 1
    assert line["max_abs_err_vs_oracle_slice"]["frames_per_rank"] == 1 << 16
    assert line["max_abs_err_vs_oracle_slice"]["value"] <= ATOL
    assert line["roofline"]["kernel"].startswith("void pm::fk_kernel<16, true, false, 0, false, false, 49>")  # (49 = DYN | RESID | BIG_RESID: the 22-joint body is shallow)


def test_bench_line_survives_a_reassembly_that_never_finishes():
    """the gather measurements run last and under a watchdog: with a zero budget they are abandoned at once, every rank
    exits cleanly and the bench line still carries the compute-only result and the oracle check"""
    line = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm-ms", "0", "--frames-per-gpu", "65536",
                   "--oracle-slice-frames", "4096", "--dry-run-shared-gpu", "--gather-timeout-s", "0"])
    assert line["n_gpus"] == 2 and line["value"] > 0
    assert line["max_abs_err_vs_oracle_slice"]["value"] <= ATOL
    assert "error" in line["gather"] and "abandoned" in line["gather"]["error"]


def test_fk_on_all_16m_frames_of_config5_in_one_call():
    """BASELINE.json configs[4] names 16 777 216 frames x 22 joints.  Sharded over 8 GPUs that is 2^21 per rank (tested
    above); one MI355X has the memory for the WHOLE batch (rot 5.9 GB, pos 4.4 GB, rotmats 13.3 GB), so the workload itself --
    and the 64-bit element offsets it needs: 3.3e9 floats of rotation matrices -- runs here in one call, with oracle slices
    from the start, the 2^31-element boundary region and the end, and the per-rank blocks compared with separate calls."""
    import torch

    import pymotion_amd.ops.skeleton_torch as skt

    F = 1 << 24
    g = torch.Generator(device="cuda")
    g.manual_seed(2024)
    rot = torch.randn((F, 22, 4), generator=g, device="cuda")
    root = torch.rand((F, 3), generator=g, device="cuda") * 4 - 2
    off_np = syn.make_offsets(22, np.random.default_rng(1), 0.3)
    off = torch.from_numpy(off_np).cuda()
    par = torch.from_numpy(syn.PARENTS_22)
    pos, rm = skt.fk(rot, root, off, par)
    torch.cuda.synchronize()
    n = 1 << 13
    f_cross = (1 << 31) // (22 * 9)  # the frame whose rotation matrices straddle element 2^31
    for s0 in (0, f_cross - n // 2, F // 2 + 7, F - n):
        sl = slice(s0, s0 + n)
        p_o, r_o = co.fk(rot[sl].cpu().numpy().astype(np.float64), root[sl].cpu().numpy().astype(np.float64),
                         off_np.astype(np.float64), syn.PARENTS_22)
        assert float(np.abs(pos[sl].cpu().numpy() - p_o).max()) <= ATOL
        assert float(np.abs(rm[sl].cpu().numpy() - r_o).max()) <= ATOL
    assert bool(torch.equal(pos[:, 0], root))
    # rank 6's block of the 8-way sharding, computed on its own, is the same bits
    s, e = 6 * SHARD, 7 * SHARD
    p6, r6 = skt.fk(rot[s:e], root[s:e], off, par)
    assert bool(torch.equal(p6, pos[s:e])) and bool(torch.equal(r6, rm[s:e]))
