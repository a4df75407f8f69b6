#!/usr/bin/env python3
"""bench.py -- fk() frames/s on MI355X, 22-joint skeleton (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the fk hot path over one device-resident batch of synthetic frames, fp32, quaternions not
pre-normalised, metre-scale offsets.  N = 1: BASELINE.json configs[1], 2^20 frames x 22 joints.  N > 1: BASELINE.json
configs[4], 16 777 216 frames x 22 joints sharded over the N GPUs (16 777 216 // N frames per GPU: the TOTAL is fixed,
"scaling": "strong"; --frames-per-gpu overrides it and makes the run a weak-scaling one).  Frames shard across ranks
with no data-path collective.  `python bench.py --gpus N` with N > 1 launches itself under torch.distributed.run.  Multi-GPU runs also report, after the timed region and never inside `value`, the reassembly
all-gather (ms, xGMI GB/s per GPU vs the link roofline, both implementations) and compute + gather combined.
Rank 0 prints ONE JSON line.  Extra objects in that line:

  roofline      the fk kernel against the HBM roofline: algorithmic bytes (64*J+12 per frame)
                / average launch time measured with HIP events on the launch stream.
  cpu_baseline  the cost-equivalent NumPy restatement of the reference's fk (oracle/numpy_ref.py,
                kind "port") timed on this box's host cores over the same workload (N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 GB/s is what a float4 copy reaches
# xGMI: 7 point-to-point links per GPU, 153.6 GB/s each counting both directions = 76.8 GB/s per direction.  An
# all-gather is receive-bound: every GPU takes W-1 shards, one per link, so its roofline is 76.8 GB/s x (W-1) links.
XGMI_LINK_GBPS_PER_DIRECTION = 76.8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed back-to-back launches before the W warmup steps so that the GPU's clocks/power state "
                         "settle (the first ~20 ms after idle run ~15%% slower); 0 disables")
    ap.add_argument("--frames-per-gpu", type=int, default=0,
                    help="0 (default): 2^20 at N = 1 (BASELINE configs[1]); 16 777 216 // N at N > 1 (BASELINE configs[4], strong scaling)")
    ap.add_argument("--joints", type=int, default=22, choices=[22, 52])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=1 << 20)
    ap.add_argument("--no-gather", action="store_true",
                    help="N>1: skip the reassembly measurements.  By default every multi-GPU run reports, AFTER the timed region "
                         "and never inside `value`: (ii) the all-gather of (pos, rotmats) -- ms and achieved xGMI GB/s per GPU for "
                         "both RCCL's all_gather_into_tensor and the direct full-mesh send/recv -- and (iii) compute + gather combined")
    ap.add_argument("--gather", action="store_true", help=argparse.SUPPRESS)  # round-1 spelling, now the default
    ap.add_argument("--gather-timeout-s", type=int, default=180,
                    help="N>1: the reassembly measurements run last, under a watchdog; past this they are abandoned and the bench line "
                         "is printed without them")
    ap.add_argument("--dry-run-shared-gpu", action="store_true",
                    help="launch-path rehearsal on a box with fewer GPUs than ranks: every rank uses cuda:0 and the process group is "
                         "gloo (RCCL refuses two ranks on one device).  Numbers from such a run mean nothing; the JSON says so")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short measurements of BASELINE configs 3 and 4 (N=1)")
    ap.add_argument("--rccl-debug-file", default="",
                    help="(default at N > 1: profiles/rccl_debug_N<N>; 'none' = off) write RCCL's own log (NCCL_DEBUG=INFO, subsystems INIT,COLL,P2P: topology, channels, the algorithm / protocol "
                         "each collective ran with) to PATH.<host>.<pid> per rank -- the record of what the reassembly actually did")
    ap.add_argument("--oracle-slice-frames", type=int, default=1 << 16,
                    help="N>1: frames of its own shard every rank checks against the CPU oracle (SURVEY 8d config 5: 2^16)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--batches", type=int, default=0,
                    help="independently allocated synthetic batches the steps cycle through (step i takes batch i %% B); 0 = 3 up to 2^21 frames per GPU, "
                         "1 beyond.  A launch's time depends on WHERE the allocator put its arrays (profiles/r06_levels.txt: 150 or 167 us on the "
                         "same J = 52 kernel, allocation by allocation): with several batches ms_per_step averages over placements instead of drawing one")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher (one rank per GPU, same
    flags), exactly the command the driver uses."""
    import socket

    with socket.socket() as sk_:
        sk_.bind(("127.0.0.1", 0))
        port = sk_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info

        return {i.get("internal_api", "?"): i.get("num_threads") for i in threadpool_info()}
    except Exception:
        return None


def cpu_baseline(rot, root, off, parents, sample_frames):
    """Time the NumPy port of the reference fk on a bounded sample; also return its outputs."""
    import numpy as np

    from oracle import numpy_ref as nr

    n = min(sample_frames, rot.shape[0])
    r, g = rot[:n], root[:n]
    nr.fk(r[:1000], g[:1000], off, parents)  # warm NumPy / page in
    c0, t0 = os.times(), time.perf_counter()
    pos, rm = nr.fk_chunked(r, g, off, parents, chunk=1 << 17)
    t1, c1 = time.perf_counter(), os.times()
    wall = t1 - t0
    cpu = (c1.user - c0.user) + (c1.system - c0.system)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    info = {
        "value": n / wall,
        "unit": "frames/s",
        "cores": max(1, int(round(cpu / wall))),
        "kind": "port",
        "sample": f"{n} frames x {rot.shape[1]} joints, oracle/numpy_ref.fk_chunked (f64 [F,J,4,4] scratch, "
                  f"per-joint batched matmul, chunks of 2^17), {wall:.1f} s wall",
        "host_cpus_visible": len(os.sched_getaffinity(0)),
        "os_cpu_count": os.cpu_count(),
        "blas_threads": _blas_threads(),
        "cpu_model": model,
    }
    return info, pos, rm


def secondary_configs(torch, _lib, syn, dev, rot, root, off, parents, sptr):
    """BASELINE.json configs[2] (dual-quaternion round trip, 22 joints, 2^20 frames) and configs[3] (fused
    ortho6d.to_quat -> fk, 52 joints, 2^18 frames): same device-resident method as the headline (150 untimed
    launches, then 100 timed back to back between two HIP events).  Reported, never folded into `value`."""
    import numpy as np

    F, J = rot.shape[0], rot.shape[1]
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    pp = parents.ctypes.data_as(C.c_void_p)
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _lib.call("pm_event_create", C.byref(e))

    def timed(fn, n=100):
        for _ in range(150):
            fn()
        # (a window is the kernel's time only while the launching thread stays ahead of the device: one in which enqueueing took more than half
        # of the device time -- the thread was descheduled, DESIGN section 6 -- is repeated, up to three times; secondary numbers only, the
        # headline region is timed exactly once)
        best = float("inf")
        for _attempt in range(3):
            t0 = time.perf_counter()
            _lib.call("pm_event_record", ev[0], sptr)
            for _ in range(n):
                fn()
            _lib.call("pm_event_record", ev[1], sptr)
            t_enq = (time.perf_counter() - t0) * 1e3
            ms = C.c_float()
            _lib.call("pm_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
            best = min(best, ms.value / n)
            if t_enq < 0.5 * ms.value:
                break
        return best

    out = {}
    rotn = rot / rot.norm(dim=-1, keepdim=True)
    # configs[1], second half: standalone quat.to_matrix on the normalised [F * 22, 4] (52 B per quaternion)
    m_out = torch.empty((F * J, 3, 3), device=dev)
    t_m = timed(lambda: _lib.call("pm_quat_to_matrix_f32", p(rotn), F * J, p(m_out), sptr))
    out["quat_to_matrix"] = {"quaternions": F * J, "ms": t_m, "quats_per_s": F * J / (t_m * 1e-3),
                             "hbm_frac": F * J * 52 / (t_m * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    del m_out
    # the headline kernel on centimetre-scale data (what mocap BVH files hold: bones ~30, roots ~200): those tiles take float64
    # local rotations and the fixed-point translation chain
    off_cm = off * 100.0
    root_cm = root * 100.0
    pos_c = torch.empty((F, J, 3), device=dev)
    rm_c = torch.empty((F, J, 3, 3), device=dev)
    t_cm = timed(lambda: _lib.call("pm_fk_f32", p(rot), p(root_cm), p(off_cm), 0, pp, F, J, p(pos_c), p(rm_c), sptr))
    out["fk_centimetre_scale_J22"] = {"frames": F, "ms": t_cm, "frames_per_s": F / (t_cm * 1e-3),
                                      "hbm_frac": F * (64 * J + 12) / (t_cm * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel": _lib.last_kernel_name()}
    del pos_c, rm_c, off_cm, root_cm
    dq = torch.empty((F, J, 8), device=dev)
    tr = torch.empty((F, J, 3), device=dev)
    qo = torch.empty((F, J, 4), device=dev)
    t_to = timed(lambda: _lib.call("pm_to_root_dq_f32", p(rotn), p(root), pp, p(off), F, J, p(dq), sptr))
    t_from = timed(lambda: _lib.call("pm_from_root_dq_f32", p(dq), pp, F, J, p(tr), p(qo), sptr))
    b_to, b_from = F * (48 * J + 12), F * 60 * J
    out["dual_quat_round_trip_J22"] = {
        "frames": F, "to_root_ms": t_to, "from_root_ms": t_from, "frames_per_s": F / ((t_to + t_from) * 1e-3),
        "hbm_frac": (b_to + b_from) / ((t_to + t_from) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "round_trip_max_abs_err": {"rot": float((qo - rotn).abs().max()), "offsets": float((tr[:, 1:] - off[1:]).abs().max()),
                                   "root": float((tr[:, 0] - root).abs().max())},
    }
    del dq, tr, qo, rotn
    # configs[3] on THREE independently allocated sets of arrays: a launch's time follows the placement of its arrays (profiles/r06_levels.txt:
    # the same kernel reads 150 us on one set and 167 us on the next, visit after visit), so `ms` is the mean over the placements and
    # `ms_by_placement` says what each one read
    F4, par52 = 1 << 18, syn.PARENTS_52
    off4 = torch.from_numpy(syn.make_offsets(52, np.random.default_rng(4), 0.15)).to(dev)
    pp4 = par52.ctypes.data_as(C.c_void_p)
    sets4 = []
    for b in range(3):
        sets4.append({"x": torch.randn((F4, 52, 3, 2), device=dev), "root": torch.rand((F4, 3), device=dev) * 4 - 2,
                      "pos": torch.empty((F4, 52, 3), device=dev), "rm": torch.empty((F4, 52, 3, 3), device=dev),
                      "q": torch.empty((F4, 52, 4), device=dev), "big": torch.empty(F4 * 52 * 12, device=dev)})

    def over_sets(make):
        ts = [timed(make(d)) for d in sets4]
        return sum(ts) / len(ts), ts

    t4, t4s = over_sets(lambda d: (lambda: _lib.call("pm_fk_from_ortho6d_f32", p(d["x"]), p(d["root"]), p(off4), 0, pp4, F4, 52, C.c_float(0.0),
                                                       p(d["pos"]), p(d["rm"]), None, sptr)))
    out["fused_ortho6d_fk_J52"] = {"frames": F4, "ms": t4, "ms_by_placement": t4s, "frames_per_s": F4 / (t4 * 1e-3),
                                   "hbm_frac": F4 * (72 * 52 + 12) / (t4 * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    t4q, t4qs = over_sets(lambda d: (lambda: _lib.call("pm_fk_from_ortho6d_f32", p(d["x"]), p(d["root"]), p(off4), 0, pp4, F4, 52, C.c_float(0.0),
                                                         p(d["pos"]), p(d["rm"]), p(d["q"]), sptr)))
    out["fused_ortho6d_fk_J52_quat_out"] = {"frames": F4, "ms": t4q, "ms_by_placement": t4qs, "frames_per_s": F4 / (t4q * 1e-3),
                                            "hbm_frac": F4 * (88 * 52 + 12) / (t4q * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    t52, t52s = over_sets(lambda d: (lambda: _lib.call("pm_fk_f32", p(d["q"]), p(d["root"]), p(off4), 0, pp4, F4, 52, p(d["pos"]), p(d["rm"]), sptr)))
    # the copy kernel of fk's own shape at this joint count on the same arrays (it follows the placement too, half as much)
    t52c, t52cs = over_sets(lambda d: (lambda: _lib.call("pm_stream_ceiling_f32", p(d["q"]), p(d["big"]), F4, 4 * 52, 12 * 52, sptr)))
    out["fk_J52"] = {"frames": F4, "ms": t52, "ms_by_placement": t52s, "frames_per_s": F4 / (t52 * 1e-3),
                     "hbm_frac": F4 * (64 * 52 + 12) / (t52 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "copy_ceiling_ms": t52c, "copy_ceiling_ms_by_placement": t52cs,
                     "copy_ceiling_hbm_frac": F4 * 64 * 52 / (t52c * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "kernel_over_copy_ceiling": t52 / t52c}
    x, root4, pos4, rm4, q4 = (sets4[0][k] for k in ("x", "root", "pos", "rm", "q"))
    for d in sets4:
        del d["big"]
    # from_root_positions (positions -> local rotations) on the fk output just made: SMPL-H's 52-joint table AS STORED
    # (level order -- round 3: 30.5 % on the two-chain tile kernel) on the operation-driven lane-per-frame kernel, 28 J B per frame
    ik4 = torch.empty((F4, 52, 4), device=dev)
    pos4 -= pos4[:, :1].clone()
    t_ik = timed(lambda: _lib.call("pm_from_root_positions_f32", p(pos4), pp4, p(off4), F4, 52, p(ik4), sptr))
    out["from_root_positions_J52_level_order"] = {"frames": F4, "ms": t_ik, "frames_per_s": F4 / (t_ik * 1e-3),
                                                   "hbm_frac": F4 * 28 * 52 / (t_ik * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel": _lib.last_kernel_name()}
    # to_root_dual_quat and mirror on the same 52-joint tree (SMPL-H, unit quaternions from the conversion above, metre-scale bones): the step-list kernels of
    # round 6 (dqwide.hip / mirror_wide_kernel: 16 / fpw joints of a frame a step from a host-made list held in registers), a slice checked against the oracle
    dq4 = torch.empty((F4, 52, 8), device=dev)
    t_dq4 = timed(lambda: _lib.call("pm_to_root_dq_f32", p(q4), p(root4), pp4, p(off4), F4, 52, p(dq4), sptr))
    k_dq4 = _lib.last_kernel_name()
    from oracle import numpy_ref as nr_

    sl4 = slice(F4 // 2, F4 // 2 + 256)
    want4 = nr_.to_root_dual_quat(q4[sl4].cpu().numpy().astype(np.float64), root4[sl4].cpu().numpy().astype(np.float64), syn.PARENTS_52, off4.cpu().numpy().astype(np.float64))
    err_dq4 = float(np.abs(dq4[sl4].cpu().numpy() - want4).max())
    assert err_dq4 <= 1e-5, err_dq4
    mir4 = torch.empty((F4, 52, 4), device=dev)
    t_mir4 = timed(lambda: _lib.call("pm_mirror_rotations_f32", p(q4), pp4, None, 0, F4, 52, p(mir4), sptr))
    out["to_root_dual_quat_J52"] = {"frames": F4, "ms": t_dq4, "hbm_frac": F4 * (48 * 52 + 12) / (t_dq4 * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel": k_dq4,
                                    "max_abs_err_vs_oracle_256_frames": err_dq4}
    out["mirror_J52"] = {"frames": F4, "ms": t_mir4, "hbm_frac": F4 * 32 * 52 / (t_mir4 * 1e-3) / 1e9 / HBM_PEAK_GBPS, "kernel": _lib.last_kernel_name()}
    del dq4, mir4
    del x, root4, off4, pos4, rm4, q4, ik4, sets4
    # ... and on a WIDE 250-joint tree (parents[j] uniform in [0, j): 10 levels), where rounds 1-5 read 39 % / 37 % (sixteen chains a frame in 25 KB of LDS a wave / the
    # chain scheduler's last joint count)
    F6, J6 = 1 << 17, 250
    par6 = syn.random_parents(J6, np.random.default_rng(J6)).astype(np.int32)
    pp6 = par6.ctypes.data_as(C.c_void_p)
    rot6 = torch.randn((F6, J6, 4), device=dev)
    rot6 /= rot6.norm(dim=-1, keepdim=True)
    root6 = torch.rand((F6, 3), device=dev) * 4 - 2
    off6 = torch.randn((J6, 3), device=dev) * 0.1
    off6[0] = 0
    dq6 = torch.empty((F6, J6, 8), device=dev)
    t6 = timed(lambda: _lib.call("pm_to_root_dq_f32", p(rot6), p(root6), pp6, p(off6), F6, J6, p(dq6), sptr), n=40)
    k6 = _lib.last_kernel_name()
    want6 = nr_.to_root_dual_quat(rot6[:64].cpu().numpy().astype(np.float64), root6[:64].cpu().numpy().astype(np.float64), par6, off6.cpu().numpy().astype(np.float64))
    err6 = float(np.abs(dq6[:64].cpu().numpy() - want6).max())
    assert err6 <= 1e-5, err6
    mir6 = torch.empty((F6, J6, 4), device=dev)
    t6m = timed(lambda: _lib.call("pm_mirror_rotations_f32", p(rot6), pp6, None, 0, F6, J6, p(mir6), sptr), n=40)
    out["wide_random_tree_J250"] = {"frames": F6, "to_root_dual_quat_ms": t6, "to_root_dual_quat_hbm_frac": F6 * (48 * J6 + 12) / (t6 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "to_root_dual_quat_kernel": k6, "to_root_max_abs_err_vs_oracle_64_frames": err6,
                                    "mirror_ms": t6m, "mirror_hbm_frac": F6 * 32 * J6 / (t6m * 1e-3) / 1e9 / HBM_PEAK_GBPS, "mirror_kernel": _lib.last_kernel_name()}
    del rot6, root6, off6, dq6, mir6
    # a LONG, chain-like skeleton (128 joints: one chain, a second one off the root, a third off joint 32; 2^18 frames): what the
    # tile kernels are worst at (round 2: to_root_dual_quat 31 %, fk 46 %).  to_root_dual_quat: the lane-per-frame kernel of deep.hip.
    F5, J5 = 1 << 18, 128
    par5 = np.maximum(np.arange(J5) - 1, 0).astype(np.int32)
    par5[J5 // 2] = 0
    par5[3 * J5 // 4] = J5 // 4
    pp5 = par5.ctypes.data_as(C.c_void_p)
    rot5 = torch.randn((F5, J5, 4), device=dev)
    rot5 /= rot5.norm(dim=-1, keepdim=True)
    root5 = torch.rand((F5, 3), device=dev) * 4 - 2
    off5 = torch.randn((J5, 3), device=dev) * 0.15
    off5[0] = 0
    dq5 = torch.empty((F5, J5, 8), device=dev)
    t5 = timed(lambda: _lib.call("pm_to_root_dq_f32", p(rot5), p(root5), pp5, p(off5), F5, J5, p(dq5), sptr), n=40)
    k5 = _lib.last_kernel_name()
    del dq5
    pos5 = torch.empty((F5, J5, 3), device=dev)
    rm5 = torch.empty((F5, J5, 3, 3), device=dev)
    t5f = timed(lambda: _lib.call("pm_fk_f32", p(rot5), p(root5), p(off5), 0, pp5, F5, J5, p(pos5), p(rm5), sptr), n=40)
    out["long_chain_like_skeleton_J128"] = {
        "frames": F5, "to_root_dual_quat_ms": t5, "to_root_dual_quat_hbm_frac": F5 * (48 * J5 + 12) / (t5 * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "to_root_dual_quat_kernel": k5, "fk_ms": t5f, "fk_hbm_frac": F5 * (64 * J5 + 12) / (t5f * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "fk_kernel": _lib.last_kernel_name()}
    return out


def config1_clip(torch, _lib, syn, dev, sptr):
    """BASELINE.json configs[0] on the GPU side: a 1000-frame 22-joint BVH clip (synthetic.write_synthetic_bvh: the reference's README
    joint names, metre-scale offsets) -> BVH.load -> get_data (from_euler + unroll + normalize, one launch) -> fk, through the NumPy
    door (host arrays in and out: two trips over PCIe) and through the torch door (device-resident tensors), median per call of 200
    calls in five batches, next to the kernel's own time (HIP events around 200 back-to-back raw pm_fk_f32 launches) and to the NumPy
    restatement of the reference on the SAME arrays on this box's host cores.  At this size the kernel is a few microseconds and the
    front door is the cost: this is the honest answer to "how much faster is a real clip"."""
    import tempfile

    import numpy as np

    import pymotion_amd.ops.skeleton as sk
    import pymotion_amd.ops.skeleton_torch as skt
    from oracle import numpy_ref as nr
    from pymotion_amd.io.bvh import BVH

    def per_call_us(fn, n=200, warm=20):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        per = []
        for _ in range(5):  # median of five batches: a process sees the odd 10-40 ms host stall
            t0 = time.perf_counter()
            for _ in range(n // 5):
                fn()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / (n // 5) * 1e6)
        return sorted(per)[2]

    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "clip1000.bvh")
        syn.write_synthetic_bvh(path, n_frames=1000, seed=7)
        b = BVH()
        t0 = time.perf_counter()
        b.load(path)
        t_load = time.perf_counter() - t0
    t_get = per_call_us(b.get_data, n=50, warm=5)
    rots, pos, parents, offsets, _, _ = b.get_data()
    root = np.ascontiguousarray(pos[:, 0, :])
    F, J = rots.shape[0], rots.shape[1]
    # CPU: the cost-equivalent NumPy restatement of the reference on these arrays (float64 as get_data returns them)
    nr.fk(rots, root, offsets, parents)
    cpu = []
    for _ in range(20):
        t0 = time.perf_counter()
        p_cpu, r_cpu = nr.fk(rots, root, offsets, parents)
        cpu.append(time.perf_counter() - t0)
    cpu_us = min(cpu) * 1e6
    # NumPy door
    t_np = per_call_us(lambda: sk.fk(rots, root, offsets, parents))
    p_np, r_np = sk.fk(rots, root, offsets, parents)
    # torch door, device-resident
    tr, tg, to = (torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev) for x in (rots, root, offsets))
    tp = torch.from_numpy(np.asarray(parents))
    with torch.no_grad():
        t_t = per_call_us(lambda: skt.fk(tr, tg, to, tp))
        p_t, r_t = skt.fk(tr, tg, to, tp)
    # the kernel alone: raw C-ABI launches back to back between two HIP events
    pp = np.ascontiguousarray(parents, dtype=np.int32)
    po, ro = torch.empty((F, J, 3), device=dev), torch.empty((F, J, 3, 3), device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    raw = lambda: _lib.call("pm_fk_f32", P(tr), P(tg), P(to), 0, pp.ctypes.data_as(C.c_void_p), F, J, P(po), P(ro), sptr)  # noqa: E731
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _lib.call("pm_event_create", C.byref(e))
    for _ in range(50):
        raw()
    _lib.call("pm_event_record", ev[0], sptr)
    for _ in range(200):
        raw()
    _lib.call("pm_event_record", ev[1], sptr)
    torch.cuda.synchronize()
    ms = C.c_float()
    _lib.call("pm_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
    t_raw_host = per_call_us(raw)
    err = max(float(np.abs(p_np - p_cpu).max()), float(np.abs(r_np - r_cpu).max()),
              float(np.abs(p_t.cpu().numpy() - p_cpu).max()), float(np.abs(r_t.cpu().numpy() - r_cpu).max()))
    return {
        "clip": "%d frames x %d joints, synthetic BVH (pymotion_amd.synthetic.write_synthetic_bvh), loaded by pymotion_amd.io.bvh" % (F, J),
        "bvh_load_ms": t_load * 1e3, "get_data_us": t_get,
        "fk_numpy_port_on_host_us": cpu_us, "fk_numpy_port_frames_per_s": F / (cpu_us * 1e-6),
        "fk_numpy_door_us": t_np, "fk_numpy_door_frames_per_s": F / (t_np * 1e-6),
        "fk_torch_door_us": t_t, "fk_torch_door_frames_per_s": F / (t_t * 1e-6),
        "fk_raw_abi_call_us": t_raw_host, "fk_kernel_us": ms.value / 200 * 1e3, "kernel": _lib.last_kernel_name(),
        "speedup_over_numpy_port": {"numpy_door": cpu_us / t_np, "torch_door": cpu_us / t_t},
        "max_abs_err_vs_numpy_port": err,
        "method": "median per call of 200 calls in five batches (device synchronised per batch); kernel: HIP events around 200 launches",
    }


def stream_ceilings(torch, _lib, dev, rot, rm, F, J, sptr):
    """What this chip moves with NO arithmetic, measured in the same process right after the timed region (never inside it):
    a copy with fk's traffic shape through fk's own tiling (`pm_stream_ceiling_f32`: 16 J B read, 48 J B written per frame;
    the 12 B root position is left out), LDS-free grid-stride streams for a pure read and the 1:1 / 1:2 / 1:3 read:write mixes, and
    the store probe's chunk-per-wave streams (round 4: the grid-stride pure write of round 3 measured its own pattern, 4.5 TB/s,
    where a chunk per wave writes 6.5 TB/s).  fk_kernel_ms / copy_ceiling_ms says how far the kernel is from a plain copy through
    its tiling; DESIGN.md section 6 says why neither is the chip's limit for a 1:3 mix."""
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _lib.call("pm_event_create", C.byref(e))

    def timed(fn, n=60, warm=60):
        for _ in range(warm):
            fn()
        _lib.call("pm_event_record", ev[0], sptr)
        for _ in range(n):
            fn()
        _lib.call("pm_event_record", ev[1], sptr)
        ms = C.c_float()
        _lib.call("pm_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
        return ms.value / n

    src, dst = rot.view(-1), rm.view(-1)
    n4 = F * J                      # dwordx4 in `rot`; `rm` holds 9/4 as many (36 J B per frame), enough for ratios up to 2
    big = torch.empty(n4 * 12, device=dev)  # 48 J B per frame: fk's output volume
    out = {}
    t = timed(lambda: _lib.call("pm_stream_ceiling_f32", p(src), p(big), F, 4 * J, 12 * J, sptr))
    out["copy_ceiling_ms"] = t
    out["copy_ceiling_GBps"] = F * 64 * J / (t * 1e-3) / 1e9
    rates = {}
    t = timed(lambda: _lib.call("pm_stream_plain_f32", p(big), p(dst), n4 * 3, 0, 8192, sptr))
    rates["pure_read"] = {"bytes": n4 * 48, "ms": t, "GBps": n4 * 48 / (t * 1e-3) / 1e9}
    t = timed(lambda: _lib.call("pm_stream_plain_f32", p(src), p(big), n4 * 3, -1, 8192, sptr))
    rates["pure_write_grid_stride"] = {"bytes": n4 * 48, "ms": t, "GBps": n4 * 48 / (t * 1e-3) / 1e9,
                                       "note": "round 3's pure-write figure: 8192 persistent 256-thread workgroups, each store instruction of a wave 1 KiB away from "
                                               "its last -- the PATTERN's rate, not the chip's (profiles/r04_store_patterns.txt)"}

    # round 4: what the store path does when every wave writes ONE contiguous chunk (pm_store_probe_f32: burst KiB per wave, reads in front,
    # nt stores, one contiguous range of chunks per XCD) -- pure writes, and the 1 : 3 mix at fk's chunk size and waves per CU
    def probe(burst, rd4, lds):
        cfg = (C.c_int32 * 12)(burst, rd4, 1, 1, 1, 64, 0, 0, 0, 0, 0, lds)
        nw = n4 * 3  # dwordx4 written = fk's outputs
        tt = timed(lambda: _lib.call("pm_store_probe_f32", p(src) if rd4 else None, p(big), nw, cfg, sptr))
        chunks = nw // (burst * 64)
        nbytes = chunks * (burst + rd4) * 1024
        return {"bytes": nbytes, "ms": tt, "GBps": nbytes / (tt * 1e-3) / 1e9}

    rates["pure_write"] = dict(probe(4, 0, 0), note="4 KiB contiguous per wave, one chunk per wave")
    rates["read_write_1_3_chunks_4k_12k_9_waves_per_cu"] = dict(probe(12, 4, 17 * 1024), note="fk's mix, chunk size and residency, no arithmetic, no LDS traffic")
    rates["read_write_1_4_chunks_2k_8k"] = dict(probe(8, 2, 0), note="the best mixed stream found")
    t = timed(lambda: _lib.call("pm_stream_plain_f32", p(src), p(big), n4, 1, 8192, sptr))
    rates["read_write_1_1"] = {"bytes": n4 * 32, "ms": t, "GBps": n4 * 32 / (t * 1e-3) / 1e9}
    t = timed(lambda: _lib.call("pm_stream_plain_f32", p(src), p(big), n4, 2, 8192, sptr))
    rates["read_write_1_2"] = {"bytes": n4 * 48, "ms": t, "GBps": n4 * 48 / (t * 1e-3) / 1e9}
    t = timed(lambda: _lib.call("pm_stream_plain_f32", p(src), p(big), n4, 3, 8192, sptr))
    rates["read_write_1_3"] = {"bytes": n4 * 64, "ms": t, "GBps": n4 * 64 / (t * 1e-3) / 1e9}
    out["stream_rates"] = rates
    del big
    return out


def cpu_baseline_extras(np, syn, threads):
    """Two more CPU lines: config 1 -- the NumPy restatement on a cache-resident 1000-frame clip, where
    Python overhead is amortised least -- and the oracle's scalar C fk (`oracle/pm_oracle.c`, float64 like the reference)
    run on every visible core (frame blocks on a thread pool; ctypes releases the GIL), which answers "is the GPU number
    just Python being slow"."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import c_oracle as co
    from oracle import numpy_ref as nr

    out = {}
    rot, root, off, parents = syn.fk_workload(1000, seed=1)
    nr.fk(rot, root, off, parents)
    best = 1e9
    for _ in range(20):
        t0 = time.perf_counter()
        nr.fk(rot, root, off, parents)
        best = min(best, time.perf_counter() - t0)
    out["config1_numpy_1000_frames"] = {"value": 1000 / best, "unit": "frames/s", "ms": best * 1e3, "kind": "port",
                                        "sample": "1000 frames x 22 joints, oracle/numpy_ref.fk, best of 20"}
    F = 1 << 21
    rot, root, off, parents = syn.fk_workload(F, seed=2)
    r64, g64, o64 = rot.astype(np.float64), root.astype(np.float64), off.astype(np.float64)
    blk = 1 << 11
    spans = [(i, min(i + blk, F)) for i in range(0, F, blk)]
    co.fk(r64[:blk], g64[:blk], o64, parents)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda se: co.fk(r64[se[0]:se[1]], g64[se[0]:se[1]], o64, parents), spans[:threads]))  # threads up
        t0 = time.perf_counter()
        list(ex.map(lambda se: co.fk(r64[se[0]:se[1]], g64[se[0]:se[1]], o64, parents), spans))
        wall = time.perf_counter() - t0
    out["c_oracle_all_cores"] = {"value": F / wall, "unit": "frames/s", "cores": threads, "kind": "port",
                                 "sample": "%d frames x 22 joints, oracle/pm_oracle.c fk_f64 (scalar C, one frame at a time), %d-frame blocks on "
                                           "%d threads, %.2f s wall" % (F, blk, threads, wall)}
    return out


def one_gpu_same_total(torch, _lib, syn, np, dev, F, J, parents, a):
    """all F frames of the N-GPU workload on this ONE GPU: prewarm, W warmup steps, K timed steps between two HIP events"""
    gen = torch.Generator(device=dev)
    gen.manual_seed(a.seed * 1000 + 999)
    rot = torch.randn((F, J, 4), generator=gen, device=dev, dtype=torch.float32)
    root = torch.rand((F, 3), generator=gen, device=dev, dtype=torch.float32) * 4 - 2
    off = torch.from_numpy(syn.make_offsets(J, np.random.default_rng(a.seed), 0.3)).to(dev)
    pos = torch.empty((F, J, 3), device=dev, dtype=torch.float32)
    rm = torch.empty((F, J, 3, 3), device=dev, dtype=torch.float32)
    sptr = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    pp = parents.ctypes.data_as(C.c_void_p)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    step = lambda: _lib.call("pm_fk_f32", P(rot), P(root), P(off), 0, pp, F, J, P(pos), P(rm), sptr)  # noqa: E731
    tp = time.perf_counter()
    while (time.perf_counter() - tp) * 1e3 < a.prewarm_ms:
        for _ in range(4):
            step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    ev = [C.c_void_p(), C.c_void_p()]
    for e in ev:
        _lib.call("pm_event_create", C.byref(e))
    steps = max(1, min(a.steps, 50))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.call("pm_event_record", ev[0], sptr)
    for _ in range(steps):
        step()
    _lib.call("pm_event_record", ev[1], sptr)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = C.c_float()
    _lib.call("pm_event_elapsed_ms", ev[0], ev[1], C.byref(ms))
    out = {"frames": F, "steps": steps, "ms_per_step": wall / steps * 1e3, "frames_per_s": F * steps / wall,
           "kernel_ms": ms.value / steps, "hbm_frac": F * (64 * J + 12) / (ms.value / steps * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "kernel": _lib.last_kernel_name(),
           "note": "the whole N-GPU workload on rank 0's GPU alone, measured before the sharded run: N-GPU value / this = the speed-up on the SAME problem"}
    del rot, root, pos, rm
    return out


def gather_report(torch, dist, a, world, rank, F, J, pos, rm, barrier, compute_s, cdev, shared):
    """(ii) reassembling (pos, rotmats) on every GPU -- the ONE collective a caller may ask for -- timed for
    both implementations in pymotion_amd.parallel, with the achieved xGMI receive rate per GPU against the (W-1)-link
    roofline, and (iii) compute + gather combined.  After the timed region, never part of `value`, never fatal."""
    from pymotion_amd.parallel import GATHER_METHODS, all_gather_frames

    shard_bytes = F * J * 48  # pos 12 J + rotmats 36 J bytes per frame
    recv_bytes = shard_bytes * (world - 1)
    peak = XGMI_LINK_GBPS_PER_DIRECTION * (world - 1)
    rep = {"shard_GB": shard_bytes / 1e9, "received_GB_per_gpu": recv_bytes / 1e9,
           "xgmi_peak_GBps_per_gpu": peak,
           "xgmi_peak_note": "%d links x %.1f GB/s per direction (153.6 GB/s per link counts both directions)"
                             % (world - 1, XGMI_LINK_GBPS_PER_DIRECTION),
           "compute_only_ms": compute_s * 1e3, "methods": {}}
    src = (pos.cpu(), rm.cpu()) if shared else (pos, rm)  # gloo has no device all-gather: the rehearsal stages through the host
    for method in GATHER_METHODS:
        try:
            times = []
            for it in range(3):  # first pass pays RCCL's per-peer connection set-up; report the best of the rest
                torch.cuda.synchronize()
                barrier()
                g0 = time.perf_counter()
                gp = all_gather_frames(src[0], F * world, method=method)
                gr = all_gather_frames(src[1], F * world, method=method)
                torch.cuda.synchronize()
                barrier()
                times.append(time.perf_counter() - g0)
                if it == 0:  # the reassembled arrays must hold every rank's shard in rank order: check our own block
                    ok = bool(torch.equal(gp[rank * F:(rank + 1) * F], src[0])) and bool(torch.equal(gr[rank * F:(rank + 1) * F], src[1]))
                    okt = torch.tensor([0.0 if ok else 1.0], device=cdev, dtype=torch.float64)
                    dist.all_reduce(okt, op=dist.ReduceOp.MAX)
                    ok = float(okt[0]) == 0.0
                del gp, gr
            tt = torch.tensor([min(times[1:])], device=cdev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sec = float(tt[0])
            rep["methods"][method] = {"ms": sec * 1e3, "first_call_ms": times[0] * 1e3, "xgmi_recv_GBps_per_gpu": recv_bytes / sec / 1e9,
                                      "frac_of_xgmi_peak": (recv_bytes / sec / 1e9 / peak) if peak else None, "own_block_intact": ok}
        except Exception as exc:  # noqa: BLE001
            rep["methods"][method] = {"error": repr(exc)[:300]}
    good = {k: v for k, v in rep["methods"].items() if "ms" in v}
    if good:
        best = min(good, key=lambda k: good[k]["ms"])
        rep["selected"] = best
        rep["ms"] = good[best]["ms"]
        comb = compute_s + good[best]["ms"] * 1e-3
        rep["combined_compute_plus_gather"] = {"ms": comb * 1e3, "frames_per_s": F * world / comb,
                                               "note": "one fk pass + one reassembly on every GPU; `value` is compute-only"}
    return rep


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn
    import pymotion_amd.ops.skeleton_torch as skt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1 and "RANK" not in os.environ:
            self_launch(a)  # does not return
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    shared = a.dry_run_shared_gpu
    if not shared and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPU(s); --dry-run-shared-gpu rehearses the "
                 "launch path on one device")
    dev_index = 0 if shared else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or os.environ.get("PM_BENCH_FORCE_DIST") == "1"
    if world > 1 and not a.rccl_debug_file and not shared:
        # every N > 1 run leaves RCCL's own record of what it did (topology, channels, algorithm / protocol per collective) under profiles/:
        # the reassembly has never run on 8 GPUs anywhere but the driver's box.  Only if the directory takes a file ("none" switches it off;
        # a log that cannot open its file would fall back to stdout, which carries the one JSON line).
        try:
            pdir = os.path.join(ROOT, "profiles")
            probe_path = os.path.join(pdir, ".rccl_probe.%d" % os.getpid())
            with open(probe_path, "w") as fh:
                fh.write("x")
            os.remove(probe_path)
            a.rccl_debug_file = os.path.join(pdir, "rccl_debug_N%d" % world)
        except OSError:
            a.rccl_debug_file = ""
    if a.rccl_debug_file == "none":
        a.rccl_debug_file = ""
    if a.rccl_debug_file:
        os.environ["NCCL_DEBUG"] = "INFO"
        os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,COLL,P2P"
        os.environ["NCCL_DEBUG_FILE"] = a.rccl_debug_file + ".%h.%p"
    if use_dist:  # one rank per GPU over RCCL (backend "nccl" on ROCm); also taken by `torchrun --nproc-per-node 1`
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # (RCCL prints a version banner on STDOUT when it initialises; stdout carries the one JSON line and nothing else)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if shared:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            dist.barrier()  # pay RCCL's lazy communicator set-up now, not inside the barrier that opens the timed region
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    cdev = torch.device("cpu") if shared else dev  # where the tiny control-plane tensors (timings, errors) live

    J = a.joints
    parents = syn.PARENTS_22 if J == 22 else syn.PARENTS_52
    CONFIG5_FRAMES = 1 << 24
    explicit_frames = a.frames_per_gpu > 0
    F = a.frames_per_gpu if explicit_frames else ((1 << 20) if world == 1 else CONFIG5_FRAMES // world)
    if (F, J, world) == (1 << 20, 22, 1):
        workload, scaling = "BASELINE.json configs[1]", "weak"
    elif J == 22 and F * world == CONFIG5_FRAMES and F == CONFIG5_FRAMES // world:
        workload = ("all 16 777 216 frames of BASELINE.json configs[4] on one GPU" if world == 1 else
                    "BASELINE.json configs[4]: 16 777 216 frames sharded over %d GPUs" % world)
        scaling = "weak" if explicit_frames else "strong"  # --frames-per-gpu fixes the work per GPU, the default fixes the total
    else:
        workload, scaling = "non-default size", "weak"
    # N > 1 on the default (strong-scaling) workload: the SAME total -- all 16 777 216 frames -- on ONE GPU first (rank 0, the others wait
    # at a barrier), same method as the timed region below, so that the N-GPU value is read against the same problem and not only against
    # the driver's N = 1 run of configs[1] (2^20 frames).  ~24 GB of device memory for a few seconds; reported, never part of `value`.
    one_gpu = None
    if world > 1 and not explicit_frames and not shared and J == 22:
        if rank == 0:
            try:
                one_gpu = one_gpu_same_total(torch, _lib, syn, np, dev, CONFIG5_FRAMES, J, parents, a)
            except Exception as exc:  # noqa: BLE001
                one_gpu = {"error": repr(exc)[:300]}
            torch.cuda.empty_cache()
        dist.barrier()
    # synthetic workload born on the device from (seed, rank): no host->device copy is ever timed.  B batches, each its own allocations
    # (see --batches); batch 0 is the one every check below reads
    NB = a.batches if a.batches > 0 else (3 if F <= (1 << 21) else 1)
    off_np = syn.make_offsets(J, np.random.default_rng(a.seed), 0.3 if J == 22 else 0.15)
    off = torch.from_numpy(off_np).to(dev)
    par_t = torch.from_numpy(parents)
    batches = []
    for b in range(NB):
        if b > 0:
            try:  # (a further batch is a nicety: a device that cannot hold it runs the line on the batches it has)
                free_b, _tot = torch.cuda.mem_get_info(dev)
                if free_b < 2 * F * (64 * J + 12) + (1 << 30):
                    NB = b
                    break
            except Exception:  # noqa: BLE001
                pass
        gen = torch.Generator(device=dev)
        gen.manual_seed(a.seed * 1000 + rank + 7919 * b)
        rot_b = torch.randn((F, J, 4), generator=gen, device=dev, dtype=torch.float32)
        root_b = torch.rand((F, 3), generator=gen, device=dev, dtype=torch.float32) * 4 - 2
        # steady-state launch path = what skeleton_torch.fk does after its tensor plumbing: one C-ABI call
        pos_b = torch.empty((F, J, 3), device=dev, dtype=torch.float32)
        rm_b = torch.empty((F, J, 3, 3), device=dev, dtype=torch.float32)
        batches.append((rot_b, root_b, pos_b, rm_b))
    rot, root, pos, rm = batches[0]
    stream = torch.cuda.current_stream(dev)
    sptr = C.c_void_p(stream.cuda_stream)
    pp = parents.ctypes.data_as(C.c_void_p)
    calls = [(C.c_void_p(r.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(p.data_ptr()), C.c_void_p(m.data_ptr())) for r, g, p, m in batches]
    offp = C.c_void_p(off.data_ptr())

    def step(i=0):
        r, g, p, m = calls[i % NB]
        _lib.call("pm_fk_f32", r, g, offp, 0, pp, F, J, p, m, sptr)

    # the public front door (tensor plumbing + the same kernel on a small batch) and the raw C-ABI call used
    # in the timed loop must agree to fp32 rounding
    with torch.no_grad():
        p2, r2 = skt.fk(rot[:4096], root[:4096], off, par_t)
    step()
    torch.cuda.synchronize()
    assert float((p2 - pos[:4096]).abs().max()) < 5e-6 and float((r2 - rm[:4096]).abs().max()) < 5e-6

    def barrier():
        if use_dist:
            dist.barrier()

    if a.prewarm_ms > 0:  # DVFS settling: untimed, not part of W or K
        tp = time.perf_counter()
        while (time.perf_counter() - tp) * 1e3 < a.prewarm_ms:
            for i in range(20):
                step(i)
            torch.cuda.synchronize()
    ev0, ev1 = C.c_void_p(), C.c_void_p()
    _lib.call("pm_event_create", C.byref(ev0))
    _lib.call("pm_event_create", C.byref(ev1))
    barrier()  # a warm one: an idle gap of >= 3 ms before the timed region would cost ~20 ms of re-ramp
    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    _lib.call("pm_event_record", ev0, sptr)
    for i in range(a.steps):
        step(i)
    _lib.call("pm_event_record", ev1, sptr)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    ms = C.c_float()
    _lib.call("pm_event_elapsed_ms", ev0, ev1, C.byref(ms))
    wall = t1 - t0
    kern_ms = ms.value / a.steps  # average launch-to-launch time of the fk kernel on its stream

    kernel_name = _lib.last_kernel_name()  # what pm_fk_f32 dispatched to, spelled like rocprofv3 prints it
    if use_dist:
        tt = torch.tensor([wall, kern_ms], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, kern_ms = float(tt[0]), float(tt[1])

    extra = {}
    if shared:
        extra["dry_run_shared_gpu"] = "launch-path rehearsal: every rank on cuda:0, gloo process group -- the numbers mean nothing"
    if use_dist:
        # every rank checks a slice of ITS shard against the CPU oracle (the checker, not the measured path)
        n_chk = min(F, a.oracle_slice_frames)
        sl = slice(F // 2, F // 2 + n_chk) if F >= 2 * n_chk else slice(0, n_chk)
        try:
            from oracle import c_oracle as co

            p_o, r_o = co.fk(rot[sl].cpu().numpy().astype(np.float64), root[sl].cpu().numpy().astype(np.float64),
                             off_np.astype(np.float64), parents)
            err = max(float(np.abs(pos[sl].cpu().numpy() - p_o).max()), float(np.abs(rm[sl].cpu().numpy() - r_o).max()))
        except Exception:  # noqa: BLE001  (checker unavailable on this box: report it, keep every rank in the collective)
            err = float("inf")
        et = torch.tensor([err], device=cdev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        extra["max_abs_err_vs_oracle_slice"] = {"value": float(et[0]), "frames_per_rank": n_chk}
    by_batch = None
    if NB > 1 and not shared:
        # outside the timed region: every batch on its own (same events, 30 launches each) -- the spread IS the placement lottery -- and a slice of
        # every other batch against the oracle (batch 0 is checked on all its frames below)
        by_batch = []
        for b in range(NB):
            for _ in range(5):
                step(b)
            _lib.call("pm_event_record", ev0, sptr)
            for _ in range(30):
                step(b)
            _lib.call("pm_event_record", ev1, sptr)
            _lib.call("pm_event_elapsed_ms", ev0, ev1, C.byref(ms))
            by_batch.append(ms.value / 30)
        if world == 1:
            try:
                from oracle import c_oracle as co

                worst = 0.0
                for b in range(1, NB):
                    rb, gb, pb, mb = batches[b]
                    sl = slice(F // 3, F // 3 + min(F - F // 3, 1 << 14))
                    p_o, r_o = co.fk(rb[sl].cpu().numpy().astype(np.float64), gb[sl].cpu().numpy().astype(np.float64), off_np.astype(np.float64), parents)
                    worst = max(worst, float(np.abs(pb[sl].cpu().numpy() - p_o).max()), float(np.abs(mb[sl].cpu().numpy() - r_o).max()))
                extra["max_abs_err_other_batches_vs_oracle_slice"] = worst
            except Exception as exc:  # noqa: BLE001
                extra["max_abs_err_other_batches_vs_oracle_slice"] = repr(exc)[:200]
    ceil = None
    if world == 1 and not a.no_secondary:
        ceil = stream_ceilings(torch, _lib, dev, rot, rm, F, J, sptr)
        step()  # `rm` served as a scratch destination above: restore the kernel's outputs for the checks below
        torch.cuda.synchronize()
    if world == 1 and not a.no_secondary and J == 22:
        extra["secondary"] = secondary_configs(torch, _lib, syn, dev, rot, root, off, parents, sptr)
        try:
            extra["secondary"]["config1_bvh_clip_1000_frames"] = config1_clip(torch, _lib, syn, dev, sptr)
        except Exception as exc:  # noqa: BLE001  (a secondary line: never costs the run its bench line)
            extra["secondary"]["config1_bvh_clip_1000_frames"] = {"error": repr(exc)[:300]}

    if rank == 0:
        bytes_per_frame = 64 * J + 12
        achieved = bytes_per_frame * F / (kern_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 + WRITE_SIZE,
        # corrections per MI355X_MICROARCH.md, calibrated on a known-byte copy kernel); same workload only.
        traffic, traffic_source = None, None
        try:
            latest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_fk_hbm_traffic.json"))[-1]
            prof = json.load(open(os.path.join(ROOT, "profiles", latest)))
            if prof["workload"] == {"frames": F, "joints": J}:
                traffic = prof["corrected_bytes_per_launch"]["total"]
                traffic_source = "profiles/%s (rocprofv3 --pmc passes of this same command, not re-measured in this run)" % latest
        except (OSError, IndexError, KeyError, ValueError):
            pass
        line = {
            "metric": "fk() frames/sec, %d-joint skeleton" % J,
            "value": F * world * a.steps / wall,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "prewarm_ms": a.prewarm_ms,
            "config": {"workload": "fk: %d frames x %d joints per GPU, fp32 (%s)" % (F, J, workload),
                       "frames_per_gpu": F, "frames_total": F * world, "joints": J, "sharding": "frames, no data-path collective",
                       "batches": NB},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes": bytes_per_frame * F,
                         "kernel": kernel_name, "kernel_ms": kern_ms,
                         "bytes_per_frame": bytes_per_frame},
        }
        try:  # which GPU this was: the timing level of a run follows the box as well as the placement (profiles/r06_levels.txt)
            pr = torch.cuda.get_device_properties(dev)
            line["config"]["gpu"] = {"name": pr.name, "uuid": str(getattr(pr, "uuid", ""))[-12:], "gcn_arch": getattr(pr, "gcnArchName", "")}
        except Exception:  # noqa: BLE001
            pass
        try:  # ... and its VBIOS (thirteen boxes looked at: five read the slow side on every placement, and it is not the VBIOS that separates them)
            import amdsmi

            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            vb = amdsmi.amdsmi_get_gpu_vbios_info(hs[dev.index if dev.index is not None and dev.index < len(hs) else 0])
            line["config"].setdefault("gpu", {})["vbios"] = (str(vb.get("part_number", "")) + " " + str(vb.get("version", ""))).strip()[:60]
        except Exception:  # noqa: BLE001
            pass
        if by_batch is not None:
            # (the steps of the timed region cycle through these batches: kernel_ms is their mean; a launch's time follows the placement of
            # its arrays -- profiles/r06_levels.txt)
            line["roofline"]["kernel_ms_by_batch"] = by_batch
        if one_gpu is not None:
            line["one_gpu_same_total"] = one_gpu
            if "frames_per_s" in one_gpu:
                line["one_gpu_same_total_frames_per_s"] = one_gpu["frames_per_s"]
                line["speedup_over_one_gpu_same_total"] = line["value"] / one_gpu["frames_per_s"]
        if ceil is not None:
            line["roofline"].update(ceil)
            line["roofline"]["kernel_over_copy_ceiling"] = kern_ms / ceil["copy_ceiling_ms"]
        line.update(extra)
    else:
        line = None
    if use_dist and (world > 1 or a.rccl_debug_file) and not a.no_gather:
        # The reassembly measurements come LAST and under a watchdog: they are the only part of this script that talks over
        # xGMI peer to peer, nothing measured above depends on them, and a collective that hangs on some node must cost the
        # run its `gather` object, not its bench line.
        import threading

        def bail():
            if rank == 0:
                line["gather"] = {"error": "reassembly measurements did not finish within %d s: abandoned" % a.gather_timeout_s}
                print(json.dumps(line), flush=True)
            os._exit(0)

        dog = threading.Timer(a.gather_timeout_s, bail)
        dog.daemon = True
        dog.start()
        g = gather_report(torch, dist, a, world, rank, F, J, pos, rm, barrier, wall / a.steps, cdev, shared)
        dog.cancel()
        if rank == 0:
            line["gather"] = g
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            rot_h, root_h = rot.cpu().numpy(), root.cpu().numpy()
            info, p_cpu, r_cpu = cpu_baseline(rot_h, root_h, off_np, parents, a.cpu_sample_frames)
            n = p_cpu.shape[0]
            line["cpu_baseline"] = info
            line["max_abs_err_vs_cpu_baseline"] = {
                "pos": float(np.abs(pos[:n].cpu().numpy() - p_cpu).max()),
                "rotmats": float(np.abs(rm[:n].cpu().numpy() - r_cpu).max()),
                "frames_checked": n,
            }
            del p_cpu, r_cpu
            try:
                line["cpu_baseline_more"] = cpu_baseline_extras(np, syn, len(os.sched_getaffinity(0)))
            except Exception as exc:  # noqa: BLE001  (extra lines only: never cost the run its bench line)
                line["cpu_baseline_more"] = {"error": repr(exc)[:200]}
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
