"""GPU: the bounds-checked debug build of the library (libpmhip_debug.so, -DPM_DEBUG: LDS accesses of the shared helpers
checked against the workgroup's allocation, device synchronisation + error check + violation read after EVERY launch --
SURVEY section 5).  Every skeleton kernel and a sample of the element-wise ones run through the raw C ABI with each global
buffer embedded between guard words; a call must return PM_OK (no HIP fault, no LDS violation), leave every guard word
intact (no out-of-bounds global store, tile-edge partial stores included) and give the production build's results (to
the last ulp or two: the checks move FMA contraction around).  Sizes are chosen to hit partial tiles, the pipelined multi-tile workgroups and both arithmetic levels of fk."""
import ctypes as C

import numpy as np
import pytest

from pymotion_amd import _lib
from pymotion_amd import synthetic as syn

pytestmark = pytest.mark.gpu
GUARD = 1024  # floats on each side


class Guarded:
    """device buffers with guard words around them"""

    def __init__(self, torch):
        self.torch = torch
        self.bufs = []

    def new(self, shape, fill=None, dtype=None):
        torch = self.torch
        dtype = dtype or torch.float32
        n = int(np.prod(shape))
        n_pad = (n + 3) // 4 * 4  # keep the payload 16-byte aligned like any allocator would
        big = torch.full((2 * GUARD + n_pad,), 12345.0 if dtype == torch.float32 else 77, dtype=dtype, device="cuda")
        view = big[GUARD:GUARD + n].view(shape)
        if fill is not None:
            view.copy_(fill)
        self.bufs.append((big, n))
        return view

    def check(self):
        for big, n in self.bufs:
            g = 12345.0 if big.dtype == self.torch.float32 else 77
            assert bool((big[:GUARD] == g).all()), "guard words BEFORE a buffer were overwritten"
            assert bool((big[GUARD + n:] == g).all()), "guard words AFTER a buffer were overwritten"


def _p(t):
    return C.c_void_p(t.data_ptr())


def _both(fn):
    """run fn(lib-variant-is-active) under a release build and the debug build; return both result lists.  The debug build keeps every
    kernel reachable at these sizes (common.hpp: lane_per_frame_pays is `true` there), so its partner is the tuning build with that
    threshold at 0 -- the production kernels under the same dispatch; below the long-skeleton kernels' joint counts that IS the
    production dispatch."""
    import os

    out = []
    old = os.environ.get("PM_LPF_MIN_JOINT_FRAMES")
    os.environ["PM_LPF_MIN_JOINT_FRAMES"] = "0"
    try:
        for name in ("tuning", "debug"):
            with _lib.variant(name):
                out.append(fn())
    finally:
        if old is None:
            del os.environ["PM_LPF_MIN_JOINT_FRAMES"]
        else:
            os.environ["PM_LPF_MIN_JOINT_FRAMES"] = old
    return out


def _same(torch, x, y):
    """the two builds run the same source, but the checks change inlining and with it where the compiler contracts a
    multiply-add: equal up to an ulp or two of the data's scale, NaNs in the same places"""
    if bool(torch.equal(x, y)):
        return True
    nan = x.isnan() & y.isnan()
    if not bool((x.isnan() == y.isnan()).all()):
        return False
    tol = 4e-7 * (1.0 + float(torch.nan_to_num(x).abs().max()))
    return bool((torch.nan_to_num(x - y).abs() <= tol).all() | nan.all())


CASES = [(22, 61, 0.3, 2.0), (22, 20 * 7 + 3, 30.0, 200.0), (52, 4 * 9 + 1, 0.15, 2.0), (52, 70_003, 30.0, 200.0), (24, 45, 0.3, 2.0),
         (64, 13, 0.3, 2.0), (100, 21, 0.2, 2.0), (130, 9, 0.2, 2.0), (3, 1000, 0.3, 2.0), (1, 65, 0.3, 2.0),
         # chain-like skeletons (negative J): to_root_dual_quat's lane-per-frame kernels (chunks of eight / the line-aligned ring), full and
         # partial tiles of 64 frames, the guard words right behind every frame row they store
         (-56, 64, 0.3, 2.0), (-57, 130, 0.3, 2.0), (-63, 65, 30.0, 200.0), (-66, 1, 0.3, 2.0), (-72, 193, 0.3, 2.0), (-129, 67, 0.2, 2.0), (-96, 64, 0.2, 2.0)]


@pytest.mark.parametrize("J,F,osc,rsc", CASES)
def test_skeleton_kernels_under_the_debug_build(J, F, osc, rsc):
    import torch

    chain_like = J < 0
    J = abs(J)
    rng = np.random.default_rng(J * 1000 + F)
    parents = {22: syn.PARENTS_22, 52: syn.PARENTS_52}.get(J)
    if chain_like:
        parents = np.maximum(np.arange(J) - 1, 0).astype(np.int32)
        parents[J // 2] = 0
        parents[3 * J // 4] = J // 4
    elif parents is None:
        parents = syn.random_parents(J, rng)
    pp = parents.ctypes.data_as(C.c_void_p)
    rot_h = rng.standard_normal((F, J, 4)).astype(np.float32)
    rot_h /= np.linalg.norm(rot_h, axis=-1, keepdims=True)
    off_h = rng.uniform(-osc, osc, (J, 3)).astype(np.float32)
    off_h[0] = 0
    root_h = rng.uniform(-rsc, rsc, (F, 3)).astype(np.float32)
    x6_h = rng.standard_normal((F, J, 3, 2)).astype(np.float32)

    def run():
        g = Guarded(torch)
        rot = g.new((F, J, 4), torch.from_numpy(rot_h))
        root = g.new((F, 3), torch.from_numpy(root_h))
        off = g.new((J, 3), torch.from_numpy(off_h))
        offf = g.new((F, J, 3), torch.from_numpy(np.tile(off_h, (F, 1, 1))))
        x6 = g.new((F, J, 3, 2), torch.from_numpy(x6_h))
        pos, rm, q = g.new((F, J, 3)), g.new((F, J, 3, 3)), g.new((F, J, 4))
        pos2, rm2 = g.new((F, J, 3)), g.new((F, J, 3, 3))
        dq, tr, qo, lo, mi, ik = g.new((F, J, 8)), g.new((F, J, 3)), g.new((F, J, 4)), g.new((F, J, 4)), g.new((F, J, 4)), g.new((F, J, 4))
        _lib.call("pm_fk_f32", _p(rot), _p(root), _p(off), 0, pp, F, J, _p(pos), _p(rm), None)
        _lib.call("pm_fk_f32", _p(rot), _p(root), _p(offf), 1, pp, F, J, _p(pos2), _p(rm2), None)
        res = [pos.clone(), rm.clone(), pos2.clone(), rm2.clone()]
        _lib.call("pm_fk_from_ortho6d_f32", _p(x6), _p(root), _p(off), 0, pp, F, J, C.c_float(1e-12), _p(pos), _p(rm), _p(q), None)
        res += [pos.clone(), rm.clone(), q.clone()]
        _lib.call("pm_fk_from_ortho6d_f32", _p(x6), _p(root), _p(off), 0, pp, F, J, C.c_float(1e-12), _p(pos), _p(rm), None, None)
        res += [pos.clone(), rm.clone()]
        _lib.call("pm_to_root_dq_f32", _p(rot), _p(root), pp, _p(off), F, J, _p(dq), None)
        _lib.call("pm_from_root_dq_f32", _p(dq), pp, F, J, _p(tr), _p(qo), None)
        _lib.call("pm_from_global_rotations_f32", _p(rot), pp, F, J, _p(lo), None)
        _lib.call("pm_mirror_rotations_f32", _p(rot), pp, None, 0, F, J, _p(mi), None)
        _lib.call("pm_from_root_positions_f32", _p(pos2), pp, _p(off), F, J, _p(ik), None)
        res += [dq, tr, qo, lo, mi, ik]
        torch.cuda.synchronize()
        g.check()
        return [r.clone() for r in res]

    a, b = _both(run)
    for k, (x, y) in enumerate(zip(a, b)):
        if k == len(a) - 1:
            # from_root_positions amplifies last-ulp differences ~100x (rotations from normalised differences of positions; a
            # direction within 0.26 degrees of its rest pose snaps to the identity like the reference's np.isclose): the two
            # builds must agree like either agrees with the float64 oracle (tests/test_ik.py: 2e-4, rare snap flips aside)
            d = torch.minimum((x - y).abs().amax(dim=-1), (x + y).abs().amax(dim=-1))  # q and -q are one rotation
            # (a flipped snap / anti-parallel decision changes a joint's answer completely: at most one element in a thousand)
            assert float(d.median()) <= 1e-6 and float((d > 5e-4).float().mean()) <= 1e-3, (k, float(d.median()), float((d > 5e-4).float().mean()))
            continue
        assert _same(torch, x, y), f"result {k} differs between the builds"


@pytest.mark.parametrize("T,S", [(700, 7), (256 * 3 + 1, 25), (65, 130), (4097, 3)])
def test_frame_coupled_and_elementwise_kernels_under_the_debug_build(T, S):
    import torch

    rng = np.random.default_rng(T + S)
    q_h = rng.standard_normal((T, S, 4)).astype(np.float32)
    d_h = rng.standard_normal((T, S, 8)).astype(np.float32)

    def run():
        g = Guarded(torch)
        q, d8 = g.new((T, S, 4), torch.from_numpy(q_h)), g.new((T, S, 8), torch.from_numpy(d_h))
        oq, od = g.new((T, S, 4)), g.new((T, S, 8))
        ws = g.new((_lib.lib().pm_quat_unroll_workspace_bytes(T, S) // 4 + 4,), dtype=torch.int32)
        _lib.call("pm_quat_unroll_f32", _p(q), T, S, _p(oq), _p(ws), None)
        _lib.call("pm_dq_unroll_f32", _p(d8), T, S, _p(od), _p(ws), None)
        n = T * S
        m, nq, mv, o6, eu = g.new((n, 3, 3)), g.new((n, 4)), g.new((n, 3)), g.new((n, 3, 2)), g.new((n, 3))
        _lib.call("pm_quat_to_matrix_f32", _p(q), n, _p(m), None)
        _lib.call("pm_quat_from_matrix_f32", _p(m), n, _p(nq), None)
        _lib.call("pm_quat_mul_vec_f32", _p(q), _p(m), n, _p(mv), None)
        _lib.call("pm_o6d_from_quat_f32", _p(q), n, _p(o6), None)
        code = g.new((3,), torch.tensor([2, 0, 1], dtype=torch.uint8), dtype=torch.uint8)
        _lib.call("pm_quat_to_euler_f32", _p(q), _p(code), 0, n, _p(eu), None)
        torch.cuda.synchronize()
        g.check()
        return [t.clone() for t in (oq, od, m, nq, mv, o6, eu)]

    a, b = _both(run)
    for k, (x, y) in enumerate(zip(a, b)):
        assert _same(torch, x, y), f"result {k} differs between the builds"


def test_the_debug_build_reports_an_lds_overrun():
    """the checker itself: shrink the allocation the helpers check against (PM_DEBUG exports a knob for exactly this test)
    and the next call must come back as PM_EHIP naming the access, not as silent corruption"""
    import torch

    with _lib.variant("debug") as h:
        assert hasattr(h, "pm_debug_shrink_lds")
        rot = torch.randn((200, 22, 4), device="cuda")
        out = torch.empty((200, 22, 3, 3), device="cuda")
        h.pm_debug_shrink_lds.argtypes = [C.c_int]
        h.pm_debug_shrink_lds(256)  # elementwise.hip tells the checker its tiles own 256 bytes
        with pytest.raises(_lib.PmhipError, match="LDS access outside"):
            _lib.call("pm_quat_to_matrix_f32", _p(rot), 200 * 22, _p(out), None)
        h.pm_debug_shrink_lds(0)
        _lib.call("pm_quat_to_matrix_f32", _p(rot), 200 * 22, _p(out), None)  # and is healthy again
