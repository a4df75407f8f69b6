"""CPU: the host side of round 6's step-list kernels (pm_step_list_plan_debug; csrc/dqwide.hip: dq_wide_words, csrc/mirror.hip: mirror_wide_words) -- the step words the
kernels hold in registers.  Every joint but the root exactly once, at the earliest one step after its parent; to_root_dual_quat: the root's children compose with the
identity slot (they stay local, pymotion/ops/skeleton.py:236-237), mirror: with the root's slot as parked; slots in bytes of the op's records; idle quads on the idle
slot; at most ceil((J - 1) / W) + depth steps; trees that need more than 48 steps at a width are declined there and taken at a narrower one."""
import ctypes as C

import numpy as np
import pytest

from pymotion_amd import _lib
from pymotion_amd import synthetic as syn

STRIDE = 56


def _trees(J, rng):
    yield "chain", np.maximum(np.arange(J) - 1, 0).astype(np.int32)
    yield "star", np.zeros(J, dtype=np.int32)
    yield "random", syn.random_parents(J, rng).astype(np.int32)
    yield "narrow", np.concatenate([[0], [rng.integers(max(0, j - 3), j) for j in range(1, J)]]).astype(np.int32)
    if J == 52:
        yield "smplh", np.asarray(syn.PARENTS_52, dtype=np.int32)
    if J == 22:
        yield "body", np.asarray(syn.PARENTS_22, dtype=np.int32)


@pytest.mark.parametrize("op,slot", [(0, 32), (1, 16)])
@pytest.mark.parametrize("J", [1, 2, 16, 22, 49, 52, 100, 129, 512])
def test_step_words_schedule_every_joint_once_after_its_parent(J, op, slot):
    h = _lib.lib()
    buf = (C.c_uint32 * (16 * STRIDE))()
    rng = np.random.default_rng(J)
    for kind, parents in _trees(J, rng):
        depth = int(syn.depth_of(parents).max())
        took = 0
        for fpw in (1, 2, 4, 8):
            W = 16 // fpw
            n = h.pm_step_list_plan_debug(parents.ctypes.data_as(C.c_void_p), J, op, fpw, buf)
            if n == _lib.PM_EUNSUPPORTED:  # declined: only when list scheduling itself may need more than the 48 steps the list holds
                assert depth > 48 or -(-(J - 1) // W) + depth > 48, (J, kind, fpw, depth)
                continue
            took += 1
            assert max(depth, -(-(J - 1) // W)) <= n <= min(48, -(-(J - 1) // W) + depth), (J, kind, fpw, n, depth)
            words = np.frombuffer(buf, dtype=np.uint32).reshape(16, STRIDE)
            idle = ((J + 1) * slot) | ((J * slot) << 16)
            assert (words[W:] == idle).all() and (words[:, n:] == idle).all()   # quads a frame does not have, steps the list does not have
            step_of = {0: -1}                                                   # the root takes no step: its slot is parked
            for s in range(n):
                for k in range(W):
                    w = int(words[k, s])
                    if w == idle:
                        continue
                    own, par = w & 0xFFFF, w >> 16
                    assert own % slot == 0 and par % slot == 0
                    j, p = own // slot, par // slot
                    assert 0 < j < J and j not in step_of, (J, kind, fpw, s, k, j)
                    step_of[j] = s
                    if op == 0 and parents[j] == 0:
                        assert p == J                                            # the identity slot: the root's children stay local
                    else:
                        assert p == parents[j] and step_of[p] < s, (j, p, parents[j])
            assert len(step_of) == J
        assert took >= 1 or depth > 48


def test_step_list_plan_argument_checks():
    h = _lib.lib()
    buf = (C.c_uint32 * (16 * STRIDE))()
    par = np.zeros(4, dtype=np.int32)
    pp = par.ctypes.data_as(C.c_void_p)
    assert h.pm_step_list_plan_debug(pp, 4, 2, 1, buf) == _lib.PM_EINVAL
    assert h.pm_step_list_plan_debug(pp, 4, 0, 3, buf) == _lib.PM_EINVAL
    assert h.pm_step_list_plan_debug(None, 4, 0, 1, buf) == _lib.PM_EINVAL
    bad = np.array([0, 2, 1], np.int32)
    assert h.pm_step_list_plan_debug(bad.ctypes.data_as(C.c_void_p), 3, 0, 1, buf) == _lib.PM_ETOPOLOGY
    assert h.pm_step_list_plan_debug(pp, 4, 1, 1, buf) == 1   # a star at sixteen joints a step: every child in the first step
    assert h.pm_step_list_plan_debug(pp, 4, 1, 8, buf) == 2   # ... at two joints a step: three children, two steps
