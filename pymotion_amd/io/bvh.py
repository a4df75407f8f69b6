"""BVH ingest for the fk hot path -- the step right before ``fk`` in every example of the reference
(``bvh.load(); rots, pos, parents, offsets, ... = bvh.get_data(); fk(rots, pos[:, 0], offsets, parents)``).

Drop-in for the reading half of ``pymotion.io.bvh.BVH`` (reference: ``pymotion/io/bvh.py:24-161`` load,
``:332-365`` get_data, ``:367-389`` set_data): same ``data`` dictionary, same return tuple.  The text is
parsed on the host by a small tokenizer; the numeric work of ``get_data`` -- Euler -> quaternion
(``quat.from_euler``), sign unrolling along frames (``quat.unroll``) and normalisation -- runs as three
GPU kernels instead of ``np.apply_along_axis`` over strings and a Python loop over frames.
Writing files (``save``), joint removal / reordering and scaling stay with the reference.
"""
import re
import warnings

import numpy as np

from .. import _backend, _ops
from ..rotations import quat


def _be():
    return _backend.numpy_backend()

_ROT = {"Xrotation": "x", "Yrotation": "y", "Zrotation": "z"}
_POS = {"Xposition": 0, "Yposition": 1, "Zposition": 2}


class BVH:
    def __init__(self):
        self.data = None

    # ---- parsing -------------------------------------------------------------------------------------
    def load(self, filename: str):
        """Read a BVH file into ``self.data`` (keys and shapes as the reference, bvh.py:36-54):
        names, offsets [J,3], end_sites, end_sites_parents, parents [J] (root's parent = 0), rot_order
        [J,3] of 'x'|'y'|'z', positions [F,J,3], rotations [F,J,3] (degrees, channel order), frame_time."""
        with open(filename, "r") as fh:
            text = fh.read()
        head, sep, motion = text.partition("MOTION")
        if not sep:
            raise ValueError(f"{filename}: no MOTION section")
        tok = head.replace("{", " { ").replace("}", " } ").split()

        names, offsets, parents, rot_order, pos_order, nchan = [], [], [], [], [], []
        end_sites, end_parents = [], []
        stack = []          # open joints (indices); None marks an End Site block
        pending = None      # joint / end site whose '{' is about to open
        i = 0
        while i < len(tok):
            t = tok[i]
            if t in ("ROOT", "JOINT"):
                names.append(tok[i + 1])
                parents.append(stack[-1] if stack else 0)
                offsets.append([0.0, 0.0, 0.0])
                rot_order.append(None)
                pos_order.append(None)
                nchan.append(0)
                pending = len(names) - 1
                i += 2
            elif t == "End" and i + 1 < len(tok) and tok[i + 1] == "Site":
                end_parents.append(stack[-1])
                end_sites.append([0.0, 0.0, 0.0])
                pending = None
                i += 2
            elif t == "{":
                stack.append(pending)
                pending = -1
                i += 1
            elif t == "}":
                stack.pop()
                i += 1
            elif t == "OFFSET":
                v = [float(x) for x in tok[i + 1:i + 4]]
                if stack[-1] is None:
                    end_sites[-1] = v
                else:
                    offsets[stack[-1]] = v
                i += 4
            elif t == "CHANNELS":
                n = int(tok[i + 1])
                ch = tok[i + 2:i + 2 + n]
                j = stack[-1]
                nchan[j] = n
                if n == 6:
                    pos_order[j] = [_POS[c] for c in ch[:3]]
                    rot_order[j] = [_ROT[c] for c in ch[3:6]]
                elif n == 3:
                    rot_order[j] = [_ROT[c] for c in ch]
                else:
                    raise ValueError("Unknown number of channels")  # as the reference (bvh.py:120)
                i += 2 + n
            else:
                i += 1  # HIERARCHY and anything unknown

        # "Frames: N" / "Frame Time: dt" (io/bvh.py:133-147 of the reference), then N rows of numbers.  The header is
        # tokenised in Python; the numbers -- almost all of the file -- go through NumPy's C text parser.
        m = re.search(r"Frames:\s*(\d+)\s+Frame\s+Time:\s*(\S+)", motion)
        if m is None:
            raise ValueError(f"{filename}: malformed MOTION header")
        n_frames = int(m.group(1))
        frame_time = float(m.group(2))
        J = len(names)
        width = sum(nchan)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", DeprecationWarning)  # text-mode fromstring: not deprecated, but some NumPy builds warn
            flat = np.fromstring(motion[m.end():], dtype=np.float64, sep=" ")
        if flat.size < n_frames * width:
            raise ValueError(f"{filename}: {flat.size} motion values, expected {n_frames} x {width}")
        vals = flat[:n_frames * width].reshape(n_frames, width)

        offsets = np.array(offsets, dtype=np.float64)
        positions = np.tile(offsets, (n_frames, 1)).reshape(n_frames, J, 3)
        rotations = np.zeros((n_frames, J, 3))
        col = 0
        for j in range(J):
            if nchan[j] == 6:
                positions[:, j, pos_order[j]] = vals[:, col:col + 3]
                rotations[:, j] = vals[:, col + 3:col + 6]
            elif nchan[j] == 3:
                rotations[:, j] = vals[:, col:col + 3]
            col += nchan[j]

        self.data = {
            "names": np.array(names),
            "offsets": offsets,
            "end_sites": np.array(end_sites),
            "end_sites_parents": np.array(end_parents),
            "parents": np.array(parents),
            "rot_order": np.array(rot_order),
            "positions": positions,
            "rotations": rotations,
            "frame_time": frame_time,
        }

    # ---- GPU part ------------------------------------------------------------------------------------
    def get_data(self):
        """-> (rots [F,J,4] unrolled unit quaternions, pos [F,J,3], parents, offsets, end_sites,
        end_sites_parents) -- reference bvh.py:332-365; the rotations go through from_euler -> unroll
        (frame axis) -> normalize on the GPU, fused into one kernel up to 64 joints."""
        d = self.data
        # the reference tiles the per-joint order over the frames (bvh.py:352); the kernel takes the [J, 3] table itself, and the
        # angles in degrees as the file holds them: from_euler, the sign scan along the frames and normalize are ONE launch
        # (pm_bvh_rotations_f32; three launches -- and three trips over the bus through this NumPy door -- before round 4)
        rots = _ops.bvh_rotations(_be(), d["rotations"], d["rot_order"])
        return rots, d["positions"], d["parents"], d["offsets"], d["end_sites"], d["end_sites_parents"]

    def set_data(self, rots, pos):
        """Store quaternions back as Euler angles in the file's channel order (reference bvh.py:367-389)."""
        assert self.data is not None and self.data["rot_order"] is not None, "load a BVH file first"
        self.data["rotations"] = np.degrees(_ops.quat_to_euler(_be(), rots, self.data["rot_order"], per_joint_table=True))
        self.data["positions"] = pos
