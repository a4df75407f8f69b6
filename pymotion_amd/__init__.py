"""pymotion_amd -- MI355X-native batched forward kinematics and rotation ops.

Drop-in for the hot path of UPC-ViRVIG/pymotion: ``pymotion_amd.ops.skeleton{,_torch}`` and
``pymotion_amd.rotations.{quat,dual_quat,ortho6d}{,_torch}`` keep the reference's function
names and signatures; the work is done by hand-written gfx950 kernels in ``libpmhip.so``
(C ABI: ``include/pmhip.h``).  No CPU fallback.
"""
__version__ = "0.1.0"


def trim():
    """Release the NumPy door's cached device blocks (per HIP device) and host staging buffers."""
    from ._backend import trim as _trim

    _trim()
