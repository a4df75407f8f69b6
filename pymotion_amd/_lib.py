"""ctypes binding of ``libpmhip.so`` (the C ABI declared in ``include/pmhip.h``).

The library is the product: there is NO CPU fallback.  If the shared object is missing
(or no HIP device is visible when a kernel is called) every entry point raises
``RuntimeError`` -- loudly, by design (parity claims are about the HIP path only).
Build it in-tree with ``python __graft_entry__.py`` or ``make -C pymotion_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpmhip.so")
# Other builds of the same sources (pymotion_amd/csrc/Makefile): "tuning" reads the PM_* tuning / ablation variables,
# "debug" bounds-checks the kernels and synchronises after every launch, "asan" has its host code under ASan + UBSan.  The product is always "prod"; the others are
# selected explicitly -- `with _lib.variant("tuning"):` in tests and probes, or PMHIP_VARIANT=tuning for a whole process.
VARIANT_PATHS = {
    "prod": LIB_PATH,
    "tuning": os.path.join(_HERE, "libpmhip_tuning.so"),
    "debug": os.path.join(_HERE, "libpmhip_debug.so"),
    "asan": os.path.join(_HERE, "libpmhip_asan.so"),  # host code under ASan + UBSan (tests/test_sanitizers.py)
    "ab": os.path.join(_HERE, "libpmhip_ab.so"),      # scratch build for same-box A/B timing (tools/ab_build.sh); never shipped
    "ab2": os.path.join(_HERE, "libpmhip_ab2.so"),    # a second one (tools/ab_file.sh)
}

PM_OK, PM_EINVAL, PM_ETOPOLOGY, PM_EHIP, PM_EUNSUPPORTED = 0, -1, -2, -3, -4

_f = C.c_void_p  # device float*
_i64, _i32, _int, _flt, _strm = C.c_int64, C.c_int32, C.c_int, C.c_float, C.c_void_p

# name -> argtypes; every function returns int.  Kept in the same order as include/pmhip.h.
SIGNATURES = {
    "pm_version": [],
    "pm_device_count": [],
    "pm_set_device": [_int],
    "pm_get_device": [C.POINTER(_int)],
    "pm_malloc": [C.POINTER(C.c_void_p), C.c_size_t],
    "pm_free": [C.c_void_p],
    "pm_memcpy_h2d": [C.c_void_p, C.c_void_p, C.c_size_t, _strm],
    "pm_memcpy_d2h": [C.c_void_p, C.c_void_p, C.c_size_t, _strm],
    "pm_memset": [C.c_void_p, _int, C.c_size_t, _strm],
    "pm_stream_synchronize": [_strm],
    "pm_event_create": [C.POINTER(C.c_void_p)],
    "pm_event_destroy": [C.c_void_p],
    "pm_event_record": [C.c_void_p, _strm],
    "pm_event_elapsed_ms": [C.c_void_p, C.c_void_p, C.POINTER(_flt)],
    "pm_event_synchronize": [C.c_void_p],
    "pm_stream_create": [C.POINTER(C.c_void_p)],
    "pm_stream_destroy": [_strm],
    "pm_host_alloc": [C.POINTER(C.c_void_p), C.c_size_t],
    "pm_host_free": [C.c_void_p],
    # skeleton ops
    "pm_fk_f32": [_f, _f, _f, _int, C.c_void_p, _i64, _i32, _f, _f, _strm],
    "pm_fk_from_ortho6d_f32": [_f, _f, _f, _int, C.c_void_p, _i64, _i32, _flt, _f, _f, _f, _strm],
    "pm_to_root_dq_f32": [_f, _f, C.c_void_p, _f, _i64, _i32, _f, _strm],
    "pm_to_root_dq_hint_f32": [_f, _f, C.c_void_p, _f, _i64, _i32, _f, _flt, _strm],
    "pm_from_root_dq_f32": [_f, C.c_void_p, _i64, _i32, _f, _f, _strm],
    "pm_from_global_rotations_f32": [_f, C.c_void_p, _i64, _i32, _f, _strm],
    "pm_from_root_positions_f32": [_f, C.c_void_p, _f, _i64, _i32, _f, _strm],
    "pm_mirror_rotations_f32": [_f, C.c_void_p, C.c_void_p, _int, _i64, _i32, _f, _strm],
    # element-wise
    "pm_quat_normalize_f32": [_f, _i64, _flt, _f, _strm],
    "pm_quat_length_f32": [_f, _i64, _f, _strm],
    "pm_quat_to_matrix_f32": [_f, _i64, _f, _strm],
    "pm_quat_from_matrix_f32": [_f, _i64, _f, _strm],
    "pm_quat_mul_f32": [_f, _f, _i64, _f, _strm],
    "pm_quat_mul_vec_f32": [_f, _f, _i64, _f, _strm],
    "pm_quat_conjugate_f32": [_f, _i64, _f, _strm],
    "pm_dq_from_rt_f32": [_f, _f, _i64, _f, _strm],
    "pm_dq_to_rt_f32": [_f, _i64, _f, _f, _strm],
    "pm_dq_from_t_f32": [_f, _i64, _f, _strm],
    "pm_o6d_to_matrix_f32": [_f, _i64, _flt, _f, _strm],
    "pm_o6d_to_quat_f32": [_f, _i64, _flt, _f, _strm],
    "pm_o6d_from_quat_f32": [_f, _i64, _f, _strm],
    "pm_o6d_from_matrix_f32": [_f, _i64, _f, _strm],
    # second wave
    "pm_quat_from_angle_axis_f32": [_f, _f, _i64, _f, _strm],
    "pm_quat_from_scaled_angle_axis_f32": [_f, _i64, _f, _strm],
    "pm_quat_to_angle_axis_f32": [_f, _i64, _f, _f, _strm],
    "pm_quat_to_scaled_angle_axis_f32": [_f, _i64, _f, _strm],
    "pm_quat_from_euler_f32": [_f, C.c_void_p, _int, _i64, _f, _strm],
    "pm_quat_to_euler_f32": [_f, C.c_void_p, _int, _i64, _f, _strm],
    "pm_quat_slerp_f32": [_f, _f, _f, _i64, _int, _f, _strm],
    "pm_quat_from_to_f32": [_f, _f, _i64, _int, _f, _strm],
    "pm_quat_from_to_axis_f32": [_f, _f, _f, _i64, _int, _f, _strm],
    "pm_quat_unroll_workspace_bytes": [_i64, _i32],
    "pm_quat_unroll_f32": [_f, _i64, _i32, _f, C.c_void_p, _strm],
    "pm_dq_unroll_f32": [_f, _i64, _i32, _f, C.c_void_p, _strm],
    "pm_quat_unroll_batched_workspace_bytes": [_i64, _i64, _i32],
    "pm_quat_unroll_batched_f32": [_f, _i64, _i64, _i32, _f, C.c_void_p, _strm],
    "pm_dq_unroll_batched_f32": [_f, _i64, _i64, _i32, _f, C.c_void_p, _strm],
    "pm_bvh_rotations_f32": [_f, C.c_void_p, _i64, _i32, _f, C.c_void_p, _strm],
    "pm_dq_normalize_f32": [_f, _i64, _int, _flt, _f, C.c_void_p, _strm],
    "pm_dq_unit_flags_f32": [_f, _i64, _flt, C.c_void_p, _strm],
    "pm_interpolate_linear_f32": [_f, C.c_void_p, _f, _i64, _i64, _i64, _i64, _f, _strm],
    # measurement helper
    "pm_stream_ceiling_f32": [_f, _f, _i64, _i32, _i32, _strm],
    "pm_stream_plain_f32": [_f, _f, _i64, _i32, _i32, _strm],
    "pm_store_probe_f32": [_f, _f, _i64, C.c_void_p, _strm],
    "pm_scan_floor_probe": [_f, _f, C.c_void_p, _i64, _i64, _i32, C.c_uint32, _i32, _strm],
    "pm_unroll_onepass_f32": [_i32, _f, C.c_void_p, _i64, _i64, _i32, _f, C.c_void_p, C.c_void_p, C.c_void_p, _i64, _strm],
    # host-only introspection (tests of the wide walk's scheduler)
    "pm_fk_wide_plan_debug": [C.c_void_p, _i32, C.c_void_p],
    "pm_step_list_plan_debug": [C.c_void_p, _i32, _i32, _i32, C.c_void_p],
}

_lib = None


def _preload_hip_runtime():
    """Make sure the process ends up with ONE HIP runtime.

    PyTorch-ROCm wheels bundle their own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``) and ask
    for it by the un-versioned file name, while ``libpmhip.so`` asks for the SONAME.  If the system
    runtime under /opt/rocm were mapped first, a later ``import torch`` would map a SECOND runtime and
    whichever initialises second sees no device.  Mapping torch's copy first (by path, no torch
    import) lets the loader satisfy our SONAME request with it, so both share streams and memory
    whatever the import order.  Without torch installed the system runtime is used.
    """
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


class PmhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libpmhip error {code}: {msg}")
        self.code = code


_handles = {}


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension is not built and pymotion_amd has no CPU "
            "fallback. Run `python __graft_entry__.py` (or `make -C pymotion_amd/csrc`)."
        )
    _preload_hip_runtime()
    h = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)  # AttributeError = ABI mismatch, let it surface
        fn.argtypes = argtypes
        fn.restype = C.c_int64 if name.endswith("_bytes") else C.c_int
    h.pm_last_error_string.argtypes = []
    h.pm_last_error_string.restype = C.c_char_p
    h.pm_last_kernel_name.argtypes = []
    h.pm_last_kernel_name.restype = C.c_char_p
    return h


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP extension is not built."""
    global _lib
    if _lib is None:
        name = os.environ.get("PMHIP_VARIANT", "prod")
        if name not in VARIANT_PATHS:
            raise RuntimeError(f"PMHIP_VARIANT={name!r}: expected one of {sorted(VARIANT_PATHS)}")
        _lib = _handles.setdefault(name, _load(VARIANT_PATHS[name]))
    return _lib


class variant:
    """``with variant("tuning"): ...`` -- route every call of this process through another build of the library
    (tests of the tile-group kernels, probes).  Not thread-safe; never used by the product itself."""

    def __init__(self, name):
        if name not in VARIANT_PATHS:
            raise ValueError(f"unknown library variant {name!r}")
        self.name = name

    def __enter__(self):
        global _lib
        self._prev = _lib
        if self.name not in _handles:
            _handles[self.name] = _load(VARIANT_PATHS[self.name])
        _lib = _handles[self.name]
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self._prev
        return False


def last_kernel_name():
    """Name of the kernel the last skeleton-op call of this thread dispatched to (bench.py's roofline.kernel)."""
    return lib().pm_last_kernel_name().decode("utf-8", "replace")


def check(code):
    if code != PM_OK:
        msg = lib().pm_last_error_string().decode("utf-8", "replace")
        if code in (PM_EINVAL, PM_ETOPOLOGY):
            raise ValueError(f"libpmhip: {msg}")
        raise PmhipError(code, msg)


def call(name, *args):
    check(getattr(lib(), name)(*args))


def device_count():
    n = lib().pm_device_count()
    if n < 0:
        check(n)
    return n


def require_device():
    if device_count() < 1:
        raise RuntimeError("pymotion_amd: no HIP device visible and there is no CPU fallback")
