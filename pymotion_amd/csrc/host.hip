// host.hip -- host-side plumbing of libpmhip.so: error reporting, topology validation, device
// memory / stream / event helpers for callers without their own HIP runtime binding.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "common.hpp"

namespace pm {

static thread_local char g_err[512] = "";

static thread_local char g_kernel[256] = "";

void set_kernel_name(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifdef PM_TUNING
int tune_env(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
#endif

int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return PM_OK;
    set_error("%s: %s (%s)", what, hipGetErrorString(e), hipGetErrorName(e));
    (void)hipGetLastError();  // clear the sticky error
    return PM_EHIP;
}

// parents[0] is ignored (ops/skeleton.py:53); every other joint must come after its parent, the
// order the reference's in-place loops need (ops/skeleton.py:51-58, :234-241).
int pack_parents(const int32_t *parents, int32_t J, Parents &out) {
    out.p[0] = 0;
    for (int32_t i = 1; i < J; ++i) {
        const int32_t p = parents[i];
        if (p < 0 || p >= i) {
            set_error("parents[%d] = %d: joints must be in topological order (0 <= parents[i] < i)", i, p);
            return PM_ETOPOLOGY;
        }
        out.p[i] = p;
    }
    return PM_OK;
}

}  // namespace pm

using namespace pm;

extern "C" int pm_version(void) { return 1; }
extern "C" const char *pm_last_error_string(void) { return g_err; }
extern "C" const char *pm_last_kernel_name(void) { return g_kernel; }

extern "C" int pm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipErrorNoDevice) { (void)hipGetLastError(); return 0; }
    if (e != hipSuccess) return check_hip(e, "hipGetDeviceCount");
    return n;
}
extern "C" int pm_set_device(int device) { return check_hip(hipSetDevice(device), "hipSetDevice"); }
extern "C" int pm_get_device(int *device) {
    PM_CHECK_ARGS(device, "pm_get_device: null");
    return check_hip(hipGetDevice(device), "hipGetDevice");
}
extern "C" int pm_malloc(void **dptr, size_t bytes) {
    PM_CHECK_ARGS(dptr, "pm_malloc: null");
    *dptr = nullptr;
    if (bytes == 0) return PM_OK;
    return check_hip(hipMalloc(dptr, bytes), "hipMalloc");
}
extern "C" int pm_free(void *dptr) { return dptr ? check_hip(hipFree(dptr), "hipFree") : PM_OK; }
extern "C" int pm_memcpy_h2d(void *dst, const void *src, size_t bytes, pm_stream_t s) {
    if (bytes == 0) return PM_OK;
    return check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(s)), "hipMemcpyAsync(H2D)");
}
extern "C" int pm_memcpy_d2h(void *dst, const void *src, size_t bytes, pm_stream_t s) {
    if (bytes == 0) return PM_OK;
    return check_hip(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(s)), "hipMemcpyAsync(D2H)");
}
extern "C" int pm_memset(void *dst, int value, size_t bytes, pm_stream_t s) {
    if (bytes == 0) return PM_OK;
    return check_hip(hipMemsetAsync(dst, value, bytes, static_cast<hipStream_t>(s)), "hipMemsetAsync");
}
extern "C" int pm_stream_synchronize(pm_stream_t s) {
    return check_hip(hipStreamSynchronize(static_cast<hipStream_t>(s)), "hipStreamSynchronize");
}
extern "C" int pm_event_create(void **ev) {
    PM_CHECK_ARGS(ev, "pm_event_create: null");
    hipEvent_t e;
    int r = check_hip(hipEventCreate(&e), "hipEventCreate");
    *ev = r ? nullptr : (void *)e;
    return r;
}
extern "C" int pm_event_destroy(void *ev) { return ev ? check_hip(hipEventDestroy((hipEvent_t)ev), "hipEventDestroy") : PM_OK; }
extern "C" int pm_event_record(void *ev, pm_stream_t s) {
    return check_hip(hipEventRecord((hipEvent_t)ev, static_cast<hipStream_t>(s)), "hipEventRecord");
}
extern "C" int pm_event_synchronize(void *ev) {
    PM_CHECK_ARGS(ev, "pm_event_synchronize: null");
    return check_hip(hipEventSynchronize((hipEvent_t)ev), "hipEventSynchronize");
}
extern "C" int pm_stream_create(pm_stream_t *stream) {
    PM_CHECK_ARGS(stream, "pm_stream_create: null");
    hipStream_t s;
    int r = check_hip(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
    *stream = r ? nullptr : (pm_stream_t)s;
    return r;
}
extern "C" int pm_stream_destroy(pm_stream_t stream) {
    return stream ? check_hip(hipStreamDestroy(static_cast<hipStream_t>(stream)), "hipStreamDestroy") : PM_OK;
}
extern "C" int pm_host_alloc(void **hptr, size_t bytes) {
    PM_CHECK_ARGS(hptr, "pm_host_alloc: null");
    *hptr = nullptr;
    if (bytes == 0) return PM_OK;
    return check_hip(hipHostMalloc(hptr, bytes, hipHostMallocDefault), "hipHostMalloc");
}
extern "C" int pm_host_free(void *hptr) { return hptr ? check_hip(hipHostFree(hptr), "hipHostFree") : PM_OK; }
extern "C" int pm_event_elapsed_ms(void *start, void *stop, float *ms) {
    PM_CHECK_ARGS(start && stop && ms, "pm_event_elapsed_ms: null");
    if (int r = check_hip(hipEventSynchronize((hipEvent_t)stop), "hipEventSynchronize")) return r;
    return check_hip(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop), "hipEventElapsedTime");
}
