// accuracy of v_sin_f32 / v_cos_f32 (argument in revolutions) against float64 sin / cos
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float *deg, float *s, float *c, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float rev = deg[i] * (1.0f / 720.0f);  // half angle in revolutions
    s[i] = __builtin_amdgcn_sinf(rev);
    c[i] = __builtin_amdgcn_cosf(rev);
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n), hs(n), hc(n);
    for (int i = 0; i < n; ++i) h[i] = -720.0f + 1440.0f * (float)i / n + ((i % 7) * 1e-4f);
    for (int i = 0; i < 4096; ++i) h[i] = (i - 2048) * 1e-3f;            // around zero
    for (int i = 4096; i < 8192; ++i) h[i] = 360.0f + (i - 6144) * 1e-3f;  // half angle around pi
    for (int i = 8192; i < 12288; ++i) h[i] = 50000.0f + (i - 8192) * 0.37f;  // large angles
    float *d, *ds, *dc;
    hipMalloc(&d, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, ds, dc, n);
    hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, es_l = 0, ec_l = 0; int is = 0, ic = 0;
    for (int i = 0; i < n; ++i) {
        const double x = (double)h[i] * M_PI / 360.0;
        const double a = fabs(hs[i] - sin(x)), b = fabs(hc[i] - cos(x));
        if (fabs(h[i]) < 2000) { if (a > es) { es = a; is = i; } if (b > ec) { ec = b; ic = i; } }
        else { if (a > es_l) es_l = a; if (b > ec_l) ec_l = b; }
    }
    printf("|deg| < 2000: max abs err sin %.3e (deg %.4f) cos %.3e (deg %.4f); large angles: sin %.3e cos %.3e\n", es, h[is], ec, h[ic], es_l, ec_l);
    return 0;
}
