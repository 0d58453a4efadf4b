// v_dot2c_f32_f16 as the residual step of the hi/lo split: for a pair (p, q) with hb = (f16(p), f16(q)),
//   ra = p + hb.x * (-1) + hb.y * 0,   rb = q + hb.x * 0 + hb.y * (-1)
// should equal p - f16(p) and q - f16(q), which are exactly representable.  1 M random values in (2^-24, 4); the host compares.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* p_in, int n, float* ra_o, float* rb_o, float* ha_o, float* hb_o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n / 2) return;
    const float p = p_in[2 * i], q = p_in[2 * i + 1];
    const h2 hb = {(_Float16)p, (_Float16)q};
    const unsigned hbu = __builtin_bit_cast(unsigned, hb);
    const unsigned k10 = __builtin_bit_cast(unsigned, (h2{(_Float16)-1.0f, (_Float16)0.0f}));
    const unsigned k01 = __builtin_bit_cast(unsigned, (h2{(_Float16)0.0f, (_Float16)-1.0f}));
    float ra = p, rb = q;
    asm volatile("v_dot2c_f32_f16_e32 %0, %1, %2" : "+v"(ra) : "v"(hbu), "v"(k10));
    asm volatile("v_dot2c_f32_f16_e32 %0, %1, %2" : "+v"(rb) : "v"(hbu), "v"(k01));
    ra_o[i] = ra; rb_o[i] = rb; ha_o[i] = (float)hb[0]; hb_o[i] = (float)hb[1];
}
int main() {
    const int n = 1 << 20;
    float* h = new float[n];
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; const float u = (s >> 8) * (1.0f / 16777216.0f); h[i] = exp2f(-24.0f + 26.0f * u); }
    float *d, *o[4];
    hipMalloc(&d, n * 4);
    for (auto& x : o) hipMalloc(&x, n * 2);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    k<<<n / 512, 256>>>(d, n, o[0], o[1], o[2], o[3]);
    float* r[4];
    for (int j = 0; j < 4; ++j) { r[j] = new float[n / 2]; hipMemcpy(r[j], o[j], n * 2, hipMemcpyDeviceToHost); }
    long bad[2][2] = {{0, 0}, {0, 0}};
    double worst[2][2] = {{0, 0}, {0, 0}};
    int shown = 0;
    for (int i = 0; i < n / 2; ++i)
        for (int lane = 0; lane < 2; ++lane) {
            const float v = h[2 * i + lane], f = r[2 + lane][i], got = r[lane][i], want = v - f;
            const int sub = v < 6.104e-5f;
            if (got != want) {
                ++bad[lane][sub];
                const double rel = fabs((double)got - want) / v;
                if (rel > worst[lane][sub]) worst[lane][sub] = rel;
                if (!sub && shown < 4) { printf("  e.g. lane %d: v = %.9g, f16(v) = %.9g, got %.9g, exact %.9g (pair partner %.9g)\n", lane, v, f, got, want, h[2 * i + 1 - lane]); ++shown; }
            }
        }
    for (int lane = 0; lane < 2; ++lane)
        printf("element %d: inexact residuals: %ld normal (worst |err|/v = %.3g), %ld in the f16 subnormal range (worst %.3g)\n", lane, bad[lane][0], worst[lane][0], bad[lane][1], worst[lane][1]);
    return 0;
}
