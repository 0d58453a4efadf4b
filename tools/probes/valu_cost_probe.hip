// Issue cost of single vector instructions on gfx950: 32 independent copies per loop iteration, one wave per SIMD;
// cycles per instruction = (cycles per iteration - loop overhead) / 32.   hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 20000
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define KERNEL(NAME, ASM)                                                                         \
__global__ __launch_bounds__(256) void NAME(float seed, float* sink, long long* out) {          \
    float v[16], w[16], z[16];                                                                    \
    for (int i = 0; i < 16; ++i) { v[i] = seed * (i + 1); w[i] = seed + i; z[i] = seed - i; }     \
    const long long t0 = __builtin_readcyclecounter();                                            \
    _Pragma("unroll 1") for (int it = 0; it < ITERS; ++it) {                                      \
        _Pragma("unroll") for (int rep = 0; rep < 2; ++rep) {                                     \
            _Pragma("unroll") for (int r = 0; r < 16; ++r) asm volatile(ASM : "+v"(v[r]), "+v"(w[r]) : "v"(z[(r + 5) & 15])); \
        }                                                                                         \
    }                                                                                             \
    const long long t1 = __builtin_readcyclecounter();                                            \
    float acc = 0; for (int i = 0; i < 16; ++i) acc += v[i];                                      \
    if (acc == 123.456f) sink[0] = acc;                                                           \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                 \
}
KERNEL(k_empty, "; nothing %0 %1 %2")
KERNEL(k_add_e32, "v_add_f32_e32 %0, %1, %0")
KERNEL(k_max_e32, "v_max_f32_e32 %0, %1, %0")
KERNEL(k_mul_e32, "v_mul_f32_e32 %0, %1, %0")
KERNEL(k_fmac_e32, "v_fmac_f32_e32 %0, %1, %2")
KERNEL(k_fma_3src, "v_fma_f32 %0, %1, %2, %0")
KERNEL(k_fma_same, "v_fma_f32 %0, %0, %1, %0")
KERNEL(k_mov, "v_mov_b32_e32 %0, %1")
KERNEL(k_cvt_pk, "v_cvt_pk_f16_f32 %0, %1, %2")
KERNEL(k_mixlo, "v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]")
KERNEL(k_mixhi, "v_fma_mixhi_f16 %0, %1, %2, %0 op_sel_hi:[0,0,1]")
KERNEL(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %1, %2, %0")
KERNEL(k_dot2c_f32_f16, "v_dot2c_f32_f16_e32 %0, %1, %2")
KERNEL(k_dot2c_literal, "v_dot2c_f32_f16_e32 %0, 0xbc000000, %1")
KERNEL(k_dot2c_inline, "v_dot2c_f32_f16_e32 %0, -1.0, %1")
KERNEL(k_add_literal, "v_add_f32_e32 %0, 0x3fc00001, %0")
KERNEL(k_split_dot2, "v_cvt_pk_f16_f32 %1, %0, %2\n v_dot2c_f32_f16_e32 %0, -1.0, %1\n v_cvt_pk_f16_f32 %0, %0, %2")
KERNEL(k_split_mix, "v_cvt_pk_f16_f32 %1, %0, %2\n v_fma_mixlo_f16 %0, %0, 1.0, -%1 op_sel_hi:[0,0,1]")
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16_e32 %0, %1")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %1, %2")
KERNEL(k_exp, "v_exp_f32_e32 %0, %1")
KERNEL(k_rcp, "v_rcp_f32_e32 %0, %1")
KERNEL(k_max3, "v_max3_f32 %0, %1, %2, %0")
KERNEL(k_xor, "v_xor_b32_e32 %0, %1, %0")
KERNEL(k_cndmask, "v_cndmask_b32_e32 %0, %1, %0, vcc")
KERNEL(k_perm32swap, "v_permlane32_swap_b32_e32 %0, %1")
KERNEL(k_perm16swap, "v_permlane16_swap_b32_e32 %0, %1")
KERNEL(k_readlane_free, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk_fma(float seed, float* sink, long long* out) {
    f2 v[16], w[16], z[16];
    for (int i = 0; i < 16; ++i) { v[i] = f2{seed * (i + 1), seed}; w[i] = f2{seed + i, seed}; z[i] = f2{seed - i, seed}; }
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(w[r]), "v"(z[(r + 5) & 15]));
    }
    const long long t1 = __builtin_readcyclecounter();
    float acc = 0; for (int i = 0; i < 16; ++i) acc += v[i][0] + v[i][1];
    if (acc == 123.456f) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
#define RUN(K) do { hipMemset(out, 0, 64); K<<<1, 256>>>(1.0f, sink, out); hipDeviceSynchronize(); long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); \
    const double per = (double)h[0] / ITERS; if (base < 0) base = per; printf("%-18s %8.1f cycles per 32  -> %5.2f per instruction\n", #K, per, (per - base) / 32); } while (0)
int main() {
    float* sink; long long* out; hipMalloc(&sink, 4); hipMalloc(&out, 64);
    double base = -1;
    RUN(k_empty); RUN(k_add_e32); RUN(k_max_e32); RUN(k_mul_e32); RUN(k_fmac_e32); RUN(k_fma_3src); RUN(k_fma_same); RUN(k_mov);
    RUN(k_cvt_pk); RUN(k_dot2_f32_f16); RUN(k_dot2c_f32_f16); RUN(k_dot2c_literal); RUN(k_dot2c_inline); RUN(k_add_literal); RUN(k_split_dot2); RUN(k_split_mix); RUN(k_cvt_f32_f16); RUN(k_pk_add_f16); RUN(k_mixlo); RUN(k_mixhi); RUN(k_exp); RUN(k_rcp); RUN(k_max3); RUN(k_xor); RUN(k_cndmask); RUN(k_perm32swap);
    RUN(k_perm16swap); RUN(k_readlane_free); RUN(k_pk_fma);
    return 0;
}
