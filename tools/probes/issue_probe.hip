// How do matrix and vector instructions share a SIMD's issue slots on gfx950?  Cycles per loop iteration (s_memtime) of
//   8 x v_mfma_f32_16x16x32_f16 (or 4 x v_mfma_f32_32x32x16_f16: the same flops), independent accumulators
//   N x v_fma_f32 / v_exp_f32 / v_cvt_pk_f16_f32 on registers of their own
// alone, interleaved in one wave, and split over the two waves of a SIMD (wave A: matrix only, wave B: vector only).
// hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define ITERS 20000

template <int MF, int NV, int KIND, int GEO>   // MF: mfma on/off; NV: vector instructions per iteration; KIND 0 fma 1 exp 2 cvt_pk; GEO 0: 16x16x32, 1: 32x32x16
__device__ __forceinline__ float body(float seed, long long* cycles) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed - i); }
    f4 c[8];
    f16v d[4];
    for (int i = 0; i < 8; ++i) c[i] = f4{seed, seed, seed, seed};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) d[i][j] = seed;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed * (i + 1);
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (MF) {
                if (GEO == 0) c[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[s], 0, 0, 0);
                else if ((s & 1) == 0) d[s >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d[s >> 1], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < NV / 8; ++k) {
                const int r = (s * (NV / 8) + k) & 15;
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[r]) : "v"(seed));
                else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
                else if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[r]) : "v"(seed));
                else if (KIND == 3) asm volatile("v_dot2c_f32_f16_e32 %0, -1.0, %1" : "+v"(v[r]) : "v"(seed));
                else asm volatile("v_fma_mixlo_f16 %0, %0, 1.0, %1 op_sel_hi:[0,0,1]" : "+v"(v[r]) : "v"(seed));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    *cycles = t1 - t0;
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += c[i][0] + c[i][3];
    for (int i = 0; i < 4; ++i) acc += d[i][0] + d[i][15];
    for (int i = 0; i < 16; ++i) acc += v[i];
    return acc;
}

template <int MF, int NV, int KIND, int GEO, int SPLIT>
__global__ __launch_bounds__(512) void probe(float seed, float* sink, long long* out) {
    // 512 threads = 8 waves = 2 per SIMD (waves w and w+4 share a SIMD).  SPLIT: waves 0-3 matrix only, waves 4-7 vector only.
    const int wave = threadIdx.x >> 6;
    long long cyc = 0;
    float r;
    if (SPLIT) r = (wave < 4) ? body<1, 0, KIND, GEO>(seed, &cyc) : body<0, NV, KIND, GEO>(seed, &cyc);
    else r = body<MF, NV, KIND, GEO>(seed, &cyc);
    if (r == 123.456f) sink[0] = r;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = cyc;
}

template <int MF, int NV, int KIND, int GEO, int SPLIT> void run(const char* name, int threads) {
    float* sink; long long* out;
    hipMalloc(&sink, 4); hipMalloc(&out, 8 * sizeof(long long));
    hipMemset(out, 0, 8 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MF, NV, KIND, GEO, SPLIT><<<1, threads>>>(1.0f, sink, out);      // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MF, NV, KIND, GEO, SPLIT><<<1, threads>>>(1.0f, sink, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-64s", name);
    for (int w = 0; w < threads / 64; ++w) printf(" %7.1f", (double)h[w] / ITERS);
    printf("   kernel %.1f us = %.1f ns per iteration\n", ms * 1e3, ms * 1e6 / ITERS);
    hipFree(sink); hipFree(out);
}

int main() {
    // one wave per SIMD (256 threads)
    run<1, 0, 0, 0, 0>("1 wave/SIMD: 8 mfma16 alone", 256);
    run<1, 0, 0, 1, 0>("1 wave/SIMD: 4 mfma32 alone", 256);
    run<0, 32, 0, 0, 0>("1 wave/SIMD: 32 v_fma alone", 256);
    run<0, 32, 1, 0, 0>("1 wave/SIMD: 32 v_exp alone", 256);
    run<0, 32, 2, 0, 0>("1 wave/SIMD: 32 v_cvt_pk alone", 256);
    run<1, 32, 0, 0, 0>("1 wave/SIMD: 8 mfma16 + 32 v_fma interleaved", 256);
    run<1, 32, 0, 1, 0>("1 wave/SIMD: 4 mfma32 + 32 v_fma interleaved", 256);
    run<1, 16, 0, 0, 0>("1 wave/SIMD: 8 mfma16 + 16 v_fma interleaved", 256);
    run<1, 32, 1, 0, 0>("1 wave/SIMD: 8 mfma16 + 32 v_exp interleaved", 256);
    run<1, 16, 1, 0, 0>("1 wave/SIMD: 8 mfma16 + 16 v_exp interleaved", 256);
    run<1, 32, 1, 1, 0>("1 wave/SIMD: 4 mfma32 + 32 v_exp interleaved", 256);
    run<0, 32, 3, 0, 0>("1 wave/SIMD: 32 v_dot2c alone", 256);
    run<0, 32, 4, 0, 0>("1 wave/SIMD: 32 v_fma_mixlo alone", 256);
    run<1, 32, 3, 0, 0>("1 wave/SIMD: 8 mfma16 + 32 v_dot2c interleaved", 256);
    run<1, 32, 4, 0, 0>("1 wave/SIMD: 8 mfma16 + 32 v_fma_mixlo interleaved", 256);
    run<1, 32, 3, 1, 0>("1 wave/SIMD: 4 mfma32 + 32 v_dot2c interleaved", 256);
    run<1, 32, 4, 1, 0>("1 wave/SIMD: 4 mfma32 + 32 v_fma_mixlo interleaved", 256);
    run<1, 32, 0, 1, 0>("1 wave/SIMD: 4 mfma32 + 32 v_fma (again)", 256);
    run<1, 16, 4, 1, 0>("1 wave/SIMD: 4 mfma32 + 16 v_fma_mixlo interleaved", 256);
    // sweep: vector instructions per MFMA, by kind (KIND 0 v_fma, 1 v_exp, 2 v_cvt_pk, 4 v_fma_mixlo), both MFMA shapes
    run<1, 8, 4, 1, 0>("sweep 4 mfma32 +  8 mixlo", 256);
    run<1, 24, 4, 1, 0>("sweep 4 mfma32 + 24 mixlo", 256);
    run<1, 48, 4, 1, 0>("sweep 4 mfma32 + 48 mixlo", 256);
    run<1, 64, 4, 1, 0>("sweep 4 mfma32 + 64 mixlo", 256);
    run<1, 64, 1, 1, 0>("sweep 4 mfma32 + 64 v_exp", 256);
    run<1, 64, 0, 1, 0>("sweep 4 mfma32 + 64 v_fma", 256);
    run<1, 64, 2, 1, 0>("sweep 4 mfma32 + 64 v_cvt_pk", 256);
    run<1, 64, 4, 0, 0>("sweep 8 mfma16 + 64 mixlo", 256);
    run<1, 64, 1, 0, 0>("sweep 8 mfma16 + 64 v_exp", 256);
    run<1, 64, 2, 0, 0>("sweep 8 mfma16 + 64 v_cvt_pk", 256);
    run<0, 64, 4, 0, 0>("sweep 64 mixlo alone", 256);
    run<0, 64, 1, 0, 0>("sweep 64 v_exp alone", 256);
    // two waves per SIMD (512 threads)
    run<1, 0, 0, 0, 0>("2 waves/SIMD: 8 mfma16 alone (each)", 512);
    run<0, 32, 0, 0, 0>("2 waves/SIMD: 32 v_fma alone (each)", 512);
    run<1, 32, 0, 0, 0>("2 waves/SIMD: 8 mfma16 + 32 v_fma interleaved (each)", 512);
    run<1, 32, 0, 1, 0>("2 waves/SIMD: 4 mfma32 + 32 v_fma interleaved (each)", 512);
    run<1, 32, 1, 0, 0>("2 waves/SIMD: 8 mfma16 + 32 v_exp interleaved (each)", 512);
    run<1, 32, 0, 0, 1>("2 waves/SIMD: wave A 8 mfma16, wave B 32 v_fma", 512);
    run<1, 32, 0, 1, 1>("2 waves/SIMD: wave A 4 mfma32, wave B 32 v_fma", 512);
    run<1, 64, 0, 0, 1>("2 waves/SIMD: wave A 8 mfma16, wave B 64 v_fma", 512);
    run<1, 32, 1, 0, 1>("2 waves/SIMD: wave A 8 mfma16, wave B 32 v_exp", 512);
    return 0;
}
