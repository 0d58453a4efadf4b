// Do stores that rewrite the same lines stay in L2 (write-back) or reach the fabric every time (write-through)?
// Each workgroup owns an 8-KiB region and rewrites it REPS times; WRITE_SIZE (rocprofv3 --pmc) per kernel tells:
// 2 MiB total = write-back, REPS x 2 MiB = write-through.   hipcc --offload-arch=gfx950 -O3 -o l2wb l2_writeback_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REPS 64
#define NAMED(N, K) __global__ void N(float* buf, float seed) { \
    float* mine = buf + (size_t)blockIdx.x * 2048; float acc = 0.f; \
    for (int r = 0; r < REPS; ++r) { \
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) { const float v = seed + r + i; \
            if (K == 0) mine[i] = v; \
            else if (K == 1) __builtin_nontemporal_store(v, mine + i); \
            else if (K == 2) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
            else if (K == 3) __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            else __hip_atomic_store(mine + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); } \
        __syncthreads(); \
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) acc += mine[(i + 64) & 2047]; \
        __syncthreads(); } \
    if (acc == 12345.678f) buf[0] = acc; }
NAMED(plain_store_hipMalloc, 0)
NAMED(nontemporal_store_hipMalloc, 1)
NAMED(workgroup_scope_store_hipMalloc, 2)
NAMED(agent_scope_store_hipMalloc, 3)
NAMED(wavefront_scope_store_hipMalloc, 4)
NAMED(plain_store_uncached, 0)
NAMED(plain_store_finegrained, 0)
NAMED(plain_store_managed, 0)
// the same rewrites, 50 / 400 us apart: does a dirty line that nobody touches leave L2 on its own?
template <int SLEEPS> __device__ void spaced(float* buf, float seed) {
    float* mine = buf + (size_t)blockIdx.x * 2048;
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) {
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) mine[i] = seed + r + i;
        __syncthreads();
        for (int i = threadIdx.x; i < 2048; i += blockDim.x) acc += mine[(i + 64) & 2047];
        for (int k = 0; k < SLEEPS; ++k) __builtin_amdgcn_s_sleep(127);          // 127 x 64 cycles ~ 3.4 us
        __syncthreads();
    }
    if (acc == 12345.678f) buf[0] = acc;
}
__global__ void plain_store_16x_spaced_50us(float* buf, float seed) { spaced<15>(buf, seed); }
__global__ void plain_store_16x_spaced_400us(float* buf, float seed) { spaced<120>(buf, seed); }
// ... and with other workgroups streaming 64 MiB through the same L2 meanwhile (blockIdx >= 256 only stream)
__global__ void plain_store_16x_spaced_400us_beside_a_stream(float* buf, float* big, float seed) {
    if (blockIdx.x < 256) { spaced<120>(buf, seed); return; }
    float acc = 0.f;
    const size_t n = (size_t)16 << 20;
    for (size_t i = (size_t)(blockIdx.x - 256) * blockDim.x + threadIdx.x; i < n; i += (size_t)(gridDim.x - 256) * blockDim.x) acc += big[i];
    if (acc == 12345.678f) buf[0] = acc;
}
// footprint: 16 rewrites of F x 8 KiB per workgroup (256 workgroups: F x 2 MiB in all, F x 256 KiB per XCD's 4-MiB L2)
template <int F> __device__ void footprint(float* buf, float seed) {
    float* mine = buf + (size_t)blockIdx.x * 2048 * F;
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) {
        for (int i = threadIdx.x; i < 2048 * F; i += blockDim.x) mine[i] = seed + r + i;
        __syncthreads();
        for (int i = threadIdx.x; i < 2048 * F; i += blockDim.x) acc += mine[(i + 64) % (2048 * F)];
        __syncthreads();
    }
    if (acc == 12345.678f) buf[0] = acc;
}
__global__ void rewrite16x_total_2MiB(float* b, float s) { footprint<1>(b, s); }
__global__ void rewrite16x_total_4MiB(float* b, float s) { footprint<2>(b, s); }
__global__ void rewrite16x_total_8MiB(float* b, float s) { footprint<4>(b, s); }
__global__ void rewrite16x_total_12MiB(float* b, float s) { footprint<6>(b, s); }
__global__ void rewrite16x_total_16MiB(float* b, float s) { footprint<8>(b, s); }
__global__ void rewrite16x_total_24MiB(float* b, float s) { footprint<12>(b, s); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); } } while (0)
int main() {
    const size_t bytes = 256 * 2048 * sizeof(float);
    float *a = nullptr, *u = nullptr, *f = nullptr, *m = nullptr;
    CK(hipMalloc(&a, bytes));
    CK(hipExtMallocWithFlags((void**)&u, bytes, hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void**)&f, bytes, hipDeviceMallocFinegrained));
    CK(hipMallocManaged(&m, bytes));
    dim3 g(256), b(256);
    plain_store_hipMalloc<<<g, b>>>(a, 1.f);
    nontemporal_store_hipMalloc<<<g, b>>>(a, 2.f);
    workgroup_scope_store_hipMalloc<<<g, b>>>(a, 3.f);
    agent_scope_store_hipMalloc<<<g, b>>>(a, 4.f);
    wavefront_scope_store_hipMalloc<<<g, b>>>(a, 5.f);
    if (u) plain_store_uncached<<<g, b>>>(u, 6.f);
    if (f) plain_store_finegrained<<<g, b>>>(f, 7.f);
    if (m) plain_store_managed<<<g, b>>>(m, 8.f);
    float* big = nullptr;
    CK(hipMalloc(&big, (size_t)64 << 20));
    plain_store_16x_spaced_50us<<<g, b>>>(a, 9.f);
    plain_store_16x_spaced_400us<<<g, b>>>(a, 10.f);
    plain_store_16x_spaced_400us_beside_a_stream<<<dim3(512), b>>>(a, big, 11.f);
    rewrite16x_total_2MiB<<<g, b>>>(big, 1.f);
    rewrite16x_total_4MiB<<<g, b>>>(big, 1.f);
    rewrite16x_total_8MiB<<<g, b>>>(big, 1.f);
    rewrite16x_total_12MiB<<<g, b>>>(big, 1.f);
    rewrite16x_total_16MiB<<<g, b>>>(big, 1.f);
    rewrite16x_total_24MiB<<<g, b>>>(big, 1.f);
    CK(hipDeviceSynchronize());
    printf("done; per kernel: %d rewrites of %zu bytes\n", REPS, bytes);
    return 0;
}
