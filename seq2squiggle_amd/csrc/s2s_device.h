// s2s_device.h -- device-side building blocks of the predict path (gfx950 / CDNA4 only).
//
// Everything runs "transposed": an activation tile lives in registers as X^T[feature][time]
// in the C/D layout of v_mfma_f32_16x16x4_f32 (lane = (g = lane>>4, c = lane&15); register r of
// feature-tile ft holds feature 16*ft + 4*g + r of time column c).  In that layout an
// accumulator is directly the B operand of the next MFMA (k-step (ft, r) supplies k-index
// 16*ft + 4*g + r), so a whole FFT block runs register-to-register; weights are the A operand,
// pre-packed on the host into fragment order ([m-tile][k-tile][lane][4], 1 KiB per
// wave-load).  Only K and V cross waves, through LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define S2S_MAX_LAYERS 4

struct LayerOff {
    int stream;                             // packed A fragments in consumption order (float offset into the arena)
    int stream_h;                           // the same units as f16 hi/lo fragments (s2s_device_h.h)
    int stream_f;                           // the hi fragments of stream_h only (S2S_MODE_F16)
    int bq, bk, bv, bfc, b1, b2;            // biases (bq/bk in the q/k row permutation)
    int bq_nat, bk_nat;                     // bq/bk in natural row order (f16 block)
    int ln1g, ln1b, ln2g, ln2b;
};
struct MlpOff { int w0, b0, w3, b3, w0h; };   // w0h: the 64x64 layer as 4 f16 hi/lo units
struct ModelDev {
    int k, enc_layers, dec_layers, pre_layers;
    float scale;                            // scaling_max_value
    int pe_enc, pe_dec;                     // [T][64] natural
    int emb_wt, emb_b;                      // W_emb^T [5k][64], bias [64]
    int pre_w[S2S_MAX_LAYERS], pre_b[S2S_MAX_LAYERS], pre_wh[S2S_MAX_LAYERS];
    LayerOff enc[S2S_MAX_LAYERS], dec[S2S_MAX_LAYERS];
    MlpOff noise, conc, rate;
    int out_w, out_b;
};
struct ParamsDev {
    float dwell_mean, dwell_std, noise_std, min_noise, min_duration;
    int noise_sampling, duration_sampling;
    unsigned seed_lo, seed_hi;
};
struct DebugDev {
    float *emb_out, *enc_out, *sigma, *conc, *rate, *g, *y_scaled, *z01;
    unsigned long long* diag;       // S2S_DIAG / S2S_TILEHIST builds only
    unsigned long long* stats;      // every build: S2S_STAT_* counters of the handle (s2s_stats_read)
    const float *emb_in, *dec_in;   // stand-alone sub-module operators (TEST instance): stage INPUTS taken from memory
};

// ReLU as an integer max on the bit pattern (v_max_i32: negative floats have the sign bit set, i.e. are negative integers):
// one VOP2 instruction.  fmaxf(x, 0.0f) compiles to TWO -- the IEEE maxNum semantics make the compiler quiet a possible
// signalling NaN first (v_max_f32 x, x, x), which it cannot rule out for a value that comes out of an MFMA: 590 such
// instructions in the f16x3 kernel, 256 of them per wave and chunk in the FFN loops.  Same bits for every non-NaN input.
__device__ __forceinline__ float relu1(const float x) {
#ifdef S2S_RELU_FMAX
    return fmaxf(x, 0.0f);
#else
    const int i = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
#endif
}

#ifdef S2S_DIAG
// diagnostic build only: per-phase wave-cycle sums (s_memtime = shader-clock cycles).  Every wave adds into ITS OWN row of a small
// LDS table (one lane, plain read-modify-write: no contention, no global traffic in the hot loop -- round 2's version sent a
// global atomicAdd per stamp from 2,048 waves to 16 hot addresses, which slowed the diagnostic build 5x and inflated the phases
// that stamp most often); the kernel folds the table into the global side buffer once, when it ends.
#define S2S_DIAG_SLOTS 48
__shared__ unsigned long long s2s_diag_lds[8 * S2S_DIAG_SLOTS];
#define DIAG_DECL unsigned long long diag_t_ = __builtin_readcyclecounter()
#define DIAG_STAMP(slot)                                                            \
    do {                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                          \
        const unsigned long long n_ = __builtin_readcyclecounter();                 \
        if ((threadIdx.x & 63) == 0) s2s_diag_lds[(threadIdx.x >> 6) * S2S_DIAG_SLOTS + (slot)] += n_ - diag_t_; \
        diag_t_ = n_;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                          \
    } while (0)
#define DIAG_COUNT(slot, n) do { if ((threadIdx.x & 63) == 0) s2s_diag_lds[(threadIdx.x >> 6) * S2S_DIAG_SLOTS + (slot)] += (n); } while (0)
#elif defined(S2S_STAMP_SB)
#define DIAG_DECL
#define DIAG_STAMP(slot) __builtin_amdgcn_sched_barrier(0)
#define DIAG_COUNT(slot, n)
#else
#define DIAG_DECL
#define DIAG_STAMP(slot)
#define DIAG_COUNT(slot, n)
#endif

// Production counters (every build, s2s_stats_read): a handful of LDS words per workgroup -- the heads whose fast softmax
// overflowed and were redone on the safe path (counted INSIDE that rare branch: nothing on the fast path), and thread 0's
// s_memtime / s_memrealtime at kernel entry -- folded into the handle's side buffer by one thread per workgroup when the kernel
// ends (256 global atomics per launch).
#define S2S_STAT_REDO 0          // (wave, head, layer) softmax runs that took the safe path
#define S2S_STAT_CYCLES 1        // shader-clock cycles (s_memtime) of thread 0, summed over the workgroups
#define S2S_STAT_TICKS 2         // 100 MHz ticks (s_memrealtime) over the same spans
#define S2S_STAT_WGS 3           // workgroups summed
// (256 bytes, not 48: the kernel's dynamic LDS follows this array, and every bank-slot placement in s2s_device_h.h -- V^T row r on
//  16-byte slot r mod 16, the zeros rows beside them -- assumes that its base is a multiple of 256 bytes, i.e. of the 64 banks;
//  with a 48-byte array in front SQ_LDS_BANK_CONFLICT went from 4.3 k back to 8.4 k cycles per chunk (profiles/r04/lds_conflicts.txt;
//  the wall clock was the same: 188.6 k shader cycles at 2.195 GHz against 190.9 k at 2.215 GHz -- the chip is power-limited))
__shared__ __attribute__((aligned(256))) unsigned s2s_stats_lds[64];      // [0] redo, [4..7] entry stamps

#ifdef S2S_TILEHIST
// diagnostic build: histogram of the largest shifted score (log2 units below the row's pass-0 maximum) per attention tile,
// [decoder layer 0|1][32-key tile | 16-key step][64 bins of one unit; bin 63: above the pass-0 maximum]
__shared__ unsigned s2s_hist_lds[2 * 2 * 64];
#endif
// static LDS in front of the kernel's dynamic region (the workgroup's budget is their sum: Fused<MODE>::LDS asserts it)
constexpr int S2S_STATIC_LDS_BYTES = 256
#ifdef S2S_DIAG
    + 8 * S2S_DIAG_SLOTS * 8
#endif
#ifdef S2S_TILEHIST
    + 2 * 2 * 64 * 4
#endif
    ;

#ifndef S2S_ABL
#define S2S_ABL 0      // timing-only ablations (tools/ablate.py); results are garbage when non-zero
#endif
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
// Output streams (signal, dwell counts) are stored past the XCD's L2 (global_store ... sc1: the line is not kept), so that they
// do not push the frontend -> decoder hand-off slots and the weights out of it.
__device__ __forceinline__ void store_stream(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_stream(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Reductions over the 4 lane groups (lanes c, c+16, c+32, c+48) with the gfx950 row/half swaps
// (v_permlane16_swap: odd rows of a <-> even rows of b; v_permlane32_swap: upper half of a <-> lower
// half of b): pure VALU, no LDS round trip.
__device__ __forceinline__ float sum_g(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const unsigned w = __float_as_uint(v);
    auto t = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float max_g(float v) {
    const unsigned u = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const unsigned w = __float_as_uint(v);
    auto t = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: the draw for (chunk, position, kind) never
// depends on batch size, launch geometry or GPU count.
struct u32x4 { unsigned x, y, z, w; };
__device__ __forceinline__ u32x4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                               unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}
__device__ __forceinline__ float u01_open0(unsigned x) { return (float)((x >> 8) + 1u) * 5.9604644775390625e-08f; }  // (0,1]
__device__ __forceinline__ float u01_open1(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }        // [0,1)
// Box-Muller on the hardware transcendentals: sqrt(-2 ln u) = sqrt(-2 ln2 * v_log_f32(u)); v_cos_f32 takes its argument in
// revolutions, so cos(2 pi v) needs no range reduction.  (About 12 instructions instead of ~100 for the libm forms; their
// 1-ulp-class errors are far below anything a distribution test of N(0,1) can see.)
__device__ __forceinline__ float box_muller(unsigned a, unsigned b) {
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u01_open0(a)));
    return r * __builtin_amdgcn_cosf(u01_open1(b));
}
enum { S2S_KIND_GAMMA = 1, S2S_KIND_DWELL = 2, S2S_KIND_NOISE = 3 };

// torch._standard_gamma (ATen/native/Distributions.h sample_gamma: Marsaglia-Tsang 2000 with the
// alpha < 1 boost), drawn from Philox; `iter` bounds the rejection loop (acceptance > 95 %).
__device__ float standard_gamma(float alpha, unsigned chunk_lo, unsigned chunk_hi, unsigned pos,
                                unsigned k0, unsigned k1) {
    float scale = 1.0f;
    unsigned draw = 0;
    if (alpha < 1.0f) {
        if (alpha == 0.0f) return 0.0f;
        const u32x4 r = philox4x32_10(chunk_lo, chunk_hi, pos | (S2S_KIND_GAMMA << 16), draw++, k0, k1);
        scale *= powf(1.0f - u01_open1(r.x), 1.0f / alpha);
        alpha += 1.0f;
    }
    const float d = alpha - 1.0f / 3.0f;
    const float c = 1.0f / sqrtf(9.0f * d);
    float v = 1.0f;
    for (int iter = 0; iter < 256; ++iter) {
        const u32x4 r = philox4x32_10(chunk_lo, chunk_hi, pos | (S2S_KIND_GAMMA << 16), draw++, k0, k1);
        const float x = box_muller(r.x, r.y);
        const float y = 1.0f + c * x;
        if (y <= 0.0f) continue;
        v = y * y * y;
        const float u = 1.0f - u01_open1(r.z);
        const float xx = x * x;
        if (u < 1.0f - 0.0331f * xx * xx) break;
        if (logf(u) < 0.5f * xx + d * (1.0f - v + logf(v))) break;
    }
    return scale * d * v;
}

// a * b + c with the product rounded before the sum, as torch evaluates `z * std + mean` (two kernels).  The __f*_rn
// intrinsics are plain operators once inlined and hipcc's default -ffp-contract=fast fuses them into one v_fma.
__device__ __forceinline__ float mul_then_add(float a, float b, float c) {
#pragma clang fp contract(off)
    const float p = a * b;
    return p + c;
}

__device__ __forceinline__ float softplus_t(float x) {      // nn.Softplus(beta=1, threshold=20)
    return x > 20.0f ? x : log1pf(expf(x));
}

// ---------------------------------------------------------------------------------------------
// LDS geometry of one attention block: K as float2 planes [head][d-pair g][key], V^T rows
// [pair][16 rows][key], a zero strip for the masked half of the PV A-operand.
template <int NKT> struct AttnLds {
    static constexpr int KEYS = 16 * NKT;
    static constexpr int KPS = ((KEYS % 32) == 16) ? KEYS : KEYS + 16;   // float2 per plane, == 16 mod 32
    static constexpr int RS = KEYS + 4;                                  // floats per V row, == 4 mod 8
    static constexpr int K_FLOATS = 8 * 4 * KPS * 2;
    static constexpr int V_FLOATS = 4 * 16 * RS;
    static constexpr int Z_FLOATS = KEYS;
    static constexpr int FLOATS = K_FLOATS + V_FLOATS + Z_FLOATS;
    static constexpr int BYTES = FLOATS * 4;
};

// acc[q][mt] (+)= W[mt-th 16 rows] * X[q]  for MT m-tiles; W packed [mt][KT][64][4].
template <int NQ, int MT, int KT>
__device__ __forceinline__ void gemm_acc(const float* __restrict__ wp, int lane, f32x4 (&acc)[NQ][MT],
                                         const f32x4 (&x)[NQ][KT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const f32x4 a = ldg4(wp + ((mt * KT + kt) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q][mt] = MFMA4(a[r], x[q][kt][r], acc[q][mt]);
            }
        }
    }
}

// nn.LayerNorm(64, eps=1e-5) over the feature axis (registers + the 4 lane groups), in place; gm/bt: this lane's
// slices of weight and bias.  HW: 1/sqrt on the hardware transcendental (v_rsq_f32, 1 ulp) instead of the correctly rounded
// sqrt + division sequences (about 25 instructions) -- the f16x3 blocks, whose operands carry 22 bits anyway.
template <int NQ, bool HW = false>
__device__ __forceinline__ void layer_norm64_r(f32x4 (&x)[NQ][4], const f32x4 (&gm)[4], const f32x4 (&bt)[4]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float s = 0.0f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += x[q][ft][r];
        const float mean = sum_g(s) * (1.0f / 64.0f);
        float v = 0.0f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = x[q][ft][r] - mean; v += d * d; }
        const float var = sum_g(v) * (1.0f / 64.0f) + 1e-5f;
        const float rstd = HW ? __builtin_amdgcn_rsqf(var) : 1.0f / sqrtf(var);
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[q][ft][r] = (x[q][ft][r] - mean) * rstd * gm[ft][r] + bt[ft][r];
    }
}
template <int NQ, bool HW = false>
__device__ __forceinline__ void layer_norm64(f32x4 (&x)[NQ][4], const float* __restrict__ gam,
                                             const float* __restrict__ bet, int g) {
    f32x4 gm[4], bt[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) { gm[ft] = ldg4(gam + 16 * ft + 4 * g); bt[ft] = ldg4(bet + 16 * ft + 4 * g); }
    layer_norm64_r<NQ, HW>(x, gm, bt);
}

// Barrier between the phases of a block that exchange K/V through LDS.  SOLO: the block's sequence is one time tile owned
// by ONE wave (the f32 encoder), so the exchange stays inside the wave -- LDS executes a wave's accesses in issue order, only the
// compiler has to keep them in program order -- and the block may run in some waves of a larger workgroup and not in others.
template <bool SOLO>
__device__ __forceinline__ void block_sync() {
    if constexpr (SOLO) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// One weight "unit" = 4 A fragments (4 KiB) = what one 16-row m-tile of a K=64 GEMM consumes.
// A layer's units are stored in the order the block consumes them (host: pack_layer), so the
// stream pointer just advances and the next unit is always requested one unit ahead of its use.
__device__ __forceinline__ void load_unit(f32x4 (&f)[4], const float* __restrict__ ws) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = ldg4(ws + i * 256);
}
// acc[q] += W_unit * x[q]   (one m-tile, k-tiles 0..3)
template <int NQ>
__device__ __forceinline__ void mm_unit(f32x4 (&acc)[NQ], const f32x4 (&f)[4], const f32x4 (&x)[NQ][4]) {
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = MFMA4(f[kt][r], x[q][kt][r], acc[q]);
}

// One FFTBlock (layers.py:116-142): post-LN multi-head attention (layers.py:44-88, 11-41; no
// mask in predict, model.py:217) + position-wise FFN (layers.py:91-113), eval mode.
//   X    : this wave's NQ time tiles of the block input, replaced by the block output;
//   qt0  : index of the wave's first time tile inside the workgroup's sequence;
//   NKT  : key tiles of the whole sequence (all waves of the workgroup together);
//   TV   : number of real keys (keys >= TV are phantom padding and get probability 0).
template <int NQ, int NKT, int TV>
__device__ __forceinline__ void fft_block(const float* __restrict__ W, const LayerOff L, f32x4 (&X)[NQ][4],
                                          float* __restrict__ lds, int qt0, int lane,
                                          [[maybe_unused]] unsigned long long* diag_buf = nullptr) {
    using G = AttnLds<NKT>;
    DIAG_DECL;
    const int g = lane >> 4, c = lane & 15;
    f32x2* __restrict__ Kl = reinterpret_cast<f32x2*>(lds);
    float* __restrict__ Vl = lds + G::K_FLOATS;
    const float* __restrict__ Zl = Vl + G::V_FLOATS;
    const float* ws = W + L.stream + lane * 4;       // weight stream, this lane's slice of each fragment
    f32x4 fa[4], fb[4];                               // ping-pong unit buffers

    load_unit(fa, ws); ws += 1024;                    // Wk, pair 0
    block_sync<NKT == 1>();
    DIAG_STAMP(0);                                  // every wave is done reading the previous block's K/V
    // ---- K^T and V^T of this wave's time tiles, all heads -> LDS (layers.py:74-78)
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        load_unit(fb, ws); ws += 1024;                // Wv, pair p
        const f32x4 bk = ldg4(W + L.bk + 16 * p + 4 * g), bv = ldg4(W + L.bv + 16 * p + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 ak[NQ], av[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { ak[q] = bk; av[q] = bv; }
        mm_unit<NQ>(ak, fa, X);
        __builtin_amdgcn_sched_barrier(0);
        load_unit(fa, ws); ws += 1024;                // Wk, pair p+1 (after the last pair: Wq, pair 0)
        __builtin_amdgcn_sched_barrier(0);
        mm_unit<NQ>(av, fb, X);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (S2S_ABL & 8) { asm volatile("" ::"v"(ak[q]), "v"(av[q])); continue; }
            const int key = 16 * (qt0 + q) + c;
            // rows of the packed Wk tile are permuted so that registers {0,1} are head 2p
            // (d = 2g, 2g+1) and registers {2,3} head 2p+1: exactly the float2 the S MFMA reads
            Kl[((2 * p + 0) * 4 + g) * G::KPS + key] = f32x2{ak[q][0], ak[q][1]};
            Kl[((2 * p + 1) * 4 + g) * G::KPS + key] = f32x2{ak[q][2], ak[q][3]};
#pragma unroll
            for (int r = 0; r < 4; ++r) Vl[(p * 16 + 4 * g + r) * G::RS + key] = av[q][r];
        }
    }
    if (qt0 == 0 && lane < G::Z_FLOATS / 4)      // zero strip (<= 64 float4)
        reinterpret_cast<f32x4*>(lds + G::K_FLOATS + G::V_FLOATS)[lane] = f32x4{0, 0, 0, 0};

    // ---- fc accumulator starts as bias + residual (layers.py:85-86)
    f32x4 acc[NQ][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b = ldg4(W + L.bfc + 16 * mt + 4 * g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q][mt] = X[q][mt] + b;
    }
    DIAG_STAMP(1);
    if (!(S2S_ABL & 4)) block_sync<NKT == 1>();   // K/V of every wave visible
    DIAG_STAMP(2);

    const float c1 = 1.4426950408889634f * 0.35355339059327373f;     // log2(e) / sqrt(d_k = 8)
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        const f32x4 bq = ldg4(W + L.bq + 16 * p + 4 * g);
        // Q^T of head pair p for this wave's time tiles (same row permutation as K); fa = Wq rows of p
        f32x4 qp[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) qp[q] = bq;
        mm_unit<NQ>(qp, fa, X);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 op[NQ];                      // O^T of the pair: rows 0-7 head 2p, rows 8-15 head 2p+1
#pragma unroll
        for (int q = 0; q < NQ; ++q) op[q] = f32x4{0, 0, 0, 0};
        // Keys are processed in NH passes of HK tiles with a running max (flash-style).  Per pass the K and
        // V^T operand fragments are read from LDS once, up front, and serve all NQ time tiles; o0/o1 are
        // two accumulation chains so that no MFMA waits on the one issued just before it.
        constexpr int NH = (NKT >= 16) ? 2 : 1;
        constexpr int HK = NKT / NH;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const f32x2* kp = Kl + ((2 * p + hh) * 4 + g) * G::KPS + c;
            const float* vp = ((c >> 3) == hh) ? (Vl + (p * 16 + c) * G::RS + 4 * g) : (Zl + 4 * g);
            f32x4 o0[NQ], o1[NQ];
            float m[NQ], l[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { o0[q] = f32x4{0, 0, 0, 0}; o1[q] = f32x4{0, 0, 0, 0}; m[q] = -__builtin_inff(); l[q] = 0.0f; }
#pragma unroll
            for (int h2 = 0; h2 < NH; ++h2) {
                f32x2 ka[HK];
                f32x4 va[HK];
#pragma unroll
                for (int kt = 0; kt < HK; ++kt) ka[kt] = (S2S_ABL & 2) ? f32x2{qp[0][0], qp[0][1]} : kp[16 * (h2 * HK + kt)];
#pragma unroll
                for (int kt = 0; kt < HK; ++kt)
                    va[kt] = (S2S_ABL & 2) ? qp[0] : *reinterpret_cast<const f32x4*>(vp + 16 * (h2 * HK + kt));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const float q0 = qp[q][2 * hh], q1 = qp[q][2 * hh + 1];
                    f32x4 s[HK];
                    if (HK == 1) {
                        s[0] = MFMA4(ka[0][0], q0, (f32x4{0, 0, 0, 0}));
                        s[0] = MFMA4(ka[0][1], q1, s[0]);
                    } else {
#pragma unroll
                        for (int kt = 0; kt + 1 < HK; kt += 2) {   // two tiles interleaved: no back-to-back dependent MFMA
                            s[kt] = MFMA4(ka[kt][0], q0, (f32x4{0, 0, 0, 0}));
                            s[kt + 1] = MFMA4(ka[kt + 1][0], q0, (f32x4{0, 0, 0, 0}));
                            s[kt] = MFMA4(ka[kt][1], q1, s[kt]);
                            s[kt + 1] = MFMA4(ka[kt + 1][1], q1, s[kt + 1]);
                        }
                    }
                    if (TV < 16 * NKT && h2 == NH - 1) {   // phantom keys -> -inf (only the last key tile has any)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * (NKT - 1) + 4 * g + r >= TV) s[HK - 1][r] = -__builtin_inff();
                    }
                    float mh = s[0][0];
                    if (!(S2S_ABL & 1)) {
#pragma unroll
                        for (int kt = 0; kt < HK; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) mh = fmaxf(mh, s[kt][r]);
                    }
                    const float mn = (S2S_ABL & 1) ? mh : fmaxf(m[q], max_g(mh));
                    if (h2 > 0 && !(S2S_ABL & 1)) {        // rescale what was accumulated against the old max
                        const float alpha = __builtin_amdgcn_exp2f((m[q] - mn) * c1);
                        l[q] *= alpha;
                        o0[q] *= alpha; o1[q] *= alpha;
                    }
                    m[q] = mn;
                    const float mc = -mn * c1;
                    float lh = 1.0f;
                    if (!(S2S_ABL & 1)) {
                        lh = 0.0f;
#pragma unroll
                        for (int kt = 0; kt < HK; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c1, mc));
                                s[kt][r] = e;
                                lh += e;
                            }
                    }
                    l[q] += lh;
                    if (hh == 1 && q == NQ - 1 && h2 == NH - 1) {
                        // the pair's last P.V covers the latency of the next two weight units:
                        // Wfc columns of pair p, then Wq rows of pair p+1 (after the last pair: W1 unit 0)
                        __builtin_amdgcn_sched_barrier(0);
                        load_unit(fb, ws); load_unit(fa, ws + 1024); ws += 2048;
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int kt = 0; kt < HK; ++kt) {
                        o0[q] = MFMA4(va[kt][0], s[kt][0], o0[q]);
                        o1[q] = MFMA4(va[kt][1], s[kt][1], o1[q]);
                        o0[q] = MFMA4(va[kt][2], s[kt][2], o0[q]);
                        o1[q] = MFMA4(va[kt][3], s[kt][3], o1[q]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const float inv = 1.0f / sum_g(l[q]);
                op[q] += (o0[q] + o1[q]) * inv;  // the other head's rows are exact zeros
            }
        }
        // fc: acc += Wfc[:, 16p : 16p+16] * O_pair^T   (unit = the 4 m-tiles of k-tile p)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q][mt] = MFMA4(fb[mt][r], op[q][r], acc[q][mt]);
    }
    DIAG_STAMP(3);
    if (!(S2S_ABL & 16)) layer_norm64<NQ>(acc, W + L.ln1g, W + L.ln1b, g);                // acc = x1
    DIAG_STAMP(4);

    // ---- FFN 64 -> 256 -> 64 in four 64-wide slices of the hidden layer (layers.py:108-113)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b = ldg4(W + L.b2 + 16 * mt + 4 * g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) X[q][mt] = acc[q][mt] + b;      // X = bias + residual accumulator
    }
#pragma unroll 1
    for (int hc = 0; hc < 4; ++hc) {
        f32x4 hid[NQ][4];
        f32x4 b1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) b1[mt] = ldg4(W + L.b1 + 64 * hc + 16 * mt + 4 * g);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {              // W1 units: rows 64hc + 16mt ..
            f32x4 t[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[q] = b1[mt];
            if (mt & 1) { load_unit(fa, ws); ws += 1024; __builtin_amdgcn_sched_barrier(0); mm_unit<NQ>(t, fb, acc); }
            else        { load_unit(fb, ws); ws += 1024; __builtin_amdgcn_sched_barrier(0); mm_unit<NQ>(t, fa, acc); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) hid[q][mt][r] = relu1(t[q][r]);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {              // W2 units: rows 16mt .., columns 64hc ..
            f32x4 t[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[q] = X[q][mt];
            if (mt & 1) { load_unit(fa, ws); ws += 1024; __builtin_amdgcn_sched_barrier(0); mm_unit<NQ>(t, fb, hid); }
            else        { load_unit(fb, ws); ws += 1024; __builtin_amdgcn_sched_barrier(0); mm_unit<NQ>(t, fa, hid); }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) X[q][mt] = t[q];
        }
    }
    DIAG_STAMP(5);
    if (!(S2S_ABL & 16)) layer_norm64<NQ>(X, W + L.ln2g, W + L.ln2b, g);
    DIAG_STAMP(6);
}
