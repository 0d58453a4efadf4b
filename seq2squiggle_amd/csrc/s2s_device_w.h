// s2s_device_w.h -- the split-f16 FFT block of s2s_device_h.h in the 32x32x16 MFMA geometry.
//
// Same arithmetic (every operand split into two f16 halves, three MFMA products per product, fp32
// accumulation), different tiling: a wave owns ONE tile of 32 time columns and every MFMA is
// v_mfma_f32_32x32x16_f16.  Why: the attention loop of the 16x16x32 version is issue bound, and a
// 16x16x32 MFMA holds the SIMD's vector issue port for 8 of its 16 cycles while a 32x32x16 MFMA holds it
// for 8 of its 32 -- half the issue cycles per flop.  The layout also makes the softmax and the P.V
// bookkeeping lane-local:
//   * C/D layout: lane (h = lane>>5, c = lane&31) holds rows (r&3) + 8(r>>2) + 4h of column c in its 16
//     registers; registers 8b..8b+7 are directly the B operand of k-block b of the next MFMA;
//   * a column's 64 scores of a pass sit in 32 registers of 2 lanes: one v_permlane32_swap per reduction;
//   * Q^T needs no LDS re-layout: one permlane32_swap pair turns the [hi | lo] split of the 4 rows a lane
//     owns into the [Q_hi (h=0) | Q_lo (h=1)] operand of QK^T;
//   * P.V has 32 output rows: 0-7 V_hi d, 8-15 V_lo d, 16 = a row of ones (the softmax row sum comes out
//     with it), so O[d] = row d + row 8+d is an in-lane add of registers r and r+4.
//   * the running max is folded into QK^T itself: the K_lo.Q_lo term (2^-22 of the score) is dropped and
//     its k-slots carry [1, 1] x [-m_hi, -m_lo] instead (constant rows in LDS on the A side).
#pragma once
#include "s2s_device_h.h"

#ifndef S2S_W_HEAD_UNROLL
#define S2S_W_HEAD_UNROLL 4
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMAW(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct AttnLdsW {
    static constexpr int K_BYTES = 8 * 2 * 256 * 8 * 2;           // [head][hi|lo][key][8 d] halves
    static constexpr int VS = 264;                                // halves per V row
    static constexpr int V_BYTES = 8 * 16 * VS * 2;               // [head][16 rows: 0-7 hi d, 8-15 lo d][VS]
    static constexpr int CK_OFF = K_BYTES + V_BYTES;              // constant K rows [256][8]: {1, 1, 0, ...}
    static constexpr int CK_BYTES = 256 * 8 * 2;
    static constexpr int CV_OFF = CK_OFF + CK_BYTES + 8;          // constant V rows [16][VS]: row 0 ones, rest 0;
    static constexpr int CV_BYTES = 16 * VS * 2;                  //   (+8 B: bank-interleaves with the real rows)
    static constexpr int BYTES = CV_OFF + CV_BYTES + 8;
};

__device__ __forceinline__ void load_unit_w(f32x4 (&f)[8], const float* __restrict__ ws) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = ldg4(ws + i * 256);
}

// B operands of one 32-feature accumulator tile: k-block b = registers 8b .. 8b+7
__device__ __forceinline__ void split_tile(const f32x16 t, const float one, HL& b0, HL& b1) {
    unsigned hi[8], lo[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) split2(t[2 * i], t[2 * i + 1], one, hi[i], lo[i]);
    b0.hi = __builtin_bit_cast(h8, (uv4{hi[0], hi[1], hi[2], hi[3]}));
    b0.lo = __builtin_bit_cast(h8, (uv4{lo[0], lo[1], lo[2], lo[3]}));
    b1.hi = __builtin_bit_cast(h8, (uv4{hi[4], hi[5], hi[6], hi[7]}));
    b1.lo = __builtin_bit_cast(h8, (uv4{lo[4], lo[5], lo[6], lo[7]}));
}

// acc += W_unit * x: one 32-row m-tile, K = 64 as four k-blocks of 16, three products each.
// Unit (8 KiB): [kb0 hi][kb0 lo] ... [kb3 hi][kb3 lo], 16 B per lane each.
__device__ __forceinline__ void mm_unit_w(f32x16& acc, const f32x4 (&f)[8], const HL (&x)[4]) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const h8 wh = as_h8(f[2 * kb]), wl = as_h8(f[2 * kb + 1]);
        acc = MFMAW(wh, x[kb].hi, acc);
        acc = MFMAW(wh, x[kb].lo, acc);
        acc = MFMAW(wl, x[kb].hi, acc);
    }
}

// per-row vector (bias, LayerNorm gain ...) in accumulator layout: element r <- v[32T + 8(r>>2) + 4h + (r&3)]
__device__ __forceinline__ f32x16 rowvec(const float* __restrict__ v, int T, int h) {
    f32x16 o;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const f32x4 t = ldg4(v + 32 * T + 8 * rr + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[4 * rr + j] = t[j];
    }
    return o;
}

__device__ __forceinline__ float sum_h(float v) {           // over the two lane halves
    const unsigned u = __float_as_uint(v);
    auto t = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
__device__ __forceinline__ float max_h(float v) {
    const unsigned u = __float_as_uint(v);
    auto t = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}

__device__ __forceinline__ void layer_norm_w(f32x16 (&x)[2], const float* __restrict__ gam, const float* __restrict__ bet, int h) {
    float s = 0.0f;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += x[T][r];
    const float mean = sum_h(s) * (1.0f / 64.0f);
    float v = 0.0f;
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = x[T][r] - mean; v += d * d; }
    const float rstd = 1.0f / sqrtf(sum_h(v) * (1.0f / 64.0f) + 1e-5f);
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const f32x16 gm = rowvec(gam, T, h), bt = rowvec(bet, T, h);
#pragma unroll
        for (int r = 0; r < 16; ++r) x[T][r] = (x[T][r] - mean) * rstd * gm[r] + bt[r];
    }
}

// One FFTBlock (layers.py:116-142) for the decoder: 8 waves x 32 time columns = 256 >= TV = 250.
//   X: this wave's block input/output, two 32-feature tiles in accumulator layout.
template <int TV>
__device__ __forceinline__ void fft_block_w(const float* __restrict__ W, const LayerOff L, f32x16 (&X)[2],
                                            char* __restrict__ lds, int wave, int lane, const float one,
                                            unsigned long long* diag_buf = nullptr) {
    using G = AttnLdsW;
    constexpr int NT = 8;                          // key tiles of 32
    const int h = lane >> 5, c = lane & 31;
    DIAG_DECL;
    _Float16* __restrict__ Kl = reinterpret_cast<_Float16*>(lds);
    _Float16* __restrict__ Vl = reinterpret_cast<_Float16*>(lds + G::K_BYTES);
    const float* ws = W + L.stream_w + lane * 4;
    f32x4 fa[8], fb[8];
    load_unit_w(fa, ws); ws += 2048;               // Wk rows 0-31 (heads 0-3)

    HL xb[4];                                      // block input as B operands
    split_tile(X[0], one, xb[0], xb[1]);
    split_tile(X[1], one, xb[2], xb[3]);

    __syncthreads();                               // every wave is done reading the previous block's K/V
    DIAG_STAMP(0);
    if (wave == 0) {                               // constant operand rows (tiny; rewritten every block)
        uv4* ck = reinterpret_cast<uv4*>(lds + G::CK_OFF);
        for (int i = lane; i < 256; i += 64) ck[i] = uv4{0x3C003C00u, 0, 0, 0};           // {1.0h, 1.0h, 0 ...}
        _Float16* cv = reinterpret_cast<_Float16*>(lds + G::CV_OFF);
        for (int i = lane; i < 16 * G::VS; i += 64) cv[i] = (i < G::VS) ? (_Float16)1.0f : (_Float16)0.0f;
    }
    // ---- K^T and V^T of this wave's 32 time columns, all heads -> LDS as hi/lo halves (layers.py:74-78)
    const int key = 32 * wave + c;
#pragma unroll
    for (int u = 0; u < 4; ++u) {                  // units: Wk mt0, Wk mt1, Wv mt0, Wv mt1
        f32x16 a;
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = 0.0f;
        if (u & 1) { load_unit_w(fa, ws); ws += 2048; __builtin_amdgcn_sched_barrier(0); mm_unit_w(a, fb, xb); }
        else       { load_unit_w(fb, ws); ws += 2048; __builtin_amdgcn_sched_barrier(0); mm_unit_w(a, fa, xb); }
        __builtin_amdgcn_sched_barrier(0);
        const int mt = u & 1;
        a += rowvec(W + (u < 2 ? L.bk_nat : L.bv), mt, h);
#pragma unroll
        for (int hq = 0; hq < 4; ++hq) {           // registers 4hq..4hq+3 = head 4mt+hq, d = 4h + 0..3
            const int head = 4 * mt + hq;
            h4 hi, lo;
            split4(f32x4{a[4 * hq], a[4 * hq + 1], a[4 * hq + 2], a[4 * hq + 3]}, one, hi, lo);
            if (u < 2) {
                *reinterpret_cast<h4*>(Kl + ((head * 2 + 0) * 256 + key) * 8 + 4 * h) = hi;
                *reinterpret_cast<h4*>(Kl + ((head * 2 + 1) * 256 + key) * 8 + 4 * h) = lo;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Vl[(head * 16 + 4 * h + r) * G::VS + key] = hi[r];
                    Vl[(head * 16 + 8 + 4 * h + r) * G::VS + key] = lo[r];
                }
            }
        }
    }
    // after the loop: fa holds the unit after Wv mt1 = Wq rows 0-31
    // ---- fc accumulator starts as bias + residual (layers.py:85-86)
    f32x16 acc[2];
#pragma unroll
    for (int T = 0; T < 2; ++T) acc[T] = X[T] + rowvec(W + L.bfc, T, h);
    DIAG_STAMP(1);
    __syncthreads();                               // K/V of every wave visible
    DIAG_STAMP(2);

    const float c1 = 1.4426950408889634f * 0.35355339059327373f;     // log2(e) / sqrt(d_k = 8)
#pragma unroll 1
    for (int grp = 0; grp < 2; ++grp) {            // heads 4grp .. 4grp+3 = one 32-row tile of Wq
        f32x16 qa;
#pragma unroll
        for (int r = 0; r < 16; ++r) qa[r] = 0.0f;
        mm_unit_w(qa, fa, xb);
        __builtin_amdgcn_sched_barrier(0);
        qa = (qa + rowvec(W + L.bq_nat, grp, h)) * c1;               // scores come out in log2 units
        f32x16 oh;                                 // O^T of the 4 heads: registers 4hq + j = head hq, d 4h + j
#pragma unroll
        for (int r = 0; r < 16; ++r) oh[r] = 0.0f;
#pragma unroll S2S_W_HEAD_UNROLL
        for (int hq = 0; hq < 4; ++hq) {           // the current head's rows are rotated into qa[0..3]
            const int head = 4 * grp + hq;
            // [Q_hi (h = 0) | Q_lo (h = 1)]: this lane's 4 rows split, halves exchanged
            h4 qhi, qlo;
            split4(f32x4{qa[0], qa[1], qa[2], qa[3]}, one, qhi, qlo);
#pragma unroll
            for (int r = 0; r < 12; ++r) qa[r] = qa[r + 4];
            const uv2 uh = __builtin_bit_cast(uv2, qhi), ul = __builtin_bit_cast(uv2, qlo);
            auto s0 = __builtin_amdgcn_permlane32_swap(uh[0], ul[0], false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(uh[1], ul[1], false, false);
            const h8 b1 = __builtin_bit_cast(h8, (uv4{s0[0], s1[0], s0[1], s1[1]}));
            const _Float16* kp1 = Kl + ((head * 2 + 0) * 256 + c) * 8;                       // K_hi, both halves
            const _Float16* kp2 = h ? reinterpret_cast<const _Float16*>(lds + G::CK_OFF) + c * 8
                                    : Kl + ((head * 2 + 1) * 256 + c) * 8;                   // K_lo | constant {1,1,0..}
            const _Float16* vp = (c < 16) ? Vl + (head * 16 + c) * G::VS + 4 * h
                                          : reinterpret_cast<const _Float16*>(lds + G::CV_OFF) + (c - 16) * G::VS + 4 * h;
            const bool last_head = (hq == 3);
            f32x16 o;
            // ---- fast attempt: pass-0 max only; QK^T of tile kt+1 is issued before the exponentials of tile kt
            {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.0f;
                f32x16 sc[2];
                h8 b2 = h ? h8{0, 0, 0, 0, 0, 0, 0, 0} : b1;
                {
                    const h8 ka1 = *reinterpret_cast<const h8*>(kp1), ka2 = *reinterpret_cast<const h8*>(kp2);
#pragma unroll
                    for (int r = 0; r < 16; ++r) sc[0][r] = 0.0f;
                    sc[0] = MFMAW(ka1, b1, sc[0]);
                    sc[0] = MFMAW(ka2, b2, sc[0]);
                }
                float mh = sc[0][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mh = fmaxf(mh, sc[0][r]);
                const float m0 = max_h(mh);
                sc[0] -= m0;
                unsigned mhi, mlo;
                split2(-m0, 0.0f, one, mhi, mlo);
                b2 = h ? __builtin_bit_cast(h8, (uv4{(mhi & 0xFFFFu) | (mlo << 16), 0, 0, 0})) : b1;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt) {
                    h8 va[2];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const h4 v0 = *reinterpret_cast<const h4*>(vp + 32 * kt + 16 * b);
                        const h4 v1 = *reinterpret_cast<const h4*>(vp + 32 * kt + 16 * b + 8);
                        va[b] = h8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    }
                    if (kt + 1 < NT) {             // next tile's scores: on the matrix core while this tile exponentiates
                        const h8 ka1 = *reinterpret_cast<const h8*>(kp1 + 32 * (kt + 1) * 8);
                        const h8 ka2 = *reinterpret_cast<const h8*>(kp2 + 32 * (kt + 1) * 8);
#pragma unroll
                        for (int r = 0; r < 16; ++r) sc[(kt + 1) & 1][r] = 0.0f;
                        sc[(kt + 1) & 1] = MFMAW(ka1, b1, sc[(kt + 1) & 1]);
                        sc[(kt + 1) & 1] = MFMAW(ka2, b2, sc[(kt + 1) & 1]);
                    }
                    f32x16& sv = sc[kt & 1];
                    if (TV < 256 && kt == NT - 1) {                  // phantom keys -> -inf
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (32 * (NT - 1) + (r & 3) + 8 * (r >> 2) + 4 * h >= TV) sv[r] = -__builtin_inff();
                    }
                    HL P[2];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                        exp_split4(f32x4{sv[8 * b], sv[8 * b + 1], sv[8 * b + 2], sv[8 * b + 3]}, one, h0, h1, l0, l1);
                        exp_split4(f32x4{sv[8 * b + 4], sv[8 * b + 5], sv[8 * b + 6], sv[8 * b + 7]}, one, h2, h3, l2, l3);
                        P[b].hi = __builtin_bit_cast(h8, (uv4{h0, h1, h2, h3}));
                        P[b].lo = __builtin_bit_cast(h8, (uv4{l0, l1, l2, l3}));
                    }
                    if (kt == NT - 1 && last_head) {
                        // the group's last P.V covers the latency of the next two units: Wfc columns of this
                        // group, then Wq of the next group (after the last: W1 unit 0)
                        __builtin_amdgcn_sched_barrier(0);
                        load_unit_w(fb, ws); load_unit_w(fa, ws + 2048);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        o = MFMAW(va[b], P[b].hi, o);                // rows 0-7 V_hi.P, 8-15 V_lo.P, 16 sum(P)
                        o = MFMAW(va[b], P[b].lo, o);
                    }
                }
            }
            // ---- an inf / NaN row sum means some P_hi left the f16 range: redo this head with the running max
            //      raised (and the sums rescaled) in every pass.  Rare; not pipelined.
            if (S2S_ALWAYS_RESCALE || __any(!(sum_h(h ? 0.0f : o[8]) <= 3.0e38f))) {
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = 0.0f;
                h8 b2 = h ? h8{0, 0, 0, 0, 0, 0, 0, 0} : b1;
                float m = 0.0f;
#pragma unroll 1
                for (int kt = 0; kt < NT; ++kt) {
                    const h8 ka1 = *reinterpret_cast<const h8*>(kp1 + 32 * kt * 8);
                    const h8 ka2 = *reinterpret_cast<const h8*>(kp2 + 32 * kt * 8);
                    f32x16 sv;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sv[r] = 0.0f;
                    sv = MFMAW(ka1, b1, sv);
                    sv = MFMAW(ka2, b2, sv);
                    if (TV < 256 && kt == NT - 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (32 * (NT - 1) + (r & 3) + 8 * (r >> 2) + 4 * h >= TV) sv[r] = -__builtin_inff();
                    }
                    float mh = sv[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mh = fmaxf(mh, sv[r]);
                    mh = max_h(mh);
                    const float delta = (kt == 0) ? mh : fmaxf(mh, 0.0f);
                    o *= __builtin_amdgcn_exp2f((kt == 0) ? 0.0f : -delta);
                    m += delta;
                    sv -= delta;
                    unsigned mhi, mlo;
                    split2(-m, 0.0f, one, mhi, mlo);
                    b2 = h ? __builtin_bit_cast(h8, (uv4{(mhi & 0xFFFFu) | (mlo << 16), 0, 0, 0})) : b1;
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
                        exp_split4(f32x4{sv[8 * b], sv[8 * b + 1], sv[8 * b + 2], sv[8 * b + 3]}, one, h0, h1, l0, l1);
                        exp_split4(f32x4{sv[8 * b + 4], sv[8 * b + 5], sv[8 * b + 6], sv[8 * b + 7]}, one, h2, h3, l2, l3);
                        const h4 v0 = *reinterpret_cast<const h4*>(vp + 32 * kt + 16 * b);
                        const h4 v1 = *reinterpret_cast<const h4*>(vp + 32 * kt + 16 * b + 8);
                        const h8 va = h8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        o = MFMAW(va, __builtin_bit_cast(h8, (uv4{h0, h1, h2, h3})), o);
                        o = MFMAW(va, __builtin_bit_cast(h8, (uv4{l0, l1, l2, l3})), o);
                    }
                }
            }
            const float inv = 1.0f / sum_h(h ? 0.0f : o[8]);         // row 16 lives in the lower half's register 8
            // rotate the result in: after 4 heads registers 4hq + j hold head hq
#pragma unroll
            for (int r = 0; r < 12; ++r) oh[r] = oh[r + 4];
#pragma unroll
            for (int j = 0; j < 4; ++j) oh[12 + j] = (o[j] + o[j + 4]) * inv;
        }
        ws += 4096;                                // the two units requested during the last head's last P.V
        // ---- fc, k-blocks 2grp and 2grp+1 (head pairs): acc += Wfc[:, 32grp : 32grp+32] * O^T
        HL ob[2];                                  // k-block p = heads 2p, 2p+1 = registers 8p .. 8p+7 of oh
        split_tile(oh, one, ob[0], ob[1]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)             // unit = [mt0: kb a hi, lo, kb b hi, lo][mt1: ...]
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const h8 wh = as_h8(fb[4 * mt + 2 * p]), wl = as_h8(fb[4 * mt + 2 * p + 1]);
                acc[mt] = MFMAW(wh, ob[p].hi, acc[mt]);
                acc[mt] = MFMAW(wh, ob[p].lo, acc[mt]);
                acc[mt] = MFMAW(wl, ob[p].hi, acc[mt]);
            }
    }
    DIAG_STAMP(3);
    // ---- FFN 64 -> 256 -> 64 in four 64-wide slices (layers.py:108-113); fa = W1 unit 0, fb requested next
    load_unit_w(fb, ws); ws += 2048;
    __builtin_amdgcn_sched_barrier(0);
    layer_norm_w(acc, W + L.ln1g, W + L.ln1b, h);                    // acc = x1
    DIAG_STAMP(4);
    HL x1b[4];
    split_tile(acc[0], one, x1b[0], x1b[1]);
    split_tile(acc[1], one, x1b[2], x1b[3]);
#pragma unroll
    for (int T = 0; T < 2; ++T) X[T] = acc[T] + rowvec(W + L.b2, T, h);   // X = bias + residual accumulator
#pragma unroll 1
    for (int hc = 0; hc < 4; ++hc) {
        f32x16 hid[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {           // W1 units: rows 64hc + 32mt ..
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = 0.0f;
            if (mt & 1) { mm_unit_w(t, fb, x1b); __builtin_amdgcn_sched_barrier(0); load_unit_w(fb, ws); ws += 2048; }
            else        { mm_unit_w(t, fa, x1b); __builtin_amdgcn_sched_barrier(0); load_unit_w(fa, ws); ws += 2048; }
            __builtin_amdgcn_sched_barrier(0);
            const f32x16 b1v = rowvec(W + L.b1 + 64 * hc, mt, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) hid[mt][r] = fmaxf(t[r] + b1v[r], 0.0f);
        }
        HL hb[4];
        split_tile(hid[0], one, hb[0], hb[1]);
        split_tile(hid[1], one, hb[2], hb[3]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {           // W2 units: rows 32mt .., columns 64hc ..
            if (mt & 1) { mm_unit_w(X[mt], fb, hb); __builtin_amdgcn_sched_barrier(0); load_unit_w(fb, ws); ws += 2048; }
            else        { mm_unit_w(X[mt], fa, hb); __builtin_amdgcn_sched_barrier(0); load_unit_w(fa, ws); ws += 2048; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    DIAG_STAMP(5);
    layer_norm_w(X, W + L.ln2g, W + L.ln2b, h);
    DIAG_STAMP(6);
}
