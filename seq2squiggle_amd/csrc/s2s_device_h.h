// s2s_device_h.h -- the FFT block on the f16 matrix cores with fp32-class accuracy.
//
// The f32-input MFMA (s2s_device.h) runs on the same FMA lanes as the vector ALU: softmax VALU
// work and matrix work serialise and the block tops out near 57 % of the f32 matrix rate.  Here
// every product a*b is evaluated as  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  with a = a_hi + a_lo split
// into two f16 values (22 significant bits), on v_mfma_f32_16x16x32_f16 with fp32 accumulation:
// products of two f16 are exact in fp32, so the only errors are the 2^-22 representation error of
// each operand and the dropped lo*lo term.  Measured end to end (tests/test_gpu_parity.py) the
// signal stays within the same 1e-4 pA MAE bound as the f32 path.  The matrix cores run beside the
// vector ALU, which is left with the softmax, the hi/lo splits and the LayerNorms.
//
// Data layout is the same "transposed" scheme as s2s_device.h: activations are B operands, taken
// straight from accumulator registers.  For the K = 32 MFMA the B operand of k-block kb is built
// from the two 16-row accumulator tiles 2kb and 2kb+1 (element j of lane (g, c): tile 2kb + (j>>2),
// register j&3, i.e. feature 32kb + 16(j>>2) + 4g + (j&3)); the host packs the weights' A fragments
// in that same k order.
#pragma once
#include "s2s_device.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// Fast softmax path: the shift is the pass-0 max plus this many log2 units, so the f16 window of P_hi is 2^18 above
// the pass-0 max instead of 2^16 (fewer safe-path redos).  The largest P is then >= 2^-2, whose lo half is still a normal
// f16: measured distance to the fp64 oracle is unchanged up to 4 and grows from 6 (tools/cmp_mae.sh).
#ifndef S2S_SHIFT_BIAS
#define S2S_SHIFT_BIAS 2.0f
#endif
#ifndef S2S_ALWAYS_RESCALE
#define S2S_ALWAYS_RESCALE 0
#endif
#ifndef S2S_FFN_LDS
#define S2S_FFN_LDS 1           // decoder FFN weights staged once per workgroup in the dead K/V region (0: every wave streams them from L2)
#endif
#define S2S_PF_FLOATS (1024 + 16 + 16)   // one frontend -> decoder hand-off slot (s2s_hip.hip: S2S_SLOT_FLOATS)
#define S2S_Z2_FLOATS (256 + 64)         // a second all-zeros V^T row (264 halves) + room to start it on 16-byte bank slot 4: see vp in fft_block_h
#define S2S_PROG_INTS 16                 // per-wave progress counters of the attention loop (prio_balance), behind the small vectors
#define S2S_SV_FLOATS 960                // bq_nat, bk_nat, bq, bk, bv, bfc (64 each), b1 (256), b2, ln1g, ln1b, ln2g, ln2b (64 each)
#define WS_ADVP(n, bit) (ws += ((S2S_ABL & (bit)) ? 0 : (n)))   // timing ablation: this phase's unit loads hit the same (L1-hot) lines
// Scheduling barriers pin the weight-unit / K,V-fragment loads in front of the MFMAs they are prefetched behind; the
// -DS2S_NO_SB_* builds measure what they are worth (the one in the attention pass: 15 % of the kernel).
#ifndef S2S_NO_SB_ATT
#define SB_ATT() __builtin_amdgcn_sched_barrier(0)
#else
#define SB_ATT()
#endif
#ifndef S2S_NO_SB_GEMM
#define SB_GEMM() __builtin_amdgcn_sched_barrier(0)
#else
#define SB_GEMM()
#endif
#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
#ifndef S2S_ATT32
#define S2S_ATT32 1
#endif
#ifndef S2S_ATT32_MSLOT
#define S2S_ATT32_MSLOT 1
#endif
#ifndef S2S_PRIO_MODE
#define S2S_PRIO_MODE 4      // 4: the two waves of a SIMD balance their progress through the attention loop (prio_balance); 0: off (A/B builds)
#endif
#ifndef S2S_ONLINE2
#define S2S_ONLINE2 1           // the exact instance runs softmax_pv32_online (0: the fast instance's out-of-line fallback, A/B)
#endif
#ifndef S2S_FAST_HI_MAX
#define S2S_FAST_HI_MAX 1       // the fast path's pass 0 takes its row maxima from the first score MFMA alone (0: from the full score; + 1.05 %, same MAE)
#endif
#ifndef S2S_ONLINE_HI_MAX
#define S2S_ONLINE_HI_MAX 1     // softmax_pv32_online takes a pass's row maxima from the FIRST score MFMA alone (0: from the full score, A/B)
#endif
#ifndef S2S_INT_SHIFT
#define S2S_INT_SHIFT 1         // softmax_pv32_online rounds its shift UP to an integer: one f16 half carries it -- no lo half, no re-encoding (round 6:
                                // 201.6 k -> 200.25 k cycles per chunk at the same clock, + 0.8 % chunks/s, MAE unchanged; 0: the two-half shift, A/B)
#endif
#ifndef S2S_MFMA_SPLIT
#define S2S_MFMA_SPLIT 0        // 1: the lo half of P comes from the matrix pipe (pv_split_mfma below) instead of v_fma_mix.  Round 6, same-box A/B
                                // (profiles/r06/ab_mfma_split.txt): 190.2 k -> 180.5 k cycles per chunk (exact: 201.6 k -> 191.2 k), MAE unchanged -- and the
                                // clock under the kernel falls from 2.29 to 2.18 GHz: the same chunks per second to 0.1 %.  The kernel is ENERGY bound.
#endif
#ifndef S2S_ONE_ZEROS_ROW
#define S2S_ONE_ZEROS_ROW 0     // 1: round 2's single zeros row (2-way LDS bank conflict on every V read; kept for the counter A/B)
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMAW(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct HL { h8 hi, lo; };

__device__ __forceinline__ h8 as_h8(const f32x4 v) { return __builtin_bit_cast(h8, v); }

// x = hi + lo with hi = f16(x) (round to nearest), lo = f16(x - hi): 22 significant bits, in three
// VALU instructions per pair of values: v_cvt_pk_f16_f32, then v_fma_mix{lo,hi}_f16 computes
// x * 1.0 - f32(hi) from the fp32 value and the f16 half with a single rounding straight to f16.
// Written in C so that every instruction stays visible to the compiler's hazard recogniser (MFMA <->
// VALU wait states, trans forwarding): `one` is an opaque 1.0f (otherwise fma(x, 1, c) folds to an add
// and the mix form is lost), the empty asm makes the packed hi opaque so its halves are reused through
// op_sel instead of being converted again, and the file is built with -fno-slp-vectorize (the SLP
// vectoriser would turn the two fmas into v_pk_fma_f32 plus separate conversions: five instructions).
// LO = false (the reduced-precision S2S_MODE_F16 decoder): only the f16 rounding of x is kept, lo = 0 -- one
// v_cvt_pk_f16_f32 per pair; every MFMA that would take a lo operand is skipped at compile time by its caller.
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
template <bool LO = true>
__device__ __forceinline__ void split2(const float a, const float b, const float one, unsigned& hi, unsigned& lo) {
    unsigned hb = __builtin_bit_cast(unsigned, (h2v{(_Float16)a, (_Float16)b}));
    if (!LO) { hi = hb; lo = 0u; return; }
    asm("" : "+v"(hb));
    const h2v h = __builtin_bit_cast(h2v, hb);
    const h2v l = {(_Float16)__builtin_fmaf(a, one, -(float)h[0]), (_Float16)__builtin_fmaf(b, one, -(float)h[1])};
    hi = hb;
    lo = __builtin_bit_cast(unsigned, l);
}
// p = exp2(s) for four scores, split.  p itself is not needed afterwards (the row sum comes out of
// the MFMA with a ones operand).
template <bool LO = true>
__device__ __forceinline__ void exp_split4(const f32x4 s, const float one, unsigned& h0, unsigned& h1, unsigned& l0,
                                           unsigned& l1) {
    const float e0 = (S2S_ABL & 1) ? s[0] : __builtin_amdgcn_exp2f(s[0]), e1 = (S2S_ABL & 1) ? s[1] : __builtin_amdgcn_exp2f(s[1]);
    const float e2 = (S2S_ABL & 1) ? s[2] : __builtin_amdgcn_exp2f(s[2]), e3 = (S2S_ABL & 1) ? s[3] : __builtin_amdgcn_exp2f(s[3]);
    if (S2S_ABL & 2) {      // timing only: no split arithmetic
        h0 = __float_as_uint(e0); l0 = __float_as_uint(e1); h1 = __float_as_uint(e2); l1 = __float_as_uint(e3);
        return;
    }
    split2<LO>(e0, e1, one, h0, l0);
    split2<LO>(e2, e3, one, h1, l1);
}
typedef unsigned uv4 __attribute__((ext_vector_type(4)));
typedef unsigned uv2 __attribute__((ext_vector_type(2)));
template <bool LO = true>
__device__ __forceinline__ HL split8(const f32x4 t0, const f32x4 t1, const float one) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2<LO>(t0[0], t0[1], one, h0, l0);
    split2<LO>(t0[2], t0[3], one, h1, l1);
    split2<LO>(t1[0], t1[1], one, h2, l2);
    split2<LO>(t1[2], t1[3], one, h3, l3);
    HL o;
    o.hi = __builtin_bit_cast(h8, (uv4{h0, h1, h2, h3}));
    o.lo = __builtin_bit_cast(h8, (uv4{l0, l1, l2, l3}));
    return o;
}
template <bool LO = true>
__device__ __forceinline__ void split4(const f32x4 t, const float one, h4& hi, h4& lo) {
    unsigned h0, h1, l0, l1;
    split2<LO>(t[0], t[1], one, h0, l0);
    split2<LO>(t[2], t[3], one, h1, l1);
    hi = __builtin_bit_cast(h4, (uv2{h0, h1}));
    lo = __builtin_bit_cast(h4, (uv2{l0, l1}));
}

// Key ORDER inside the decoder's K and V^T images (ATT32 layout): the blocks of four consecutive keys are dealt out over the
// four 64-key passes of the attention loop -- pass j holds the key blocks b = j (mod 4), i.e. keys 4j .. 4j+3, 16+4j .., 32+4j ..
// -- so that pass 0, whose row maxima are the fast path's softmax shift, is a sample of the WHOLE row instead of its first 64
// keys.  A softmax does not care about the order of its keys; the images are written in this order (an address, nothing
// else), the loop is unchanged.  What it buys (tools/redo_model.py, the share of heads the fast path has to redo): decoder
// w_qs / w_ks x 2: 55 % -> 7 %; attention on the positional encoding (w_qs = w_ks = 2 I): 9 % -> 0.2 %, (3 I): 29 % -> 9 %.
// Inside pass j the blocks are rotated by j & 1, so that the 16 lanes of a K store (keys 16 T .. 16 T + 15, four of them in each
// pass) still spread over eight 8-byte bank slots (2-way, as in the natural order) instead of four.
// position of key (T, c) = 16 T + c:  64 j + 4 ((T + (j & 1)) & 15) + (c & 3) with j = c >> 2;  key at position p:
// NATURAL = true (the EXACT kernel instance): position = key.  An online softmax has no use for a sample in pass 0, and in the
// dealt-out order the six phantom keys (250-255) sit in two passes (tiles 5 and 6) instead of one (tile 7): their masks cost the
// exact instance 5.3 k cycles per chunk and 7 spilled registers (round 4's review, item 3).
template <bool NATURAL = false>
__host__ __device__ constexpr int att32_key_at(const int p) {
    return NATURAL ? p : 16 * ((((p & 63) >> 2) - ((p >> 6) & 1)) & 15) + 4 * (p >> 6) + (p & 3);
}
// phantom keys (>= TV) -> -inf in the score tile `tile` (a compile-time index; h: the lane half)
template <int TV, bool NATURAL = false>
__device__ __forceinline__ void att32_mask_tile(f32x16& t, const int tile, const int h) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const bool p0 = att32_key_at<NATURAL>(32 * tile + row) >= TV, p1 = att32_key_at<NATURAL>(32 * tile + row + 4) >= TV;   // lane halves 0 / 1
        if (p0 && p1) t[r] = -__builtin_inff();
        else if (p1) t[r] = h ? -__builtin_inff() : t[r];
        else if (p0) t[r] = h ? t[r] : -__builtin_inff();
    }
}

// LDS of the f16 block: K [head][hi|lo][key][8 d] halves, V^T [head][16 rows: 0-7 hi d, 8-15 lo d][VS keys]
// halves, and a per-wave scratch for the Q^T operand re-layout.
template <int NQ, int WAVES, int NKT = 16> struct AttnLdsH {
    static constexpr int KEYS = 16 * NKT;
    static constexpr int K_BYTES = 8 * 2 * KEYS * 8 * 2;
    // halves per V^T row.  NKT even (the decoder): a row is stored as [32-key block][lane group g][8 halves] -- the 4 keys 4g..4g+3 of
    // the block's first tile, then those of its second -- so the P.V operand of a K = 32 block is ONE ds_read_b128 per lane;
    // 272 halves = 136 dwords == 8 (mod 64) makes the 16 lanes of every b128 lane group hit 16 distinct 4-dword bank slots.
    // (The natural key order needed two b64 reads, which hipcc fuses into ds_read2_b64: banked mod 32, 2-way conflicts on
    // every access -- the 11,264 SQ_LDS_BANK_CONFLICT cycles per chunk of profiles/r01.)  Odd NKT: natural order.
    static constexpr bool V128 = (NKT % 2) == 0;
    // ATT32 (the decoder with S2S_ATT32): the attention core runs on v_mfma_f32_32x32x16_f16 (softmax_pv32).  A V^T row is then
    // stored as [16-key step][lane half h][8 halves] -- keys 4h..4h+3 and 8+4h..8+4h+3 of the step: the 8 k-slots a lane half feeds
    // -- 264 halves = 132 dwords == 4 (mod 64): V^T row r lands in 16-byte bank slot r mod 16, so the 16 data rows of a head never collide
    // (the constant rows the other lanes read are placed per lane group: see vp in fft_block_h).  Constant rows follow the V region (all ones: the A-operand row that makes the MFMA add up P; all zeros: rows 17-31 of that
    // operand and the unused k-slots of the second Q operand; {1, 1, 0, 0, 0, 0, 0, 0} repeated: the k-slots of the second score
    // MFMA's A operand that take the softmax shift), written once per kernel (att32_consts).
    static constexpr bool ATT32 = S2S_ATT32 && NQ == 2 && NKT == 16;
    static constexpr int VS = ATT32 ? KEYS + 8 : V128 ? KEYS + 16 : KEYS + 8;
    static constexpr int V_BYTES = 8 * 16 * VS * 2;
    static constexpr int C_ROWS = 3;                               // rows: ones | zeros | {1, 1, 0, 0, 0, 0, 0, 0} repeated (a second zeros row: S2S_Z2_*)
    static constexpr int C_BYTES = ATT32 ? C_ROWS * VS * 2 : 0;
    static constexpr int Q_WAVE_BYTES = NQ * 2 * 2 * 16 * 8 * 2;
    static constexpr int BYTES = K_BYTES + V_BYTES + C_BYTES + WAVES * Q_WAVE_BYTES;
};

// acc[q] += W_unit * x[q]: one 16-row m-tile, K = 64 as two k-blocks, three products each.
// Unit layout (4 KiB): [kb0 hi][kb0 lo][kb1 hi][kb1 lo], 16 B per lane each.
template <int NQ, bool LO = true>
__device__ __forceinline__ void mm_unit_h(f32x4 (&acc)[NQ], const f32x4 (&f)[4], const HL (&x)[NQ][2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const h8 wh = as_h8(f[2 * kb]), wl = as_h8(f[2 * kb + 1]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = MFMAH(wh, x[q][kb].hi, acc[q]);
        if (LO) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = MFMAH(wh, x[q][kb].lo, acc[q]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = MFMAH(wl, x[q][kb].hi, acc[q]);
        }
    }
}

// acc[q] += (W_unit * x[q])^T: the A and B fragments of the K = 32 MFMA have the same lane layout, so swapping the operands
// transposes the product for free -- rows of the accumulator tile are then 4 consecutive TIME columns, its column one
// output feature (used for V^T, whose LDS rows run along the keys).
template <int NQ, bool LO = true>
__device__ __forceinline__ void mm_unit_h_t(f32x4 (&acc)[NQ], const f32x4 (&f)[4], const HL (&x)[NQ][2]) {
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const h8 wh = as_h8(f[2 * kb]), wl = as_h8(f[2 * kb + 1]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q] = MFMAH(x[q][kb].hi, wh, acc[q]);
        if (LO) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = MFMAH(x[q][kb].lo, wh, acc[q]);
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = MFMAH(x[q][kb].hi, wl, acc[q]);
        }
    }
}

// y = W x + b for a 64x64 Linear stored as 4 f16 units (one per 16-row m-tile), NQ time tiles.
template <int NQ>
__device__ __forceinline__ void linear64_h(const float* __restrict__ wu, const float* __restrict__ bias, int lane,
                                           const HL (&xb)[NQ][2], f32x4 (&y)[NQ][4]) {
    const int g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f32x4 f[4], t[NQ];
        load_unit(f, wu + mt * 1024 + lane * 4);
        const f32x4 b = ldg4(bias + 16 * mt + 4 * g);           // the bias: C operand of the tile's first MFMA
#pragma unroll
        for (int q = 0; q < NQ; ++q) t[q] = b;
        mm_unit_h<NQ>(t, f, xb);
#pragma unroll
        for (int q = 0; q < NQ; ++q) y[q][mt] = t[q];
    }
}

// Softmax(Q K^T) V of one head for this wave's NQ query tiles: sums of V.P (oH: P_hi, oL: P_lo; rows 0-7 against V_hi,
// rows 8-15 against V_lo) and of P (lH, lL), over NH passes of HK key tiles.  Pass 0 subtracts its own column max; later
// passes get "score - m" straight out of the MFMA (the accumulator starts at -m).
//   SAFE = false (fast): m stays the pass-0 max (+ S2S_SHIFT_BIAS) and later passes compute no max at all.  Softmax is
//     shift-invariant, so this is exact as long as no later score beats m by the f16 range of P_hi; if one does, the
//     row sum turns inf/NaN, which the caller checks once per head, and then runs
//   SAFE = true (rare): a textbook online softmax, the running max raised and the sums rescaled in every pass.
template <int NQ, int NKT, int TV, bool SAFE, bool LO = true>
__device__ __forceinline__ void softmax_pv(const _Float16* __restrict__ kp, const _Float16* __restrict__ vp, const h8 (&qb)[NQ],
                                           const h8 ones, const float one, const int g, f32x4 (&oH)[NQ], f32x4 (&oL)[NQ],
                                           f32x4 (&lH)[NQ], f32x4 (&lL)[NQ]) {
    constexpr int NH = (NKT >= 16) ? 4 : 1, HK = NKT / NH, HB = (HK + 1) / 2;
    f32x4 negm[NQ];
    float m[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        oH[q] = f32x4{0, 0, 0, 0}; oL[q] = f32x4{0, 0, 0, 0};
        lH[q] = f32x4{0, 0, 0, 0}; lL[q] = f32x4{0, 0, 0, 0};
        negm[q] = f32x4{0, 0, 0, 0};
        m[q] = 0.0f;
    }
#pragma unroll
    for (int h2 = 0; h2 < ((S2S_ABL & 256) ? 1 : NH); ++h2) {
        h8 ka[HK], va[HB];
#pragma unroll
        for (int kt = 0; kt < HK; ++kt) ka[kt] = *reinterpret_cast<const h8*>(kp + 16 * (h2 * HK + kt) * 8);
#pragma unroll
        for (int kb = 0; kb < HB; ++kb) {
            if (AttnLdsH<NQ, 1, NKT>::V128) {              // (vp already points at this lane group's 8 halves of block 0)
                va[kb] = *reinterpret_cast<const h8*>(vp + 32 * (h2 * HB + kb));
            } else {
                const h4 v0 = *reinterpret_cast<const h4*>(vp + 16 * (h2 * HK + 2 * kb));
                h4 v1 = h4{0, 0, 0, 0};                    // a K = 32 block past the last key tile: zero keys
                if (2 * kb + 1 < HK) v1 = *reinterpret_cast<const h4*>(vp + 16 * (h2 * HK + 2 * kb + 1));
                va[kb] = h8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            }
        }
        SB_ATT();
        // all NQ time tiles go through a pass together, so that one tile's MFMAs can run
        // beside the other's exponentials inside the same wave
        f32x4 s[NQ][HK];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int kt = 0; kt < HK; ++kt)
                s[q][kt] = (S2S_ABL & 1024) ? (negm[q] + __builtin_bit_cast(f32x4, ka[kt])) : (h2 == 0) ? MFMAH(ka[kt], qb[q], (f32x4{0, 0, 0, 0})) : MFMAH(ka[kt], qb[q], negm[q]);
        if (TV < 16 * NKT && h2 == NH - 1) {       // phantom keys -> -inf (only the last key tile has any)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * (NKT - 1) + 4 * g + r >= TV) s[q][HK - 1][r] = -__builtin_inff();
        }
        if (h2 == 0 || SAFE) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                float mh = s[q][0][0];
                if (!(S2S_ABL & 32)) {
#pragma unroll
                    for (int kt = 0; kt < HK; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) mh = fmaxf(mh, s[q][kt][r]);
                    mh = max_g(mh);
                }
                if (h2 == 0) {
                    if (!SAFE) mh += S2S_SHIFT_BIAS;
                    m[q] = mh;
                    negm[q] = f32x4{-mh, -mh, -mh, -mh};
                    // "score - m" comes from the matrix cores again: 8 issue cycles per tile instead of four subtractions (16)
                    if (!SAFE && !(TV < 16 * NKT && NH == 1)) {
#pragma unroll
                        for (int kt = 0; kt < HK; ++kt) s[q][kt] = MFMAH(ka[kt], qb[q], negm[q]);
                    } else {                                      // (a single-pass block has already masked its phantom keys in s)
#pragma unroll
                        for (int kt = 0; kt < HK; ++kt) s[q][kt] -= mh;
                    }
                } else {                              // safe attempt: raise the running max, rescale the sums
                    const float delta = fmaxf(mh, 0.0f);
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    oH[q] *= alpha; lH[q] *= alpha;
                    if (LO) { oL[q] *= alpha; lL[q] *= alpha; }
                    m[q] += delta;
                    negm[q] = f32x4{-m[q], -m[q], -m[q], -m[q]};
#pragma unroll
                    for (int kt = 0; kt < HK; ++kt) s[q][kt] -= delta;
                }
            }
        }
        HL P[NQ][HB];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int kb = 0; kb < HB; ++kb) {
                unsigned h0, h1, h2_ = 0, h3 = 0, l0, l1, l2 = 0, l3 = 0;
                exp_split4<LO>(s[q][2 * kb], one, h0, h1, l0, l1);
                if (2 * kb + 1 < HK) exp_split4<LO>(s[q][2 * kb + 1], one, h2_, h3, l2, l3);
                P[q][kb].hi = __builtin_bit_cast(h8, (uv4{h0, h1, h2_, h3}));
                P[q][kb].lo = __builtin_bit_cast(h8, (uv4{l0, l1, l2, l3}));
            }
#pragma unroll
            for (int kb = 0; kb < HB; ++kb) {
                if (S2S_ABL & 512) { asm volatile("" ::"v"(P[q][kb].hi), "v"(P[q][kb].lo)); continue; }
                oH[q] = MFMAH(va[kb], P[q][kb].hi, oH[q]);  // rows 0-7: V_hi.P_hi, rows 8-15: V_lo.P_hi
                if (LO) oL[q] = MFMAH(va[kb], P[q][kb].lo, oL[q]);  // rows 0-7: V_hi.P_lo, rows 8-15: V_lo.P_lo
                if (!(S2S_ABL & 64)) {
                lH[q] = MFMAH(ones, P[q][kb].hi, lH[q]);    // every row: sum of the P actually used
                if (LO) lL[q] = MFMAH(ones, P[q][kb].lo, lL[q]);
                }
            }
        }
    }
}

// ---- The decoder's attention core on the 32x32x16 MFMA (S2S_ATT32): Softmax(Q K^T) V of one head for the wave's 32 queries.
// A SIMD issues matrix and vector instructions through one port, a well-spaced MFMA holds it for ~8 cycles whatever its shape
// (tools/probes/issue_probe.hip), and the attention loop is bound by exactly that port: the 32x32x16 shape does the same flops
// in half the instructions.  Lane (h = lane >> 5, n = lane & 31); an accumulator tile (f32x16) holds rows (r & 3) + 8 (r >> 2) + 4h
// of column n in register r, so registers 8s .. 8s+7 of a score tile ARE the lane's 8 k-slots of 16-key step s of the P.V MFMA:
//   scores (32 keys x 32 queries): A = K row of key 32t + n': [K_hi | K_lo] over the lane halves, B1 = [Q_hi | Q_hi]; then
//     A2 = [K_hi | 1, 1, 0..] (kp2: a constant LDS row in the upper half), B2 = [Q_lo | -m_hi, -m_lo, 0..]
//     -> K_hi.Q_hi + K_lo.Q_hi + K_hi.Q_lo - m in two MFMAs (the SAFE path passes -m as the C operand instead);
//   P.V (16 keys per MFMA): A rows 0-7 V_hi d, 8-15 V_lo d, 16 ones, 17-31 zeros (constant LDS rows); B = P_hi, then P_lo, into ONE
//     accumulator: O[d] = row d + row 8+d is an in-lane add and row 16 is the softmax row sum (V_lo.P_lo rides along: 2^-22).
// 52 MFMAs per head instead of 104, no cross-lane traffic between the score tile and the P.V operand.  SAFE as in softmax_pv.
__device__ __forceinline__ float max_h(float v) {              // over the two lane halves
    const unsigned u = __float_as_uint(v);
    auto t = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(t[0]), __uint_as_float(t[1]));
}
__device__ __forceinline__ float sum_h(float v) {
    const unsigned u = __float_as_uint(v);
    auto t = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(t[0]) + __uint_as_float(t[1]);
}
// P = hi + lo WITHOUT the vector ALU's v_fma_mix (8 issue cycles per score, the largest single item of the attention loop beside the
// exponential itself): hi = f16(P) is one v_cvt_pk per two scores; lo = P - hi is computed by the MATRIX pipe as
//     D = C + A . B   with C = the f32 tile of P (the exponentials, in place), B = the hi operand the P.V product takes anyway,
//                     A = a constant 32 x 16 selection matrix holding one -1 per row,
// once per 16-key step (two MFMAs per 32-key tile): every element of D receives exactly one non-zero product, -hi of its own
// position, so D = P - hi to the accumulator's precision (hi is P rounded to 11 bits: the difference is exact in fp32), and a second
// v_cvt_pk per two scores turns it into the lo operand.  Vector issue per score: 7.9 (v_exp_f32) + 2 x 2.06 (v_cvt_pk / 2) = 12.0
// cycles against 18.0 (profiles/r02/valu_cost_probe.txt); the matrix pipe takes 2 more 32x32x16 products per tile (6 -> 8).
// Which element of A: the B operand of step st carries, in k-slot 8 h' + j of column n, the register 8 st + j of lane (n, h'),
// i.e. tile row (j & 3) + 8 (j >> 2) + 4 h' + 16 st; lane (row, h') of the A operand holds k-slots 8 h' .. 8 h' + 7 of that row: the
// -1 sits in the lane whose half h' equals bit 2 of the row, at j = (row & 3) + 4 ((row >> 3) & 1), in the step st = row >> 4.
struct SelA { h8 a[2]; };
__device__ __forceinline__ SelA split_selectors(const int h) {
    const int row = threadIdx.x & 31, rho = row & 15;
    const int j = (rho & 3) + 4 * (rho >> 3);
    const unsigned w = (((rho >> 2) & 1) == h) ? ((j & 1) ? 0xBC000000u : 0x0000BC00u) : 0u;      // f16 -1.0 in the low / high half
    uv4 sel;
#pragma unroll
    for (int d = 0; d < 4; ++d) sel[d] = (j >> 1) == d ? w : 0u;
    const uv4 z = {0u, 0u, 0u, 0u};
    SelA o;
    o.a[0] = __builtin_bit_cast(h8, row < 16 ? sel : z);
    o.a[1] = __builtin_bit_cast(h8, row < 16 ? z : sel);
    return o;
}
// one 32-key tile: sc = the shifted scores (f32, C layout) -> O += [V_hi; V_lo; 1] . exp2(sc), hi and lo halves
template <bool LO>
__device__ __forceinline__ void pv_split_mfma(f32x16 sc, const h8 (&va)[2], const SelA& sel, f32x16& O) {
    unsigned hw[8];
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = __builtin_amdgcn_exp2f(sc[r]);
#pragma unroll
    for (int p = 0; p < 8; ++p) hw[p] = __builtin_bit_cast(unsigned, (h2v{(_Float16)sc[2 * p], (_Float16)sc[2 * p + 1]}));
    const h8 hb0 = __builtin_bit_cast(h8, (uv4{hw[0], hw[1], hw[2], hw[3]})), hb1 = __builtin_bit_cast(h8, (uv4{hw[4], hw[5], hw[6], hw[7]}));
    if (LO) {
        sc = MFMAW(sel.a[1], hb1, MFMAW(sel.a[0], hb0, sc));          // P - hi, from the matrix pipe
        unsigned lw[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) lw[p] = __builtin_bit_cast(unsigned, (h2v{(_Float16)sc[2 * p], (_Float16)sc[2 * p + 1]}));
        O = MFMAW(va[0], hb0, O);
        O = MFMAW(va[0], __builtin_bit_cast(h8, (uv4{lw[0], lw[1], lw[2], lw[3]})), O);
        O = MFMAW(va[1], hb1, O);
        O = MFMAW(va[1], __builtin_bit_cast(h8, (uv4{lw[4], lw[5], lw[6], lw[7]})), O);
    } else {
        O = MFMAW(va[0], hb0, O);
        O = MFMAW(va[1], hb1, O);
    }
}
template <int TV, bool SAFE, bool LO, bool NATURAL = false>
__device__ __forceinline__ void softmax_pv32(const _Float16* __restrict__ kp, const _Float16* __restrict__ kp2, const _Float16* __restrict__ vp,
                                             const h8 qb1, const h8 qb2, const float one, const int h, f32x16& O,
                                             [[maybe_unused]] const int layer = 0) {
    constexpr int NT = 8;                                       // key tiles of 32
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto k_of = [&](const int t) { return *reinterpret_cast<const h8*>(kp + 32 * t * 8); };
    auto scores = [&](const h8 ka, const f32x16 c0) { return MFMAW(ka, qb2, MFMAW(ka, qb1, c0)); };
    O = zero16;
    auto v_of = [&](const int t, const int st) { return *reinterpret_cast<const h8*>(vp + 16 * (2 * t + st)); };
    auto mask_pass = [&](f32x16 (&t)[2], const int h2) {                  // phantom keys -> -inf (att32_key_at: they sit in tiles 5 and 6)
        att32_mask_tile<TV, NATURAL>(t[0], 2 * h2, h);
        att32_mask_tile<TV, NATURAL>(t[1], 2 * h2 + 1, h);
    };
    auto pv = [&](const f32x16& t, const h8 (&va)[2]) {                   // O += [V_hi; V_lo; 1] . exp2(t), 16 keys per MFMA
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            unsigned h0, h1, h2_, h3, l0, l1, l2, l3;
            exp_split4<LO>(f32x4{t[8 * st], t[8 * st + 1], t[8 * st + 2], t[8 * st + 3]}, one, h0, h1, l0, l1);
            exp_split4<LO>(f32x4{t[8 * st + 4], t[8 * st + 5], t[8 * st + 6], t[8 * st + 7]}, one, h2_, h3, l2, l3);
            O = MFMAW(va[st], __builtin_bit_cast(h8, (uv4{h0, h1, h2_, h3})), O);
            if (LO) O = MFMAW(va[st], __builtin_bit_cast(h8, (uv4{l0, l1, l2, l3})), O);
        }
    };
    f32x16 negm = zero16;
    h8 qb2m = qb2;
#if S2S_MFMA_SPLIT
    [[maybe_unused]] const SelA sel = split_selectors(h);
#endif
    if constexpr (!SAFE) {
        // fast path: the shift is pass 0's column max (+ head-room), later passes compute no max at all.  Two tiles (64 keys) per
        // pass, as straight-line code.  (Measured: issuing tile t+1's score MFMAs ahead of tile t's exponentials by hand -- no
        // s_nop in front of the first exponential any more -- is 1.3 % SLOWER than leaving the pass to the compiler.)
#pragma unroll
        for (int h2 = 0; h2 < NT / 2; ++h2) {
            h8 ka[2], va[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ka[i] = k_of(2 * h2 + i);
                va[i][0] = v_of(2 * h2 + i, 0); va[i][1] = v_of(2 * h2 + i, 1);
            }
            SB_ATT();
#if S2S_ATT32_MSLOT
            // the shift rides in the second score MFMA: its A operand carries {1, 1, 0..} instead of K_lo in the upper lane half's
            // k-slots (kp2: a constant LDS row there, the K_hi row below), its B operand {-m_hi, -m_lo, 0..} where Q_lo has none
            h8 kb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) kb[i] = *reinterpret_cast<const h8*>(kp2 + (2 * h2 + i) * (h ? 0 : 32 * 8));
            f32x16 sc[2];
#if S2S_FAST_HI_MAX
            // (pass 0's row maxima need only the first score MFMA, as in softmax_pv32_online: the shift has 2 log2 units of head-room
            //  and the missing K_hi . Q_lo term is 2^-11 of the score)
#pragma unroll
            for (int i = 0; i < 2; ++i) sc[i] = h2 == 0 ? MFMAW(ka[i], qb1, zero16) : MFMAW(kb[i], qb2m, MFMAW(ka[i], qb1, zero16));
#else
#pragma unroll
            for (int i = 0; i < 2; ++i) sc[i] = MFMAW(kb[i], qb2m, MFMAW(ka[i], qb1, zero16));
#endif
            mask_pass(sc, h2);
            if (h2 == 0) {
                float mh = sc[0][0];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mh = fmaxf(mh, sc[i][r]);
                const float nm = -(max_h(mh) + S2S_SHIFT_BIAS);
                const _Float16 nh = (_Float16)nm;
                const unsigned pk = __builtin_bit_cast(unsigned, (h2v{nh, (_Float16)(nm - (float)nh)}));
                uv4 q2 = __builtin_bit_cast(uv4, qb2);
                q2[0] = h ? pk : q2[0];
                qb2m = __builtin_bit_cast(h8, q2);
#if S2S_FAST_HI_MAX
                // (what the maxima were taken from IS the first product of the score: the second MFMA accumulates onto it)
#pragma unroll
                for (int i = 0; i < 2; ++i) sc[i] = MFMAW(kb[i], qb2m, sc[i]);
#else
#pragma unroll
                for (int i = 0; i < 2; ++i) sc[i] = MFMAW(kb[i], qb2m, MFMAW(ka[i], qb1, zero16));
#endif
            }
#else
            f32x16 sc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) sc[i] = scores(ka[i], negm);     // (pass 0: negm = 0)
            mask_pass(sc, h2);
            if (h2 == 0) {
                float mh = sc[0][0];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mh = fmaxf(mh, sc[i][r]);
                mh = max_h(mh) + S2S_SHIFT_BIAS;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -mh;
#pragma unroll
                for (int i = 0; i < 2; ++i) sc[i] = scores(ka[i], negm); // "score - m" from the matrix cores again
            }
#endif
#ifdef S2S_TILEHIST
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float m16[2];
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    float m = sc[i][8 * st];
#pragma unroll
                    for (int r = 1; r < 8; ++r) m = fmaxf(m, sc[i][8 * st + r]);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                    m16[st] = m + S2S_SHIFT_BIAS;                  // relative to the row's pass-0 maximum
                }
                auto bin = [](const float t) { return t > 0.0f ? 63 : (-t >= 62.0f ? 62 : (int)(-t)); };
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&s2s_hist_lds[(layer * 2 + 0) * 64 + bin(fmaxf(m16[0], m16[1]))], 1u);
                    atomicAdd(&s2s_hist_lds[(layer * 2 + 1) * 64 + bin(m16[0])], 1u);
                    atomicAdd(&s2s_hist_lds[(layer * 2 + 1) * 64 + bin(m16[1])], 1u);
                }
            }
#endif
#if S2S_MFMA_SPLIT
#pragma unroll
            for (int i = 0; i < 2; ++i) pv_split_mfma<LO>(sc[i], va[i], sel, O);
#else
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    unsigned h0, h1, h2_, h3, l0, l1, l2, l3;
                    exp_split4<LO>(f32x4{sc[i][8 * st], sc[i][8 * st + 1], sc[i][8 * st + 2], sc[i][8 * st + 3]}, one, h0, h1, l0, l1);
                    exp_split4<LO>(f32x4{sc[i][8 * st + 4], sc[i][8 * st + 5], sc[i][8 * st + 6], sc[i][8 * st + 7]}, one, h2_, h3, l2, l3);
                    O = MFMAW(va[i][st], __builtin_bit_cast(h8, (uv4{h0, h1, h2_, h3})), O);
                    if (LO) O = MFMAW(va[i][st], __builtin_bit_cast(h8, (uv4{l0, l1, l2, l3})), O);
                }
#endif
        }
    } else {
        float m = 0.0f;
#pragma unroll
        for (int h2 = 0; h2 < NT / 2; ++h2) {                            // a textbook online softmax, 64 keys per pass
            h8 ka[2], va[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ka[i] = k_of(2 * h2 + i);
                va[i][0] = v_of(2 * h2 + i, 0); va[i][1] = v_of(2 * h2 + i, 1);
            }
            f32x16 sc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) sc[i] = scores(ka[i], negm);     // (pass 0: negm = 0)
            mask_pass(sc, h2);
            float mh = sc[0][0];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mh = fmaxf(mh, sc[i][r]);
            mh = max_h(mh);
            const float delta = (h2 == 0) ? mh : fmaxf(mh, 0.0f);         // raise the running max, rescale the sums
            if (h2 > 0) O *= __builtin_amdgcn_exp2f(-delta);
            m += delta;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] = -m;
#pragma unroll
            for (int i = 0; i < 2; ++i) { sc[i] -= delta; pv(sc[i], va[i]); }
        }
    }
}
// The EXACT attention path (s2s_fused_kernel<.., EXACT = true>, s2s_set_attention_path; s2s_create picks it for weights whose
// calibration launch redoes more than S2S_ATTENTION_REDO_THRESHOLD of its heads) is the online softmax -- running maximum raised
// and sums rescaled in every 64-key pass, branch-free -- as the ONLY path of its kernel instance: the same shader cycles per chunk
// on every checkpoint, whatever the weights (201.6 k, profiles/r05/attention_paths.txt), where "fast path, then redo" costs 190.2 k on
// diffuse attention and 310-328 k once most
// heads overflow.  As softmax_pv32<TV, SAFE = true> (the fast instance's out-of-line fallback) it measured 221.7 k; as
// softmax_pv32_online below 209.0 k with every pass's maxima taken from the full score (206.5 k on round 4's device, 211.8 k while
// the exact instance still shared the fast one's dealt-out key order: the phantom-key masks sat in two passes and the kernel
// spilled 7 registers), 203.8 k with the maxima from the first score MFMA alone (S2S_ONLINE_HI_MAX, round 5), 201.6 k with the score's second MFMA
// accumulating onto that product instead of both being issued again (bit-identical; the masks applied once).
// (Round 5 also tried it in two phases -- the row maxima of all eight tiles from that first MFMA, then the fast path's body with the
// exact shift, no update or rescale in the loop, commit 7131d25: unrolled, hipcc hoists phase 1 to the front and spills 52 registers; as
// a real loop it runs 218.8 k: the separate phase exposes the K reads and the MFMA chain that the online loop hides.)
// Three cleverer exact paths were built in round 4 and lost to it (LABNOTES.md,
// profiles/r04/attention_paths_*.txt; their code is in commits 3a2cca9 and the two after b739589):
//  * exact running maximum with lazy re-centring and every 16-key step CLASSIFIED by its largest shifted score (skipped below
//    -32, P_lo dropped below -16): 229-251 k, 224 k with 51 % of the steps skipped.  The verdict has to be wave-uniform, i.e. a
//    branch per pass, and a branch ends the basic block hipcc schedules across -- each pass then exposes its LDS reads and its
//    score-MFMA chain -- which costs more (27 k) than the skipped steps save;
//  * four other schedules of that path (per-step branches 246 k; a tile pipeline 283 k with 105 spilled registers; K operands a
//    pass ahead 245 k; next pass's scores at the end of each block: 121 spills);
//  * two passes over the keys, branch-free (all row maxima first, then the fast path's body with the exact shift): 280 k, 93 spills.
// softmax_pv32_online: that online softmax with the fast path's operand tricks -- the shift rides in the k-slots of the second
// score MFMA (no 16-register C operand), a pass's scores are ISSUED AGAIN with the raised shift instead of being lowered by 32
// subtractions (the matrix pipe has the room), and only accumulator registers 0-8 are rescaled (rows 17-31 of O are zeros).
template <int TV, bool LO, bool NATURAL>
__device__ __forceinline__ void softmax_pv32_online(const _Float16* __restrict__ kp, const _Float16* __restrict__ kp2, const _Float16* __restrict__ vp,
                                                    const h8 qb1, const h8 qb2, const float one, const int h, f32x16& O) {
    constexpr int NT = 8;
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto k_of = [&](const int t) { return *reinterpret_cast<const h8*>(kp + 32 * t * 8); };
    auto v_of = [&](const int t, const int st) { return *reinterpret_cast<const h8*>(vp + 16 * (2 * t + st)); };
    O = zero16;
    h8 qb2m = qb2;
#if S2S_MFMA_SPLIT
    const SelA sel = split_selectors(h);
#endif
    float m = 0.0f;                                                       // the shift qb2m carries: exactly -(hi + lo)
#pragma unroll
    for (int h2 = 0; h2 < NT / 2; ++h2) {
        h8 ka[2], kb[2], va[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ka[i] = k_of(2 * h2 + i);
            kb[i] = *reinterpret_cast<const h8*>(kp2 + (2 * h2 + i) * (h ? 0 : 32 * 8));
            va[i][0] = v_of(2 * h2 + i, 0); va[i][1] = v_of(2 * h2 + i, 1);
        }
        SB_ATT();
        f32x16 sc[2];
        auto score_pass = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) sc[i] = MFMAW(kb[i], qb2m, MFMAW(ka[i], qb1, zero16));
            att32_mask_tile<TV, NATURAL>(sc[0], 2 * h2, h);               // phantom keys -> -inf (natural key order: tile 7 only)
            att32_mask_tile<TV, NATURAL>(sc[1], 2 * h2 + 1, h);
        };
#if S2S_ONLINE_HI_MAX
        // The shift only has to keep P inside the f16 range and be the SAME in the numerator and the row sum -- it need not be the
        // exact maximum.  The first score MFMA alone ([K_hi | K_lo] . [Q_hi | Q_hi]: everything but K_hi . Q_lo, i.e. the score to
        // 2^-11 of its size, a fraction of a log2 unit) gives the pass's row maxima for half the MFMAs of a full score pass, and
        // unshifted: the new shift is max(old shift, this pass's maximum).
#pragma unroll
        for (int i = 0; i < 2; ++i) sc[i] = MFMAW(ka[i], qb1, zero16);
        att32_mask_tile<TV, NATURAL>(sc[0], 2 * h2, h);
        att32_mask_tile<TV, NATURAL>(sc[1], 2 * h2 + 1, h);
#else
        score_pass();
#endif
        float mh = sc[0][0];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) mh = fmaxf(mh, sc[i][r]);
        mh = max_h(mh);                                                   // the row's maximum over this pass (HI_MAX: absolute; else relative to the shift)
#if S2S_ONLINE_HI_MAX
        const float nm = -(h2 == 0 ? mh : fmaxf(m, mh));
#else
        const float nm = -(h2 == 0 ? mh : m + fmaxf(mh, 0.0f));
#endif
#if S2S_INT_SHIFT
        // (the shift need not be the maximum, only no smaller: rounded up to an integer it fits ONE f16 half -- exactly below 2048,
        //  and what the half encodes is what is used otherwise -- so the lo half, its subtraction and two conversions go)
        const _Float16 nh = (_Float16)(-__builtin_ceilf(-nm));
        const unsigned pk = __builtin_bit_cast(unsigned, (h2v{nh, (_Float16)0.0f}));
        uv4 q2 = __builtin_bit_cast(uv4, qb2);
        q2[0] = h ? pk : q2[0];
        qb2m = __builtin_bit_cast(h8, q2);
        const float m_enc = -(float)nh;
#else
        const _Float16 nh = (_Float16)nm;
        const _Float16 nl = (_Float16)(nm - (float)nh);
        const unsigned pk = __builtin_bit_cast(unsigned, (h2v{nh, nl}));
        uv4 q2 = __builtin_bit_cast(uv4, qb2);
        q2[0] = h ? pk : q2[0];
        qb2m = __builtin_bit_cast(h8, q2);
        const float m_enc = -((float)nh + (float)nl);                     // what the two halves encode: exact in fp32
#endif
        if (h2 > 0) {
            const float f = __builtin_amdgcn_exp2f(m - m_enc);            // (1 for a row whose maximum did not rise)
#pragma unroll
            for (int r = 0; r < 9; ++r) O[r] *= f;
        }
        m = m_enc;
#if S2S_ONLINE_HI_MAX
        // the full score = what the maxima were taken from + the second MFMA (K_hi . Q_lo and the shift); the phantom keys stay -inf
#pragma unroll
        for (int i = 0; i < 2; ++i) sc[i] = MFMAW(kb[i], qb2m, sc[i]);
#else
        score_pass();
#endif
#if S2S_MFMA_SPLIT
#pragma unroll
        for (int i = 0; i < 2; ++i) pv_split_mfma<LO>(sc[i], va[i], sel, O);
#else
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                unsigned h0, h1, h2_, h3, l0, l1, l2, l3;
                exp_split4<LO>(f32x4{sc[i][8 * st], sc[i][8 * st + 1], sc[i][8 * st + 2], sc[i][8 * st + 3]}, one, h0, h1, l0, l1);
                exp_split4<LO>(f32x4{sc[i][8 * st + 4], sc[i][8 * st + 5], sc[i][8 * st + 6], sc[i][8 * st + 7]}, one, h2_, h3, l2, l3);
                O = MFMAW(va[i][st], __builtin_bit_cast(h8, (uv4{h0, h1, h2_, h3})), O);
                if (LO) O = MFMAW(va[i][st], __builtin_bit_cast(h8, (uv4{l0, l1, l2, l3})), O);
            }
#endif
    }
}
// the constant rows behind the V region (see AttnLdsH): called once per kernel, before the first barrier
template <class G> __device__ __forceinline__ void att32_consts(char* __restrict__ lds, const int tid, const int nthreads) {
    if constexpr (G::ATT32) {
        _Float16* cr = reinterpret_cast<_Float16*>(lds + G::K_BYTES + G::V_BYTES);
        for (int i = tid; i < G::C_ROWS * G::VS; i += nthreads)
            cr[i] = (i < G::VS || (i >= 2 * G::VS && i < 3 * G::VS && ((i - 2 * G::VS) & 7) < 2)) ? (_Float16)1.0f : (_Float16)0.0f;
    }
}

// the second zeros row (see vp in fft_block_h): behind the small vectors and the progress counters, first byte on bank slot 4
__device__ __forceinline__ const _Float16* zeros_row2(const float* sv_lds) {
    const unsigned a = (unsigned)(size_t)(sv_lds + S2S_SV_FLOATS + S2S_PROG_INTS);          // (LDS addresses are 32-bit offsets)
    return reinterpret_cast<const _Float16*>(sv_lds + S2S_SV_FLOATS + S2S_PROG_INTS) + (((4u * 16u - (a & 255u)) & 255u) >> 1);
}

// ---- wave balancing inside a SIMD (S2S_PRIO_MODE >= 4).  The two waves of a SIMD run the same attention loop, and the SIMD's
// arbiter always prefers the OLDER wave (waves 0-3 of the workgroup): it runs at the speed of a lone wave, the younger one only
// fills its stalls, and once the older wave has reached the barrier behind the loop the younger one runs alone with all of its
// own stalls exposed (measured with the per-wave phase stamps: 79 k against 126 k cycles per chunk in the loop, 48 k of barrier
// wait for the older wave).  Here every wave counts the heads it has finished in an LDS word and raises its priority whenever
// it is not ahead of its partner (wave ^ 4), so that the two alternate as the favoured wave and reach the barrier together.
__device__ __forceinline__ void prio_balance(int* __restrict__ prog, const int wave) {
    const int lane = threadIdx.x & 63;
    int mine = prog[wave] + 1;
    if (lane == 0) prog[wave] = mine;
    const int theirs = __builtin_amdgcn_readfirstlane(prog[wave ^ 4]);
    mine = __builtin_amdgcn_readfirstlane(mine);
    if (mine <= theirs) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
}

// A weight unit of the f16 streams: [kb0 hi][kb0 lo][kb1 hi][kb1 lo] (4 KiB), or, for the single-product mode, the hi-only
// stream [kb0 hi][kb1 hi] (2 KiB) loaded into the same even slots.
template <bool LO>
__device__ __forceinline__ void load_unit_h(f32x4 (&f)[4], const float* __restrict__ ws) {
    if (LO) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = ldg4(ws + i * 256);
    } else {
        f[0] = ldg4(ws); f[2] = ldg4(ws + 256);
        f[1] = f32x4{0, 0, 0, 0}; f[3] = f32x4{0, 0, 0, 0};
    }
}

// One DECODER FFTBlock (layers.py:116-142), same contract as fft_block in s2s_device.h (the f16x3 encoder has its own,
// register-resident form at the end of this file).
template <int NQ, int WAVES, int NKT, int TV, bool LO = true, bool EXACT = false>
__device__ __forceinline__ void fft_block_h(const float* __restrict__ W, const LayerOff L, f32x4 (&X)[NQ][4],
                                            char* __restrict__ lds, int qt0, int wave, int lane, const float one,
                                            [[maybe_unused]] unsigned long long* diag_buf = nullptr, const float* __restrict__ pf_src = nullptr,
                                            float* __restrict__ pf_dst = nullptr, float* __restrict__ sv_lds = nullptr) {
    using G = AttnLdsH<NQ, WAVES, NKT>;
    // SV: the layer's small vectors (biases, LayerNorm weights: S2S_SV_FLOATS contiguous floats from L.bq_nat, pack_layer) are
    // copied into LDS at the block's entry -- one f32x4 per thread, loaded before the entry barrier, stored after it, visible
    // from the K/V barrier on -- so that every later use is a ds_read instead of an L2 round trip in front of the MFMAs or
    // the LayerNorm that needs it.  (bk / bv are needed before that barrier and stay global, one pair ahead.)
    constexpr bool SV = (WAVES == 8);                 // (the decoder; a single-wave instantiation reads them from L2)
    auto svp = [&](const int off) -> const float* {
        if constexpr (SV) return sv_lds + (off - L.bq_nat); else return W + off;
    };
    const int sv_i = wave * 64 + lane;
    f32x4 svv = f32x4{0, 0, 0, 0};
    if (SV && sv_i < S2S_SV_FLOATS / 4) svv = ldg4(W + L.bq_nat + 4 * sv_i);
    constexpr int NH = (NKT >= 16) ? 4 : 1, HK = NKT / NH;                       // 256 keys: 4 passes of 64
    [[maybe_unused]] constexpr int HB = (HK + 1) / 2;
    const int g = lane >> 4, c = lane & 15;
    DIAG_DECL;
    _Float16* __restrict__ Kl = reinterpret_cast<_Float16*>(lds);
    _Float16* __restrict__ Vl = reinterpret_cast<_Float16*>(lds + G::K_BYTES);
    _Float16* __restrict__ Ql = reinterpret_cast<_Float16*>(lds + G::K_BYTES + G::V_BYTES + G::C_BYTES + wave * G::Q_WAVE_BYTES);
    constexpr int UF = LO ? 1024 : 512;             // floats per weight unit: hi+lo fragments, or the hi-only stream
    const float* ws = W + (LO ? L.stream_h : L.stream_f) + lane * 4;
    f32x4 fa[4], fb[4];
    load_unit_h<LO>(fa, ws); WS_ADVP(UF, 2048);                    // Wk, pair 0

    HL xb[NQ][2];                                     // block input as B operands
#pragma unroll
    for (int q = 0; q < NQ; ++q) { xb[q][0] = split8<LO>(X[q][0], X[q][1], one); xb[q][1] = split8<LO>(X[q][2], X[q][3], one); }

    if (!(S2S_ABL & 4)) block_sync<WAVES == 1>();     // every wave is done reading the previous block's K/V
    DIAG_STAMP(0);
    if (SV && sv_i < S2S_SV_FLOATS / 4) *reinterpret_cast<f32x4*>(sv_lds + 4 * sv_i) = svv;
    // ---- K^T and V^T of this wave's time tiles, all heads -> LDS as hi/lo halves (layers.py:74-78)
    f32x4 bk_n = ldg4(W + L.bk_nat + 4 * g);
    float bv_n = W[L.bv + c];                                  // V comes out transposed: this lane's column is one feature
#pragma unroll 1
    for (int p = 0; p < 4; ++p) {
        if (S2S_ABL & 4096) { for (int i = 0; i < 4; ++i) fb[i] = fa[i]; } else load_unit_h<LO>(fb, ws);   // (4096: timing without this phase's loads)
        WS_ADVP(UF, 2048);                                   // Wv, pair p
        const f32x4 bk = bk_n;
        const float bv = bv_n;
        bk_n = ldg4(W + L.bk_nat + 16 * (p < 3 ? p + 1 : 3) + 4 * g);    // the next pair's, behind this pair's MFMAs
        bv_n = W[L.bv + 16 * (p < 3 ? p + 1 : 3) + c];
        SB_GEMM();
        // the biases are the accumulators' initial values (the C operand of each tile's first MFMA): no add afterwards
        f32x4 ak[NQ], av[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { ak[q] = bk; av[q] = f32x4{bv, bv, bv, bv}; }
        if (!((S2S_ABL & (1 << 22)) && pf_src == nullptr)) mm_unit_h<NQ, LO>(ak, fa, xb);      // (1 << 22: timing without layer 0's Q/K/V MFMAs)
        SB_GEMM();
        if (!(S2S_ABL & 4096) || p == 3) load_unit_h<LO>(fa, ws);
        WS_ADVP(UF, 2048);                                   // Wk, pair p+1 (after the last pair: Wq, pair 0)
        SB_GEMM();
        if (!((S2S_ABL & (1 << 22)) && pf_src == nullptr)) mm_unit_h_t<NQ, LO>(av, fb, xb);                           // av[q]: rows = times 4g..4g+3 of the tile, column c = feature 16p + c
        const int head = 2 * p + (g >> 1), d0 = 4 * (g & 1);   // K accumulator rows 4g..4g+3 = head, d0..d0+3
        const int vrow = (2 * p + (c >> 3)) * 16 + (c & 7);    // V^T row of this lane's feature (hi; lo is 8 rows below)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int T = qt0 + q;
            // (ATT32, fast instance: the position of att32_key_at; the EXACT instance keeps the natural order)
            const int key = G::ATT32 && !EXACT ? 64 * (c >> 2) + 4 * ((T + ((c >> 2) & 1)) & 15) + (c & 3) : 16 * T + c;
            h4 hi, lo;
            split4<LO>(ak[q], one, hi, lo);
            *reinterpret_cast<h4*>(Kl + ((head * 2 + 0) * G::KEYS + key) * 8 + d0) = hi;
            *reinterpret_cast<h4*>(Kl + ((head * 2 + 1) * G::KEYS + key) * 8 + d0) = lo;
            split4<LO>(av[q], one, hi, lo);                           // 4 consecutive keys of one V^T row: one b64 store each
            // (ATT32: this lane's four keys are key block 4 T + g, at block position 16 g + Tr, Tr = (T + (g & 1)) & 15: 16-key step
            //  4 g + (Tr >> 2), block Tr & 3 of it)
            const int Tr = (T + (g & 1)) & 15;
            //  natural order (EXACT): key block 4 T + g = 16-key step T, block g of it)
            const int vcol = G::ATT32 ? (EXACT ? 16 * T + 8 * (g & 1) + 4 * ((g >> 1) & 1)
                                               : 16 * (4 * g + (Tr >> 2)) + 8 * (Tr & 1) + 4 * ((Tr >> 1) & 1))
                           : G::V128 ? 32 * (T >> 1) + 8 * g + 4 * (T & 1) : 16 * T + 4 * g;   // (position inside the row: see AttnLdsH::VS)
            *reinterpret_cast<h4*>(Vl + vrow * G::VS + vcol) = hi;
            *reinterpret_cast<h4*>(Vl + (vrow + 8) * G::VS + vcol) = lo;
        }
    }
    DIAG_STAMP(1);
    if (!(S2S_ABL & 4)) block_sync<WAVES == 1>();     // K/V of every wave visible (and the small vectors: svp() from here on)
    DIAG_STAMP(2);
    // ---- fc accumulator starts as bias + residual (layers.py:85-86)
    f32x4 acc[NQ][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b = ldg4(svp(L.bfc) + 16 * mt + 4 * g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q][mt] = X[q][mt] + b;
    }

    [[maybe_unused]] const float c1 = 1.4426950408889634f * 0.35355339059327373f;     // log2(e) / sqrt(d_k = 8)
    h8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;
#pragma unroll 1
    for (int u = 0; u < 2; ++u) {                     // two head pairs per iteration = one K = 32 block of fc
        f32x4 opair[2][NQ];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int p = 2 * u + pp;
            // weight units of this iteration: Wq(2u) [fa], Wq(2u+1) [fb], Wfc(u) m-tiles 0-1 [fa], 2-3 [fb]
            if (pp == 0) { load_unit_h<LO>(fb, ws); WS_ADVP(UF, 8192); } else { load_unit_h<LO>(fa, ws); WS_ADVP(UF, 8192); }
            const f32x4 bq = ldg4(svp(L.bq_nat) + 16 * p + 4 * g);   // (pack_layer: Wq and bq of the f16 streams carry the log2(e)/sqrt(d_k) factor)
            SB_GEMM();
            f32x4 qa[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) qa[q] = bq;
            if (!((S2S_ABL & (1 << 22)) && pf_src == nullptr)) { if (pp == 0) mm_unit_h<NQ, LO>(qa, fa, xb); else mm_unit_h<NQ, LO>(qa, fb, xb); }
            SB_GEMM();
            // Q^T rows live 4 per lane group; the S MFMA wants all 8 d of a head in every lane
            // ([Q_hi | Q_lo | Q_hi | Q_lo] over the lane groups): re-layout through the wave's scratch
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                h4 hi, lo;
                split4<LO>(qa[q], one, hi, lo);                           // scores come out in log2 units
                *reinterpret_cast<h4*>(Ql + ((((g >> 1) * NQ + q) * 2 + 0) * 16 + c) * 8 + 4 * (g & 1)) = hi;   // [head of the pair][q][hi|lo][c][8 d]
                *reinterpret_cast<h4*>(Ql + ((((g >> 1) * NQ + q) * 2 + 1) * 16 + c) * 8 + 4 * (g & 1)) = lo;
            }
            if constexpr (G::ATT32) {
                // both heads of the pair on the 32x32x16 core; a head's output (4 d per lane) goes through its own, by then dead,
                // half of the wave's Q scratch into the pair-tile layout the fc operand wants (row 4g+r: head g >> 1, d = 4 (g & 1) + r)
                const int hl = lane >> 5, n = lane & 31;
                const _Float16* const crow = Vl + 8 * 16 * G::VS;         // [ones row][zeros row][{1, 1, 0..} row]
                const _Float16* const zrow2 = zeros_row2(sv_lds);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int head = 2 * p + hh;
                    if constexpr (S2S_PRIO_MODE >= 4 && WAVES == 8) prio_balance(reinterpret_cast<int*>(sv_lds + S2S_SV_FLOATS), wave);
                    const _Float16* qrow = Ql + (((hh * NQ + (n >> 4)) * 2 + 0) * 16 + (n & 15)) * 8;      // [head of the pair][q][hi|lo][c][8 d]
                    const h8 qb1 = *reinterpret_cast<const h8*>(qrow);
                    const h8 qb2 = *reinterpret_cast<const h8*>(hl ? crow + G::VS : qrow + 16 * 8);
                    const _Float16* kp = Kl + ((head * 2 + hl) * G::KEYS + n) * 8;                         // K_hi rows for h = 0, K_lo for h = 1
                    const _Float16* kp2 = hl ? crow + 2 * G::VS : kp;                                       // (second score MFMA: K_hi | {1, 1, 0..})
                    // (rows 17-31 of the P.V operand are zeros.  A ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27} and
                    // {4-11, 16-19, 28-31} of each lane half, and a V^T row r sits in 16-byte bank slot r mod 16: the zeros row the lanes of
                    // the FIRST group read must not share a slot with rows 0-3 / 12-15, that of the second none with rows 4-11 or the
                    // ones row -- two copies: one behind the small vectors, started on slot 4, and const row 1 (slot 1).  With a single zeros row in slot 1 lane 1's row
                    // and the zeros row collided in every such read: 2 extra LDS cycles per read, about half of round 2's
                    // SQ_LDS_BANK_CONFLICT count.)
                    const _Float16* vp = (n < 16 ? Vl + (head * 16 + n) * G::VS : n == 16 ? crow : (n >= 20 && n < 28 && !S2S_ONE_ZEROS_ROW) ? zrow2 : crow + G::VS) + 8 * hl;
                    f32x16 O;
                    float lsum;
                    if constexpr (EXACT) {                                 // the handle's attention path is "exact" (its own kernel instance)
#if S2S_ONLINE2 && S2S_ATT32_MSLOT
                        softmax_pv32_online<TV, LO, true>(kp, kp2, vp, qb1, qb2, one, hl, O);
#else
                        softmax_pv32<TV, true, LO, true>(kp, kp2, vp, qb1, qb2, one, hl, O);
#endif
                        lsum = sum_h(O[8]);                                // row 16 lives in the lower lane half
                    } else {
                    softmax_pv32<TV, S2S_ALWAYS_RESCALE != 0, LO>(kp, kp2, vp, qb1, qb2, one, hl, O, pf_src ? 1 : 0);
                    lsum = sum_h(O[8]);
#if !S2S_ALWAYS_RESCALE && !defined(S2S_NO_FALLBACK)
                    {
                        const bool redo = __any(!(lsum <= 3.0e38f));       // inf or NaN row sum: some P_hi left the f16 range
#ifdef S2S_DIAG
                        DIAG_COUNT(11, 1ull); if (redo) DIAG_COUNT(10, 1ull);
#endif
                        if (__builtin_expect(redo, 0)) {
                            if (lane == 0) atomicAdd(&s2s_stats_lds[S2S_STAT_REDO], 1u);     // production counter (s2s_stats_read)
                            // (the textbook fallback, not softmax_pv32_online: as the out-of-line branch of THIS kernel the faster function
                            //  costs the fast path 2.2 k cycles of registers and schedule -- commit 066dc5f -- and the weights that redo much run
                            //  the exact instance anyway)
                            softmax_pv32<TV, true, LO>(kp, kp2, vp, qb1, qb2, one, hl, O);
                            lsum = sum_h(O[8]);
                        }
                    }
#endif
                    }
                    const float inv = __builtin_amdgcn_rcpf(lsum);
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (O[r] + O[4 + r]) * inv;          // V_hi row d + V_lo row d, d = 4h + r
                    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(Ql + hh * (NQ * 2 * 16 * 8)) + n * 8 + 4 * hl) = o;
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    opair[pp][q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(Ql + (g >> 1) * (NQ * 2 * 16 * 8)) +
                                                                   (q * 16 + c) * 8 + 4 * (g & 1));
            } else {
            // The pair's fc operand takes head 2p from lanes g < 2 and head 2p+1 from lanes g >= 2.  Head 2p's output is
            // parked, for the lanes that will use it, in the first half of the wave's Q scratch (head 2p's Q^T, dead once
            // qb has been read) instead of eight more live registers through the second head's softmax.
            f32x4 ohead1[NQ];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int head = 2 * p + hh;
                h8 qb[NQ];
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    qb[q] = *reinterpret_cast<const h8*>(Ql + (((hh * NQ + q) * 2 + (g & 1)) * 16 + c) * 8);
                const _Float16* kp = Kl + ((head * 2 + (g >> 1)) * G::KEYS + c) * 8;   // [K_hi | K_hi | K_lo | K_lo]
                const _Float16* vp = Vl + (head * 16 + c) * G::VS + (G::V128 ? 8 : 4) * g;   // row c: 0-7 V_hi d, 8-15 V_lo d
                f32x4 oH[NQ], oL[NQ], lH[NQ], lL[NQ];
                softmax_pv<NQ, NKT, TV, S2S_ALWAYS_RESCALE != 0, LO>(kp, vp, qb, ones, one, g, oH, oL, lH, lL);
#if !S2S_ALWAYS_RESCALE && !defined(S2S_NO_FALLBACK)   // (NO_FALLBACK: test-only build, proves test_peaked_attention... needs the redo)
                {
                    bool bad = false;                          // inf or NaN row sum: some P_hi left the f16 range
#pragma unroll
                    for (int q = 0; q < NQ; ++q) bad = bad || !(lH[q][0] + (LO ? lL[q][0] : 0.0f) <= 3.0e38f);
                    const bool redo = __any(bad);
#ifdef S2S_DIAG
                    DIAG_COUNT(11, 1ull); if (redo) DIAG_COUNT(10, 1ull);
#endif
                    if (__builtin_expect(redo, 0)) {
                        if (lane == 0) atomicAdd(&s2s_stats_lds[S2S_STAT_REDO], 1u);
                        softmax_pv<NQ, NKT, TV, true, LO>(kp, vp, qb, ones, one, g, oH, oL, lH, lL);
                    }
                }
#endif
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const f32x4 t = LO ? oH[q] + oL[q] : oH[q];
                    const float inv = __builtin_amdgcn_rcpf(LO ? lH[q][0] + lL[q][0] : lH[q][0]);   // v_rcp_f32 (1 ulp), not the 10-instruction division
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {            // O[d] = row d + row 8+d: the other half-wave's value
                        const unsigned uu = __float_as_uint(t[r]);
                        auto sw = __builtin_amdgcn_permlane32_swap(uu, uu, false, false);
                        o[r] = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) * inv;
                    }
                    // lanes g and g^2 both hold d = 4(g&1) + r
                    if (hh == 0) { if (g < 2) *reinterpret_cast<f32x4*>(Ql + (q * 32 + lane) * 8) = o; }
                    else ohead1[q] = o;
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {                 // pair tile: row 4g+r
                const f32x4 o0 = *reinterpret_cast<const f32x4*>(Ql + (q * 32 + (lane & 31)) * 8);
                opair[pp][q] = (g < 2) ? o0 : ohead1[q];
            }
            }
        }
        // ---- fc, k-block u (the 4 heads just finished): acc += Wfc[:, 32u : 32u+32] * O^T
        HL ob[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) ob[q] = split8<LO>(opair[0][q], opair[1][q], one);
            load_unit_h<LO>(fb, ws); WS_ADVP(UF, 8192);                // Wfc(u), m-tiles 2-3
        SB_GEMM();
#pragma unroll
        for (int half = 0; half < 2; ++half) {        // unit = [mt a hi][mt a lo][mt b hi][mt b lo]
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const h8 wh = as_h8(half == 0 ? fa[2 * mm] : fb[2 * mm]);
                const h8 wl = as_h8(half == 0 ? fa[2 * mm + 1] : fb[2 * mm + 1]);
                const int mt = 2 * half + mm;
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q][mt] = MFMAH(wh, ob[q].hi, acc[q][mt]);
                if (LO) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q][mt] = MFMAH(wh, ob[q].lo, acc[q][mt]);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q][mt] = MFMAH(wl, ob[q].hi, acc[q][mt]);
                }
            }
            if (half == 0) {
                SB_GEMM();
                load_unit_h<LO>(fa, ws); WS_ADVP(UF, 8192);        // next iteration's Wq (after the last: W1 unit 0)
                SB_GEMM();
            }
        }
    }
    if (S2S_PRIO_MODE != 0 && WAVES == 8) __builtin_amdgcn_s_setprio(0);
    if constexpr (!(S2S_FFN_LDS && WAVES == 8 && LO)) DIAG_STAMP(3);

    // ---- FFN 64 -> 256 -> 64 in four 64-wide slices of the hidden layer (layers.py:108-113)
    // A weight unit feeds only 6*NQ MFMAs (~200 cycles) here, less than an L2 round trip, so the stream
    // runs three units ahead through a ring of four unit buffers (8 units per slice: slots repeat).
    // (a single-tile instantiation gets a two-deep ring: 32 registers less)
    constexpr int RD = (NQ >= 2) ? 4 : 2, RM = RD - 1;               // ring depth; units in flight = RD - 1
    f32x4 ring[RD][4];
    // FFN_LDS (the 8-wave decoder): every wave needs every FFN weight unit, and eight copies of the 128 KiB through the CU's
    // 64 B/clk vector-L1 path cost more than the FFN's MFMAs (timing without these loads: +7.7 %).  The K/V region is dead
    // once every wave has left the attention loop, so the workgroup copies the FFN stream into it ONCE -- wave w: unit
    // 8hc + w of every slice hc, 16 KiB -- and the FFN reads its A fragments with ds_read_b128 (256 B/clk).  The copy goes
    // through registers (global_load_dwordx4 issued before the barrier, ds_write_b128 after it): LDS-DMA lands at only
    // ~12 B/clk per CU and cost 11 k cycles per layer when tried.  The next block's entry barrier protects the region again.
    constexpr bool FFN_LDS = S2S_FFN_LDS && WAVES == 8 && LO && (G::K_BYTES + G::V_BYTES >= 32 * 4096);
    const float* wl = reinterpret_cast<const float*>(lds) + lane * 4;    // this lane's 16 B of every staged 1-KiB fragment
    if constexpr (FFN_LDS) {
        f32x4 gm[4], bt[4], b2v[4], stage[16];
#pragma unroll
        for (int hc = 0; hc < 4; ++hc)
#pragma unroll
            for (int i = 0; i < 4; ++i) stage[4 * hc + i] = ldg4(ws + (8 * hc + wave - 1) * 1024 + i * 256);   // (ws is already one unit into the FFN)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            gm[ft] = ldg4(svp(L.ln1g) + 16 * ft + 4 * g); bt[ft] = ldg4(svp(L.ln1b) + 16 * ft + 4 * g);
            b2v[ft] = ldg4(svp(L.b2) + 16 * ft + 4 * g);
        }
        // piggy-back (the decoder's last layer): PF_FLOATS floats from pf_src -> LDS at pf_dst, visible to every wave after
        // the second barrier below like the weights themselves
        constexpr int PF_VEC = S2S_PF_FLOATS / 4;
        const int pf_i = wave * 64 + lane;
        f32x4 pfv = f32x4{0, 0, 0, 0};
        if (pf_src && pf_i < PF_VEC) pfv = ldg4(pf_src + 4 * pf_i);
        DIAG_STAMP(3);
        __syncthreads();                                             // every wave is done reading K/V
        DIAG_STAMP(12);
        if (pf_src && pf_i < PF_VEC) *reinterpret_cast<f32x4*>(pf_dst + 4 * pf_i) = pfv;
        float* wst = reinterpret_cast<float*>(lds) + lane * 4;
#pragma unroll
        for (int hc = 0; hc < 4; ++hc)
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(wst + (8 * hc + wave) * 1024 + i * 256) = stage[4 * hc + i];
        layer_norm64_r<NQ, true>(acc, gm, bt);                             // acc = x1
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int q = 0; q < NQ; ++q) X[q][mt] = acc[q][mt] + b2v[mt];    // X = bias + residual accumulator
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[0][i] = fa[i];              // W1 unit 0 (requested during the last P.V)
        if constexpr (RD == 4) { load_unit_h<LO>(ring[1], ws); load_unit_h<LO>(ring[2], ws + UF); WS_ADVP(2 * UF, 16384); }
        SB_GEMM();
        layer_norm64<NQ, true>(acc, svp(L.ln1g), svp(L.ln1b), g);          // acc = x1
    }
    if constexpr (!FFN_LDS) DIAG_STAMP(4);
    HL x1b[NQ][2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { x1b[q][0] = split8<LO>(acc[q][0], acc[q][1], one); x1b[q][1] = split8<LO>(acc[q][2], acc[q][3], one); }
    if constexpr (FFN_LDS) {
        DIAG_STAMP(4);
        __syncthreads();                                             // every wave's fragments are in LDS
        DIAG_STAMP(14);
#pragma unroll
        for (int u = 0; u < RM; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[u][i] = ldg4(wl + u * 1024 + i * 256);
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const f32x4 b = ldg4(svp(L.b2) + 16 * mt + 4 * g);
#pragma unroll
            for (int q = 0; q < NQ; ++q) X[q][mt] = acc[q][mt] + b;  // X = bias + residual accumulator
        }
    }
#pragma unroll 1
    for (int hc = 0; hc < ((S2S_ABL & 128) ? 0 : 4); ++hc) {
        f32x4 hid[NQ][4];
        f32x4 b1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) b1[mt] = ldg4(svp(L.b1) + 64 * hc + 16 * mt + 4 * g);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {              // W1 units: rows 64hc + 16mt ..
            f32x4 t[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[q] = b1[mt];   // (the bias: C operand of the tile's first MFMA)
            if constexpr (FFN_LDS) {                  // unit 8hc + mt + RM, three ahead of its use (wraps past the end: harmless)
#pragma unroll
                for (int i = 0; i < 4; ++i) ring[(mt + RM) & RM][i] = ldg4(wl + ((8 * hc + mt + RM) & 31) * 1024 + i * 256);
            } else {
                if (!(S2S_ABL & 32768)) load_unit_h<LO>(ring[(mt + RM) & RM], ws);     // (32768: timing without the FFN's weight loads)
                WS_ADVP(UF, 16384);
            }
            SB_GEMM();
            mm_unit_h<NQ, LO>(t, ring[mt & RM], x1b);
            SB_GEMM();
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) hid[q][mt][r] = (S2S_ABL & (1 << 23)) ? t[q][r] : relu1(t[q][r]);      // (1 << 23: timing without the FFN's relu / split VALU work)
        }
        HL hb[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (S2S_ABL & (1 << 23)) {
                hb[q][0].hi = as_h8(hid[q][0]); hb[q][0].lo = as_h8(hid[q][1]); hb[q][1].hi = as_h8(hid[q][2]); hb[q][1].lo = as_h8(hid[q][3]);
            } else { hb[q][0] = split8<LO>(hid[q][0], hid[q][1], one); hb[q][1] = split8<LO>(hid[q][2], hid[q][3], one); }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {              // W2 units: rows 16mt .., columns 64hc ..
            f32x4 t[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[q] = X[q][mt];
            if constexpr (FFN_LDS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) ring[(mt + RM) & RM][i] = ldg4(wl + ((8 * hc + 4 + mt + RM) & 31) * 1024 + i * 256);
            } else {
                if (!(S2S_ABL & 32768)) load_unit_h<LO>(ring[(mt + RM) & RM], ws);
                WS_ADVP(UF, 16384);
            }
            SB_GEMM();
            mm_unit_h<NQ, LO>(t, ring[mt & RM], hb);
            SB_GEMM();
#pragma unroll
            for (int q = 0; q < NQ; ++q) X[q][mt] = t[q];
        }
    }
    DIAG_STAMP(5);
    layer_norm64<NQ, true>(X, svp(L.ln2g), svp(L.ln2b), g);
    DIAG_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The ENCODER FFTBlock (layers.py:116-142) for NQ independent 16-position sequences (one chunk each) owned by ONE wave, with the
// attention entirely in registers: with T = 16 a head pair's K^T, V^T and Q^T are single accumulator tiles, and
//   * a K^T tile (lane (g, c): features 4g..4g+3 of key c) IS the A operand of the score MFMA -- its eight k-slots take the
//     four features' hi halves and their lo halves -- against B = [Q_hi | Q_hi] and [Q_lo | 0] of the same lanes (A and B
//     fragments of the K = 32 MFMA share one lane layout); the contraction then runs over the pair's 16 features, so the
//     other head's lane groups are zeroed in B (head 2p: groups 0-1, head 2p+1: groups 2-3);
//   * the operand-swapped V GEMM leaves V^T (lane (g, c): keys 4g..4g+3 of feature c), the A operand of the P.V MFMA,
//     against B = [P_hi | P_hi] and [P_lo | 0] straight from the score tile's layout (rows = keys, column = query); the
//     product's rows are the pair's 16 features, of which head 2p owns 0-7 (lane groups 0-1) and head 2p+1 the rest.
// No LDS image of K/V, no barrier, no re-layout; the eight softmaxes of a sequence are independent 16x16 tiles (the decoder's
// version of this code streams 256 keys through LDS instead).  Products are the same three f16 terms as everywhere else.
//
// `wl`: the layer's f16 weight units (1024 floats each; this lane's 16 bytes of every fragment at + 4 * lane) in the order of
// pack_layer's stream, read straight from L2.

// attention half: X (block input) -> acc = fc(attention) + bias + residual (layers.py:74-86).  wl: the layer's units 0-15.
template <int NQ>
__device__ __forceinline__ void enc_attention_h(const float* __restrict__ W, const LayerOff L, const float* __restrict__ wl,
                                                const f32x4 (&X)[NQ][4], f32x4 (&acc)[NQ][4], const int lane, const float one) {
    const int g = lane >> 4, c = lane & 15;
    constexpr int UF = 1024;                                          // floats per unit
    wl += lane * 4;
    f32x4 fk[4], fv[4], fq[4], fc0[4], fc1[4];
    load_unit_h<true>(fk, wl);                                        // Wk, Wv of pair 0, Wq(0)
    load_unit_h<true>(fv, wl + UF);
    load_unit_h<true>(fq, wl + 8 * UF);
    HL xb[NQ][2];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { xb[q][0] = split8(X[q][0], X[q][1], one); xb[q][1] = split8(X[q][2], X[q][3], one); }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {                                  // fc accumulator: bias + residual
        const f32x4 b = ldg4(W + L.bfc + 16 * mt + 4 * g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[q][mt] = X[q][mt] + b;
    }
    [[maybe_unused]] const float c1 = 1.4426950408889634f * 0.35355339059327373f;     // log2(e) / sqrt(d_k = 8): scores in log2 units
    const bool low = g < 2;                                           // lane groups of the pair's first head
#pragma unroll 1
    for (int u = 0; u < 2; ++u) {
        load_unit_h<true>(fc0, wl + (8 + 4 * u + 2) * UF);            // Wfc(u): m-tiles 0-1, 2-3
        load_unit_h<true>(fc1, wl + (8 + 4 * u + 3) * UF);
        f32x4 opair[2][NQ];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int p = 2 * u + pp;
            const f32x4 bk = ldg4(W + L.bk_nat + 16 * p + 4 * g), bq = ldg4(W + L.bq_nat + 16 * p + 4 * g);
            const float bv = W[L.bv + 16 * p + c];                    // V comes out transposed: this lane's column is one feature
            SB_GEMM();
            f32x4 ak[NQ], av[NQ], aq[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) { ak[q] = bk; av[q] = f32x4{bv, bv, bv, bv}; aq[q] = bq; }   // biases: the first MFMAs' C operands
            mm_unit_h<NQ>(ak, fk, xb);
            mm_unit_h_t<NQ>(av, fv, xb);
            mm_unit_h<NQ>(aq, fq, xb);
            SB_GEMM();
            {   // the next pair's units, requested behind this pair's softmaxes (after the last pair: harmless re-reads)
                const int pn = p < 3 ? p + 1 : 3;
                load_unit_h<true>(fk, wl + (2 * pn) * UF);
                load_unit_h<true>(fv, wl + (2 * pn + 1) * UF);
                load_unit_h<true>(fq, wl + (8 + 4 * (pn >> 1) + (pn & 1)) * UF);
            }
            SB_GEMM();
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                h4 khi, klo, vhi, vlo, qhi, qlo;
                split4(ak[q], one, khi, klo);
                split4(av[q], one, vhi, vlo);
                split4(aq[q], one, qhi, qlo);             // (Wq, bq carry the log2(e)/sqrt(d_k) factor: scores in log2 units)
                const uv2 kh = __builtin_bit_cast(uv2, khi), kl = __builtin_bit_cast(uv2, klo);
                const uv2 vh = __builtin_bit_cast(uv2, vhi), vl = __builtin_bit_cast(uv2, vlo);
                const uv2 qh = __builtin_bit_cast(uv2, qhi), ql = __builtin_bit_cast(uv2, qlo);
                const h8 Ka = __builtin_bit_cast(h8, (uv4{kh[0], kh[1], kl[0], kl[1]}));
                const h8 Va = __builtin_bit_cast(h8, (uv4{vh[0], vh[1], vl[0], vl[1]}));
                f32x4 O[2];
                float inv[2];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const bool mine = (hh == 0) == low;               // this lane group carries features of head 2p + hh
                    const h8 Q1 = __builtin_bit_cast(h8, (uv4{mine ? qh[0] : 0u, mine ? qh[1] : 0u, mine ? qh[0] : 0u, mine ? qh[1] : 0u}));
                    const h8 Q2 = __builtin_bit_cast(h8, (uv4{mine ? ql[0] : 0u, mine ? ql[1] : 0u, 0u, 0u}));
                    f32x4 s = MFMAH(Ka, Q1, (f32x4{0, 0, 0, 0}));     // rows: keys 4g..4g+3, column: query c
                    s = MFMAH(Ka, Q2, s);
                    const float m = max_g(fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3])));
                    const float e0 = __builtin_amdgcn_exp2f(s[0] - m), e1 = __builtin_amdgcn_exp2f(s[1] - m);
                    const float e2 = __builtin_amdgcn_exp2f(s[2] - m), e3 = __builtin_amdgcn_exp2f(s[3] - m);
                    unsigned h0, h1, l0, l1;
                    split2(e0, e1, one, h0, l0);
                    split2(e2, e3, one, h1, l1);
                    const h8 P1 = __builtin_bit_cast(h8, (uv4{h0, h1, h0, h1}));
                    const h8 P2 = __builtin_bit_cast(h8, (uv4{l0, l1, 0u, 0u}));
                    inv[hh] = __builtin_amdgcn_rcpf(sum_g((e0 + e1) + (e2 + e3)));   // (the halves above carry these to 22 bits)
                    O[hh] = MFMAH(Va, P1, (f32x4{0, 0, 0, 0}));       // rows: the pair's features 4g..4g+3, column: query c
                    O[hh] = MFMAH(Va, P2, O[hh]);
                }
                opair[pp][q] = low ? O[0] * inv[0] : O[1] * inv[1];
            }
        }
        // ---- fc, k-block u (the 4 heads just finished): acc += Wfc[:, 32u : 32u+32] * O^T
        HL ob[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) ob[q] = split8(opair[0][q], opair[1][q], one);
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const h8 wh = as_h8(half == 0 ? fc0[2 * mm] : fc1[2 * mm]);
                const h8 wlo = as_h8(half == 0 ? fc0[2 * mm + 1] : fc1[2 * mm + 1]);
                const int mt = 2 * half + mm;
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q][mt] = MFMAH(wh, ob[q].hi, acc[q][mt]);
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q][mt] = MFMAH(wh, ob[q].lo, acc[q][mt]);
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q][mt] = MFMAH(wlo, ob[q].hi, acc[q][mt]);
            }
    }
}

// FFN entry: acc = attention output -> x1 = LN1(acc), its B operands x1b, and X = x1 + b2 (the FFN's accumulator; layers.py:108-113)
template <int NQ>
__device__ __forceinline__ void enc_ffn_begin_h(const float* __restrict__ W, const LayerOff L, f32x4 (&acc)[NQ][4], f32x4 (&X)[NQ][4],
                                                HL (&x1b)[NQ][2], const int lane, const float one) {
    const int g = lane >> 4;
    layer_norm64<NQ, true>(acc, W + L.ln1g, W + L.ln1b, g);                // acc = x1
#pragma unroll
    for (int q = 0; q < NQ; ++q) { x1b[q][0] = split8(acc[q][0], acc[q][1], one); x1b[q][1] = split8(acc[q][2], acc[q][3], one); }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b = ldg4(W + L.b2 + 16 * mt + 4 * g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) X[q][mt] = acc[q][mt] + b;
    }
}

// two of the FFN's four 64-wide hidden slices: X += W2[:, slice] relu(W1[slice] x1 + b1[slice]) for hc = hc0, hc0 + 1.
// wl: the 16 staged units of these slices (per slice: 4 of W1, then 4 of W2), read one unit ahead of use.
template <int NQ>
__device__ __forceinline__ void enc_ffn_half_h(const float* __restrict__ W, const LayerOff L, const float* __restrict__ wl, const int hc0,
                                               const HL (&x1b)[NQ][2], f32x4 (&X)[NQ][4], const int lane, const float one) {
    const int g = lane >> 4;
    constexpr int UF = 1024;
    wl += lane * 4;
    f32x4 ring[2][4];
    load_unit_h<true>(ring[0], wl);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        f32x4 hid[NQ][4], b1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) b1[mt] = ldg4(W + L.b1 + 64 * (hc0 + h) + 16 * mt + 4 * g);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {              // W1 units: rows 64hc + 16mt ..
            const int un = 8 * h + mt;                // this unit's index inside wl; the next one is requested now
            f32x4 t[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[q] = b1[mt];
            load_unit_h<true>(ring[(un + 1) & 1], wl + (un + 1) * UF);
            SB_GEMM();
            mm_unit_h<NQ>(t, ring[un & 1], x1b);
            SB_GEMM();
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) hid[q][mt][r] = relu1(t[q][r]);
        }
        HL hb[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { hb[q][0] = split8(hid[q][0], hid[q][1], one); hb[q][1] = split8(hid[q][2], hid[q][3], one); }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {              // W2 units: rows 16mt .., columns 64hc ..
            const int un = 8 * h + 4 + mt;
            f32x4 t[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) t[q] = X[q][mt];
            if (un < 15) load_unit_h<true>(ring[(un + 1) & 1], wl + (un + 1) * UF);
            SB_GEMM();
            mm_unit_h<NQ>(t, ring[un & 1], hb);
            SB_GEMM();
#pragma unroll
            for (int q = 0; q < NQ; ++q) X[q][mt] = t[q];
        }
    }
}
