// s2s_hip.hip -- kernels and C ABI (include/s2s_hip.h) of the MI355X-native seq2squiggle
// predict path.  gfx950 only; no fallbacks.
//
// One launch per tile of chunks, s2s_fused_kernel: persistent 8-wave workgroups, each walking its share of the chunks in
// groups of 16:
//   frontend   wave w, two chunks at a time: k-mer embedding gather, pre-net, the noise / duration heads, Philox Gamma or
//              Normal dwell, encoder FFT blocks -> enc_out, sigma, dur in the workgroup's L2-resident hand-off slots
//   decoder    all 8 waves, one chunk after the other: length-regulator gather + positional add, decoder FFT blocks
//              (T = 250 padded to 256; 32 time columns per wave), output projection, x165, Philox noise, clamp -> signal[250]
// plus s2s_export_* for the per-read zero-strip / int16 conversion and s2s_svb_* for the containers' signal codecs.
#include "s2s_device.h"
#include "s2s_device_h.h"
#include "../../include/s2s_hip.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

// ================================================================================ frontend
// The frontend of a chunk is run by ONE wave (T = 16: every chunk is a single time tile and a sequence of its own): embedding,
// pre-net, the three heads and the dwell source, encoder blocks.  Two chunks per wave (f16x3) give the in-order wave two
// independent dependency chains to interleave; the eight waves of the workgroup work side by side on different chunks.
template <int NQ>
__device__ __forceinline__ void relu_tiles(f32x4 (&x)[NQ][4]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[q][mt][r] = relu1(x[q][mt][r]);
}

__device__ __forceinline__ int base_code(unsigned char ch) {       // utils.py:74 letter_to_int
    return ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : ch == 'T' ? 4 : ch == '_' ? 0 : -1;
}

// One chunk as the frontend sees it.  bp: its 16+k-1 bases; inj_g / inj_zdw: its 16 injected variates (or null); slot: where
// the decoder picks it up -- enc_out [16][64], sigma [16] at +1024, dur [16] (int32) at +1040; out_dur: the caller's [16] row;
// dbg_idx: its row in the debug arrays; live = false: a filler (an odd chunk count), computed and thrown away.
struct FrontChunk {
    const uint8_t* bp;
    int nv;
    unsigned long long chunk;
    const float* inj_g;
    const float* inj_zdw;
    float* slot;
    int* out_dur;
    long long dbg_idx;
    bool live;
};
#define S2S_SLOT_FLOATS S2S_PF_FLOATS

// ---- src_emb on the one-hot k-mer == bias + sum of k gathered columns of W_emb, then ReLU (modules.py:70-73)
template <int NQ>
__device__ __forceinline__ void front_embed(const ModelDev& M, const float* __restrict__ W, const FrontChunk (&io)[NQ], const int lane,
                                            f32x4 (&X)[NQ][4]) {
    const int g = lane >> 4, c = lane & 15;
    f32x4 eb[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) eb[ft] = ldg4(W + M.emb_b + 16 * ft + 4 * g);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) X[q][ft] = eb[ft];
    for (int j = 0; j < M.k; ++j) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int code = (c < io[q].nv) ? base_code(io[q].bp[c + j]) : 0;   // pad k-mer = "_" * k (utils.py:342-347)
            const float* row = W + M.emb_wt + (5 * j + (code < 0 ? 0 : code)) * 64 + 4 * g;
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) {
                const f32x4 w = ldg4(row + 16 * ft);
                if (code >= 0) X[q][ft] += w;                        // unknown letter: all-zero one-hot row (utils.py:86)
            }
        }
    }
    relu_tiles<NQ>(X);
}

// ---- second layer of a head: Linear(64,1) + Softplus on the ReLU'd hidden tile (modules.py:267-278, 182-195)
template <int NQ>
__device__ __forceinline__ void head_out(const float* __restrict__ W, const MlpOff m, f32x4 (&hid)[NQ][4], const int lane, float (&out)[NQ]) {
    const int g = lane >> 4;
    relu_tiles<NQ>(hid);
    f32x4 w[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) w[mt] = ldg4(W + m.w3 + 16 * mt + 4 * g);
    const float b3 = W[m.b3];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        float part = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part += hid[q][mt][r] * w[mt][r];
        out[q] = softplus_t(sum_g(part) + b3);
    }
}

// ---- the dwell source (modules.py:396-438) and the sigma / dur stores, given the heads' values.  Every lane group holds the
//      heads' values of all tiles (sum_g is an all-reduce), so lane group q finishes chunk q: ONE pass of the sampler per wave.
template <int NQ>
__device__ __forceinline__ void front_dwell(const FrontChunk (&io)[NQ], const ParamsDev& P, const float (&sig)[NQ], const float (&cq)[NQ],
                                            const float (&rq)[NQ], const DebugDev& dbg, const int lane) {
    const int g = lane >> 4, c = lane & 15;
    float sigma = sig[0], conc = cq[0], rate = rq[0];
    FrontChunk me = io[0];
#pragma unroll
    for (int q = 1; q < NQ; ++q)
        if (g == q) { sigma = sig[q]; conc = cq[q]; rate = rq[q]; me = io[q]; }
    const bool mine = g < NQ && me.live;
    if (mine) {
        me.slot[1024 + c] = sigma;
        if (dbg.sigma) dbg.sigma[me.dbg_idx * 16 + c] = sigma;
    }
    float gv;
    if (P.duration_sampling) {
        conc = fmaxf(conc, 1e-8f);                                   // modules.py:215-216
        rate = fmaxf(rate, 1e-8f);                                   // modules.py:217-218
        if (mine) {
            if (dbg.conc) dbg.conc[me.dbg_idx * 16 + c] = conc;
            if (dbg.rate) dbg.rate[me.dbg_idx * 16 + c] = rate;
        }
        if (me.inj_g) {
            gv = mine ? me.inj_g[c] : 1.0f;
        } else {
            float sg = 0.0f;
            if (mine) sg = standard_gamma(conc, (unsigned)me.chunk, (unsigned)(me.chunk >> 32), c, P.seed_lo, P.seed_hi);
            gv = fmaxf(sg / rate, 1.17549435e-38f);                 // Gamma.sample: /rate, clamp_(tiny)
        }
        gv = fmaxf(gv, 1.0f);                                       // modules.py:223
        gv = fmaxf(gv, P.min_duration);                             // modules.py:414-416
    } else if (P.dwell_std <= 0.0f) {
        gv = P.dwell_mean;                                          // modules.py:420-423
    } else {
        float z;
        if (me.inj_zdw) {
            z = mine ? me.inj_zdw[c] : 0.0f;
        } else {
            const u32x4 r = philox4x32_10((unsigned)me.chunk, (unsigned)(me.chunk >> 32), c | (S2S_KIND_DWELL << 16), 0,
                                          P.seed_lo, P.seed_hi);
            z = box_muller(r.x, r.y);
        }
        gv = fmaxf(mul_then_add(z, P.dwell_std, P.dwell_mean), P.min_duration);   // modules.py:425-432
    }
    if (mine) {
        const float rd = fminf(fmaxf(rintf(gv), -1.0e9f), 1.0e9f);  // torch.round: half-to-even (modules.py:437)
        reinterpret_cast<int*>(me.slot + 1040)[c] = (int)rd;
        store_stream(me.out_dur + c, (int)rd);
        if (dbg.g) dbg.g[me.dbg_idx * 16 + c] = gv;
    }
}

// stand-alone operators (TEST instance, dbg.emb_in): the pre-net output is replaced by rows the caller supplies
template <int NQ>
__device__ __forceinline__ void front_emb_in(const FrontChunk (&io)[NQ], const DebugDev& dbg, f32x4 (&X)[NQ][4], const int lane) {
    if (!dbg.emb_in) return;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) X[q][ft] = ldg4(dbg.emb_in + (io[q].dbg_idx * 16 + c) * 64 + 16 * ft + 4 * g);
}

// X -> the chunks' hand-off slots (enc = true: enc_out) and, for the tests, the debug array of that stage (emb_out / enc_out)
template <int NQ>
__device__ __forceinline__ void front_store(const FrontChunk (&io)[NQ], const DebugDev& dbg, const f32x4 (&X)[NQ][4], const int lane,
                                            const bool enc) {
    const int g = lane >> 4, c = lane & 15;
    float* const dbg_dst = enc ? dbg.enc_out : dbg.emb_out;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (!io[q].live) continue;
        if (enc) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft) *reinterpret_cast<f32x4*>(io[q].slot + c * 64 + 16 * ft + 4 * g) = X[q][ft];
        }
        if (dbg_dst) {
#pragma unroll
            for (int ft = 0; ft < 4; ++ft)
                *reinterpret_cast<f32x4*>(dbg_dst + (io[q].dbg_idx * 16 + c) * 64 + 16 * ft + 4 * g) = X[q][ft];
        }
    }
}

// ---- f32 mode: one chunk per wave, every product on the f32-input MFMA, weights as f32 fragments from L2, encoder K/V in the
//      wave's own slice of LDS (AttnLds<1>)
struct FrontLdsF32 { static constexpr int BYTES = AttnLds<1>::BYTES; };
__device__ __forceinline__ void frontend_f32(const ModelDev& M, const float* __restrict__ W, const FrontChunk (&io)[1],
                                             const ParamsDev& P, char* __restrict__ lds_raw, const DebugDev& dbg, const int lane) {
    const int g = lane >> 4, c = lane & 15;
    f32x4 X[1][4];
    front_embed<1>(M, W, io, lane, X);
#pragma unroll 1
    for (int i = 0; i < M.pre_layers; ++i) {                         // modules.py:74-77
        f32x4 Y[1][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) Y[0][mt] = ldg4(W + M.pre_b[i] + 16 * mt + 4 * g);
        gemm_acc<1, 4, 4>(W + M.pre_w[i], lane, Y, X);
        relu_tiles<1>(Y);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) X[0][mt] = Y[0][mt];
    }
    front_emb_in<1>(io, dbg, X, lane);
    front_store<1>(io, dbg, X, lane, false);                // emb_out (debug only)
    float sig[1], cq[1] = {1.0f}, rq[1] = {1.0f};
    auto head = [&](const MlpOff m, float (&out)[1]) {
        f32x4 hid[1][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) hid[0][mt] = ldg4(W + m.b0 + 16 * mt + 4 * g);
        gemm_acc<1, 4, 4>(W + m.w0, lane, hid, X);
        head_out<1>(W, m, hid, lane, out);
    };
    head(M.noise, sig);                                              // modules.py:275-278
    if (P.duration_sampling) { head(M.conc, cq); head(M.rate, rq); }
    front_dwell<1>(io, P, sig, cq, rq, dbg, lane);
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) X[0][ft] += ldg4(W + M.pe_enc + c * 64 + 16 * ft + 4 * g);   // modules.py:80
#pragma unroll 1
    for (int l = 0; l < M.enc_layers; ++l) fft_block<1, 1, 16>(W, M.enc[l], X, reinterpret_cast<float*>(lds_raw), 0, lane);
    front_store<1>(io, dbg, X, lane, true);
}

// ---- f16x3: two chunks per wave, every product as three f16 MFMA terms, weights as f16 hi/lo units from L2, encoder attention
//      in registers (s2s_device_h.h: enc_attention_h) -- no LDS at all.  (Measured: the phase is bound by its instruction count,
//      about 6 k per chunk, like the decoder's; where the weights come from -- eight copies through the vector L1, or one copy
//      staged in LDS behind ten barriers per group -- made no difference to its 71 us per 16 chunks.)
template <int NQ>
__device__ __forceinline__ void frontend_h16(const ModelDev& M, const float* __restrict__ W, const FrontChunk (&io)[NQ],
                                             const ParamsDev& P, const DebugDev& dbg, const int lane, const float one) {
    const int g = lane >> 4, c = lane & 15;
    DIAG_DECL;
#define FDIAG(slot) DIAG_STAMP(32 + (slot))
    f32x4 X[NQ][4];
    front_embed<NQ>(M, W, io, lane, X);
    FDIAG(0);
#pragma unroll 1
    for (int i = 0; i < M.pre_layers; ++i) {                         // modules.py:74-77
        HL xb[NQ][2];
        f32x4 Y[NQ][4];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { xb[q][0] = split8(X[q][0], X[q][1], one); xb[q][1] = split8(X[q][2], X[q][3], one); }
        linear64_h<NQ>(W + M.pre_wh[i], W + M.pre_b[i], lane, xb, Y);
        relu_tiles<NQ>(Y);
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) X[q][mt] = Y[q][mt];
    }
    front_emb_in<NQ>(io, dbg, X, lane);
    front_store<NQ>(io, dbg, X, lane, false);                        // emb_out (debug only)
    FDIAG(1);
    {   // the three heads read emb_out only (modules.py:275-278, 197-225), so they run BEFORE the encoder blocks: emb_out is dead
        // by then instead of being carried (and spilled) through them
        HL Sb[NQ][2];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { Sb[q][0] = split8(X[q][0], X[q][1], one); Sb[q][1] = split8(X[q][2], X[q][3], one); }
        float sig[NQ], cq[NQ], rq[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { cq[q] = 1.0f; rq[q] = 1.0f; }
        auto head = [&](const MlpOff m, float (&out)[NQ]) {
            f32x4 hid[NQ][4];
            linear64_h<NQ>(W + m.w0h, W + m.b0, lane, Sb, hid);
            head_out<NQ>(W, m, hid, lane, out);
        };
        head(M.noise, sig);
        if (P.duration_sampling) { head(M.conc, cq); head(M.rate, rq); }
        FDIAG(4);
        front_dwell<NQ>(io, P, sig, cq, rq, dbg, lane);
    }
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) {
        const f32x4 pe = ldg4(W + M.pe_enc + c * 64 + 16 * ft + 4 * g);
#pragma unroll
        for (int q = 0; q < NQ; ++q) X[q][ft] += pe;                 // modules.py:80
    }
    FDIAG(5);
#pragma unroll 1
    for (int l = 0; l < M.enc_layers; ++l) {
        const LayerOff L = M.enc[l];
        const float* const wl = W + L.stream_h;                      // pack_layer: units 0-15 attention, 16-47 FFN
        f32x4 acc[NQ][4];
        HL x1b[NQ][2];
        enc_attention_h<NQ>(W, L, wl, X, acc, lane, one);
        FDIAG(2);
        enc_ffn_begin_h<NQ>(W, L, acc, X, x1b, lane, one);
        enc_ffn_half_h<NQ>(W, L, wl + 16 * 1024, 0, x1b, X, lane, one);
        enc_ffn_half_h<NQ>(W, L, wl + 32 * 1024, 2, x1b, X, lane, one);
        layer_norm64<NQ, true>(X, W + L.ln2g, W + L.ln2b, g);
        FDIAG(3);
    }
    front_store<NQ>(io, dbg, X, lane, true);
}

// ================================================================================ decoder
#ifndef DEC_WAVES
#define DEC_WAVES 8
#define DEC_NQ 2            // 16-column time tiles per wave: 8 waves x 2 x 16 = 256 >= 250
#endif
#define DEC_WPS (DEC_WAVES / 4)   // waves per SIMD
#define DEC_NKT 16
static constexpr int DEC_LDS_F32 = AttnLds<DEC_NKT>::BYTES;
static constexpr int DEC_LDS_H = AttnLdsH<DEC_NQ, DEC_WAVES, DEC_NKT>::BYTES;

// The decoder of ONE chunk, run by the 8 waves of a workgroup, in three pieces so that the fused kernel can request the next
// chunk's rows before it finishes the current one: dec_gather (length regulator), dec_blocks, dec_emit (projection, noise).
// slot: the chunk's frontend outputs -- enc_out [16][64], sigma [16] at +1024, dur [16] (int32) at +1040.

// ---- length regulator (modules.py:344-392) as a gather: row t copies encoder row i(t) = #{j : cum[j] <= t}; rows past
//      cum[15] are zero; crop at 250; then + position_enc (modules.py:136, also on the zero rows).  Two halves: dec_gather_issue
//      requests the rows, dec_gather_finish combines them -- whatever the caller puts in between runs under the loads.
struct GatherRaw {
    f32x4 e[DEC_NQ][4], pe[DEC_NQ][4];
    float sg[DEC_NQ];
    int idx[DEC_NQ];
};
__device__ __forceinline__ void dec_gather_issue(const ModelDev& M, const float* __restrict__ W, const float* __restrict__ slot,
                                                 const int wave, const int lane, GatherRaw& R) {
    const int g = lane >> 4, c = lane & 15;
    const int qt0 = DEC_NQ * wave;
    int cum[16];
    {
        // a lane-indexed (vector) load, then broadcasts: in the fused kernel the dwell counts were stored by another wave of
        // this launch, and a uniform-address load could be served from the scalar cache, which vector stores do not update
        const int dv = reinterpret_cast<const int*>(slot + 1040)[c];
        int run = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {          // a dwell past the crop at 250 (modules.py:386) acts like 251: no int32 overflow
            const int dj = __builtin_amdgcn_readlane(dv, j);
            run += dj < S2S_T_DEC + 1 ? dj : S2S_T_DEC + 1;
            cum[j] = run;
        }
    }
#pragma unroll
    for (int q = 0; q < DEC_NQ; ++q) {
        const int t = 16 * (qt0 + q) + c;
        int idx = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) idx += (cum[j] <= t) ? 1 : 0;
        R.idx[q] = idx;
        const int row = idx < 16 ? idx : 0;
        const float* er = slot + row * 64 + 4 * g;
        const float* pr = W + M.pe_dec + (t < S2S_T_DEC ? t : 0) * 64 + 4 * g;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) { R.e[q][ft] = ldg4(er + 16 * ft); R.pe[q][ft] = ldg4(pr + 16 * ft); }
        R.sg[q] = slot[1024 + row];
    }
}
__device__ __forceinline__ void dec_gather_finish(const GatherRaw& R, const int wave, const int lane, f32x4 (&X)[DEC_NQ][4],
                                                  float (&sig_ext)[DEC_NQ]) {
    const int c = lane & 15;
#pragma unroll
    for (int q = 0; q < DEC_NQ; ++q) {
        const int t = 16 * (DEC_NQ * wave + q) + c;
        const bool live = R.idx[q] < 16, real = t < S2S_T_DEC;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
            X[q][ft] = real ? ((live ? R.e[q][ft] : f32x4{0, 0, 0, 0}) + R.pe[q][ft]) : f32x4{0, 0, 0, 0};
        sig_ext[q] = live ? R.sg[q] : 0.0f;
    }
}

// stand-alone Decoder (TEST instance, dbg.dec_in): the chunk's decoder input comes from memory instead of the gather
__device__ __forceinline__ void dec_override(const float* __restrict__ dec_in, const long long dbg_idx, const int wave, const int lane,
                                             f32x4 (&X)[DEC_NQ][4]) {
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int q = 0; q < DEC_NQ; ++q) {
        const int t = 16 * (DEC_NQ * wave + q) + c;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
            X[q][ft] = t < S2S_T_DEC ? ldg4(dec_in + (dbg_idx * S2S_T_DEC + t) * 64 + 16 * ft + 4 * g) : f32x4{0, 0, 0, 0};
    }
}

// next_slot (MODE 1 only): the NEXT chunk's hand-off slot, copied into the LDS words behind the block's own image by the last
// layer's FFN weight staging step (one more load per lane before its barrier, one more LDS store after it), so that the gather
// that follows this chunk reads LDS instead of paying two dependent L2 round trips.
template <int MODE, bool EXACT = false>   // 0: f32-input MFMA block, 1: split-f16 block (s2s_device_h.h), 3: the same block with single f16 products
__device__ __forceinline__ void dec_blocks(const ModelDev& M, const float* __restrict__ W, f32x4 (&X)[DEC_NQ][4],
                                           char* __restrict__ lds_raw, const int wave, const int lane, const float one,
                                           unsigned long long* diag, const float* __restrict__ next_slot = nullptr) {
    const int qt0 = DEC_NQ * wave;
#pragma unroll 1
    for (int l = 0; l < M.dec_layers; ++l) {
        float* const slot_lds = reinterpret_cast<float*>(lds_raw + DEC_LDS_H), *const sv_lds = slot_lds + S2S_SLOT_FLOATS;
        if (MODE == 1) fft_block_h<DEC_NQ, DEC_WAVES, DEC_NKT, S2S_T_DEC, true, EXACT>(W, M.dec[l], X, lds_raw, qt0, wave, lane, one, diag,
                                                                          l == M.dec_layers - 1 ? next_slot : nullptr, slot_lds, sv_lds);
        else if (MODE == 3) fft_block_h<DEC_NQ, DEC_WAVES, DEC_NKT, S2S_T_DEC, false, EXACT>(W, M.dec[l], X, lds_raw, qt0, wave, lane, one, diag,
                                                                                      nullptr, slot_lds, sv_lds);
        else           fft_block<DEC_NQ, DEC_NKT, S2S_T_DEC>(W, M.dec[l], X, reinterpret_cast<float*>(lds_raw), qt0, lane, diag);
    }
}

// ---- out_linear + ReLU (modules.py:140-141).  Every lane group ends up with the row sums of all tiles (sum_g is an
//      all-reduce), so lane group q keeps tile q: ys = its sample (scaled units), se = its expanded sigma.
__device__ __forceinline__ void dec_project(const ModelDev& M, const float* __restrict__ W, const f32x4 (&X)[DEC_NQ][4],
                                            const float (&sig_ext)[DEC_NQ], const int lane, float& ys, float& se) {
    const int g = lane >> 4;
    f32x4 wo[4];
#pragma unroll
    for (int ft = 0; ft < 4; ++ft) wo[ft] = ldg4(W + M.out_w + 16 * ft + 4 * g);
    const float bo = W[M.out_b];
    ys = 0.0f; se = 0.0f;
#pragma unroll
    for (int q = 0; q < DEC_NQ; ++q) {
        float part = 0.0f;
#pragma unroll
        for (int ft = 0; ft < 4; ++ft)
#pragma unroll
            for (int r = 0; r < 4; ++r) part += X[q][ft][r] * wo[ft][r];
        const float v = relu1(sum_g(part) + bo);
        if (g == q) { ys = v; se = sig_ext[q]; }
    }
}

// ---- x165 (model.py:221), noise where != 0 (model.py:224-238), clamp (model.py:240): ONE pass of Philox + Box-Muller per
//      wave for its DEC_NQ <= 4 tiles.  inj_z01 / out_signal: the chunk's [250] rows.
__device__ __forceinline__ void dec_emit(const ModelDev& M, const float ys, const float se, const unsigned long long chunk,
                                         const ParamsDev& P, const float* __restrict__ inj_z01, float* __restrict__ out_signal,
                                         const DebugDev& dbg, const long long dbg_idx, const int wave, const int lane) {
    const int g = lane >> 4, c = lane & 15;
    const int t = 16 * (DEC_NQ * wave + g) + c;
    float y = __fmul_rn(ys, M.scale);
    if (g < DEC_NQ && t < S2S_T_DEC) {
        if (dbg.y_scaled) dbg.y_scaled[dbg_idx * S2S_T_DEC + t] = ys;
        if (P.noise_std > 0.0f) {
            float z;
            if (inj_z01) {
                z = inj_z01[t];
            } else {
                const u32x4 r = philox4x32_10((unsigned)chunk, (unsigned)(chunk >> 32),
                                              (unsigned)t | (S2S_KIND_NOISE << 16), 0, P.seed_lo, P.seed_hi);
                z = box_muller(r.x, r.y);
            }
            if (dbg.z01) dbg.z01[dbg_idx * S2S_T_DEC + t] = z;
            const float sd = P.noise_sampling
                                 ? __fmul_rn(__fmul_rn(fmaxf(se, P.min_noise), P.noise_std), M.scale)
                                 : P.noise_std;
            if (y != 0.0f) y = mul_then_add(z, sd, y);
        }
        store_stream(out_signal + t, fmaxf(y, 0.0f));
    }
}

// ONE launch per tile of chunks, nothing but the bases comes in and nothing but dwell counts and
// signal goes out.  A persistent 8-wave workgroup owns a contiguous range of the tile's chunks and walks it in groups of up
// to 8 * FNQ: wave w runs the one-wave frontend of chunks FNQ*w .. FNQ*w+FNQ-1 of the group (their encoder K/V in a slice of
// the decoder's K/V region, dead between chunks), then all 8 waves decode the group's chunks one after the other.  The
// frontend outputs of a group (4.2 KB per chunk) wait in the workgroup's own slots of `handoff` -- rewritten every group by the
// CU that reads them back, so the reads hit that XCD's L2.  The dirty lines do get written back once per group, though: an XCD's
// L2 keeps about 1.5 MB of rewritten data (tools/probes/l2_writeback_probe.hip) and 32 workgroups x 16 slots are 2.2 MB.
template <int MODE> struct Fused {
    static constexpr int FMODE = (MODE == 3) ? 1 : MODE;          // the reduced-precision decoder keeps the f16x3 frontend
    static constexpr int FNQ = (FMODE == 1) ? 2 : 1;              // chunks per frontend wave
    static constexpr int GROUP = DEC_WAVES * FNQ;
    static constexpr bool PF = (MODE == 1);                       // next chunk's slot prefetched into LDS (dec_blocks)
    static constexpr int LDS = (MODE == 0) ? DEC_LDS_F32 : DEC_LDS_H + (S2S_SLOT_FLOATS + S2S_SV_FLOATS + S2S_PROG_INTS + S2S_Z2_FLOATS) * 4;   // + next slot, small vectors, progress counters, 2nd zeros row
    // (the static arrays sit in front, largest alignment first; the dynamic region follows on its own 16-byte alignment.  The
    //  diagnostic build -- 3 KB of per-wave stamps -- fills the CU's 160 KB to the byte.)
    static_assert(LDS + S2S_STATIC_LDS_BYTES + 16 <= 160 * 1024, "LDS per workgroup: dynamic + the static arrays in front of it");
    static_assert(FMODE == 1 || DEC_WAVES * FrontLdsF32::BYTES <= LDS, "the f32 frontend waves' K/V images share the decoder's LDS");
};
#define S2S_MAX_GROUP (2 * DEC_WAVES)
// TEST = false is the production instance: no injected variates and no stage outputs, whose address arithmetic would otherwise
// sit (and spill) in the hot loop; the parity tests that inject or ask for stage outputs run the TEST = true instance of the
// same code.
// EXACT: the decoder attention's exact path (the online softmax, s2s_device_h.h) instead of "fast path, redone on overflow" -- a
// kernel instance of its own, so that neither path's registers and schedule depend on the other (split-f16 modes only).
template <int MODE, bool TEST, bool EXACT = false>
__global__ __launch_bounds__(DEC_WAVES * 64, DEC_WPS) void s2s_fused_kernel(
    const ModelDev M, const float* __restrict__ W, const uint8_t* __restrict__ bases,
    const long long* __restrict__ chunk_start, const uint8_t* __restrict__ n_valid, int n_chunks, long long first_chunk,
    ParamsDev P, const float* __restrict__ inj_g_, const float* __restrict__ inj_zdw_, const float* __restrict__ inj_z01_,
    float* __restrict__ handoff, int* __restrict__ out_dur, float* __restrict__ out_signal, DebugDev dbg_, long long dbg_base) {
    using F = Fused<MODE>;
    const float* const inj_g = TEST ? inj_g_ : nullptr;
    const float* const inj_zdw = TEST ? inj_zdw_ : nullptr;
    const float* const inj_z01 = TEST ? inj_z01_ : nullptr;
    DebugDev dbg = dbg_;
    if (!TEST) { dbg = DebugDev{}; dbg.diag = dbg_.diag; dbg.stats = dbg_.stats; }
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nb = S2S_T_ENC + M.k - 1;
    if constexpr (MODE != 0) att32_consts<AttnLdsH<DEC_NQ, DEC_WAVES, DEC_NKT>>(lds_raw, threadIdx.x, DEC_WAVES * 64);   // (visible after the group loop's first barrier)
    if constexpr (MODE != 0) {                                     // the attention loop's progress counters (prio_balance)
        if (threadIdx.x < S2S_PROG_INTS + S2S_Z2_FLOATS)              // (and the second zeros row behind them)
            reinterpret_cast<int*>(lds_raw + DEC_LDS_H + (S2S_SLOT_FLOATS + S2S_SV_FLOATS) * 4)[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) {                // production counters: redo count, entry stamps (kept in LDS, not in SGPRs, across the kernel)
        const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
        s2s_stats_lds[0] = 0;
        s2s_stats_lds[4] = (unsigned)c0; s2s_stats_lds[5] = (unsigned)(c0 >> 32);
        s2s_stats_lds[6] = (unsigned)r0; s2s_stats_lds[7] = (unsigned)(r0 >> 32);
    }
#ifdef S2S_TILEHIST
    for (int i = threadIdx.x; i < 2 * 2 * 64; i += DEC_WAVES * 64) s2s_hist_lds[i] = 0;
#endif
#ifdef S2S_DIAG
    for (int i = threadIdx.x; i < 8 * S2S_DIAG_SLOTS; i += DEC_WAVES * 64) s2s_diag_lds[i] = 0;
    const unsigned long long diag_c0 = __builtin_readcyclecounter(), diag_r0 = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
#endif
    float* const slot0 = handoff + (size_t)blockIdx.x * S2S_MAX_GROUP * S2S_SLOT_FLOATS;
    const int lo = (int)((long long)blockIdx.x * n_chunks / gridDim.x), hi = (int)((long long)(blockIdx.x + 1) * n_chunks / gridDim.x);
#pragma unroll 1
    for (int g0 = lo; g0 < hi; g0 += F::GROUP) {
        const int n_here = hi - g0 < F::GROUP ? hi - g0 : F::GROUP;
        float one = 1.0f;                  // opaque to the optimiser: see split2 in s2s_device_h.h
        asm volatile("" : "+s"(one));
        DIAG_DECL;
        __syncthreads();                   // the previous group's last block is done with the K/V region and with the slots
        {
            int lnf = lane;                // (opaque per group, like `ln` below: nothing lane-dependent is hoisted out of the loops and spilled)
            asm volatile("" : "+v"(lnf));
            auto chunk_of = [&](const int j, const int j_filler) {
                const bool live = j < n_here;
                const int b = g0 + (live ? j : j_filler);            // a filler repeats the wave's first chunk
                return FrontChunk{chunk_start ? bases + chunk_start[b] : bases + (size_t)b * nb, n_valid[b],
                                  (unsigned long long)(first_chunk + b), inj_g ? inj_g + (size_t)b * 16 : nullptr,
                                  inj_zdw ? inj_zdw + (size_t)b * 16 : nullptr, slot0 + j * S2S_SLOT_FLOATS,
                                  out_dur + (size_t)b * 16, dbg_base + b, live};
            };
            if constexpr (F::FMODE == 1) {
                if (n_here > DEC_WAVES) {                            // two chunks per wave
                    const FrontChunk io[2] = {chunk_of(2 * wave, 2 * wave), chunk_of(2 * wave + 1, 2 * wave)};
                    if (2 * wave < n_here) frontend_h16<2>(M, W, io, P, dbg, lnf, one);
                } else {                                             // a short group (small batches): one chunk per wave, more waves busy
                    const FrontChunk io[1] = {chunk_of(wave, 0)};
                    if (wave < n_here) frontend_h16<1>(M, W, io, P, dbg, lnf, one);
                }
            } else {
                const FrontChunk io[1] = {chunk_of(wave, 0)};
                if (wave < n_here) frontend_f32(M, W, io, P, lds_raw + wave * FrontLdsF32::BYTES, dbg, lnf);
            }
        }
        __syncthreads();                   // the group's slots are written (global stores: workgroup-scope release/acquire)
        DIAG_STAMP(7);                     // frontend phase and its two barriers
        f32x4 X[DEC_NQ][4];
        float sig_ext[DEC_NQ];
        {
            int lg = lane;                 // (opaque, like lnf / ln: the gather's per-lane addresses are not kept across the group loop)
            asm volatile("" : "+v"(lg));
            GatherRaw R;
            dec_gather_issue(M, W, slot0, wave, lg, R);
            dec_gather_finish(R, wave, lg, X, sig_ext);
            if (TEST && dbg.dec_in) dec_override(dbg.dec_in, dbg_base + g0, wave, lg, X);
        }
        DIAG_STAMP(8);
#pragma unroll 1
        for (int j = 0; j < ((S2S_ABL & 65536) ? 0 : n_here); ++j) {      // (65536: timing of the frontend phase alone)
            const int b = g0 + j;
            // (after the group's last chunk everything "next" is that chunk again, computed and thrown away: no branches here,
            // the register allocator spills values that live from one conditional block to another)
            const float* next = slot0 + (j + 1 < n_here ? j + 1 : j) * S2S_SLOT_FLOATS;
            dec_blocks<MODE, EXACT>(M, W, X, lds_raw, wave, lane, one, dbg.diag, F::PF ? next : nullptr);
            DIAG_STAMP(15);   // (time inside the blocks is accounted by their own stamps)
            // (an opaque copy of the lane id: the per-lane addresses below are recomputed per chunk, a handful of VALU
            // instructions, instead of being hoisted out of the loop and spilled across the blocks -- a scratch reload here
            // would also wait for every load issued before it)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            float ys, se;
            dec_project(M, W, X, sig_ext, ln, ys, se);
            // the next chunk's rows are requested before this one's noise is drawn, from the copy of its slot that the last
            // block left in LDS (f16x3), or from the slot itself: two dependent L2 round trips behind Philox + Box-Muller
            GatherRaw R;
            if constexpr (F::PF) dec_gather_issue(M, W, reinterpret_cast<const float*>(lds_raw + DEC_LDS_H), wave, ln, R);
            else                 dec_gather_issue(M, W, next, wave, ln, R);
            dec_emit(M, ys, se, (unsigned long long)(first_chunk + b), P, inj_z01 ? inj_z01 + (size_t)b * S2S_T_DEC : nullptr,
                     out_signal + (size_t)b * S2S_T_DEC, dbg, dbg_base + b, wave, ln);
            dec_gather_finish(R, wave, ln, X, sig_ext);
            if (TEST && dbg.dec_in) dec_override(dbg.dec_in, dbg_base + g0 + (j + 1 < n_here ? j + 1 : j), wave, ln, X);
            DIAG_STAMP(9);
        }
    }
#ifdef S2S_DIAG
    // slots 16 / 17: this wave's whole-kernel time on the shader clock (s_memtime) and on the constant 100 MHz clock (s_memrealtime):
    // their ratio is the clock the SIMDs actually ran at (MI355X_MICROARCH.md, DVFS)
    DIAG_COUNT(16, __builtin_readcyclecounter() - diag_c0);
    DIAG_COUNT(17, __builtin_amdgcn_s_memrealtime() - diag_r0);
    DIAG_COUNT(18, 1ull);
    __syncthreads();
    if (dbg.diag)
        for (int i = threadIdx.x; i < 8 * S2S_DIAG_SLOTS; i += DEC_WAVES * 64)
            if (s2s_diag_lds[i]) atomicAdd(dbg.diag + i, s2s_diag_lds[i]);        // [wave][slot], summed over the workgroups
#endif
    __syncthreads();                       // every wave's redo count is in LDS
#ifdef S2S_TILEHIST
    if (dbg.diag)
        for (int i = threadIdx.x; i < 2 * 2 * 64; i += DEC_WAVES * 64)
            if (s2s_hist_lds[i]) atomicAdd(dbg.diag + i, (unsigned long long)s2s_hist_lds[i]);
#endif
    if (threadIdx.x == 0 && dbg.stats) {
        const unsigned long long c0 = ((unsigned long long)s2s_stats_lds[5] << 32) | s2s_stats_lds[4];
        const unsigned long long r0 = ((unsigned long long)s2s_stats_lds[7] << 32) | s2s_stats_lds[6];
        if (s2s_stats_lds[0]) atomicAdd(dbg.stats + S2S_STAT_REDO, (unsigned long long)s2s_stats_lds[0]);
        atomicAdd(dbg.stats + S2S_STAT_CYCLES, __builtin_readcyclecounter() - c0);
        atomicAdd(dbg.stats + S2S_STAT_TICKS, __builtin_amdgcn_s_memrealtime() - r0);
        atomicAdd(dbg.stats + S2S_STAT_WGS, 1ull);
    }
}

// ================================================================================ export
// per-chunk count of non-zero samples (model.py:286 strips by value)
__global__ __launch_bounds__(256) void s2s_count_kernel(const float* __restrict__ signal, int B, int* __restrict__ counts) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    int n = 0;
    for (int t = lane; t < S2S_T_DEC; t += 64) n += signal[(size_t)b * S2S_T_DEC + t] != 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0) counts[b] = n;
}

// exclusive scan of counts[B] -> offs[B+1] (int64), one workgroup walking the array, 8 consecutive elements per thread and step
// (131,072 chunk counts = 16 steps)
__global__ __launch_bounds__(1024) void s2s_scan_kernel(const int* __restrict__ counts, int B, long long* __restrict__ offs) {
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 8192) {
        const int i0 = base + 8 * tid;
        int c[8];
        if (i0 + 8 <= B) {                                     // (the workspace is 16-byte aligned, i0 a multiple of 8)
            const int4 a = *reinterpret_cast<const int4*>(counts + i0), b = *reinterpret_cast<const int4*>(counts + i0 + 4);
            c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w; c[4] = b.x; c[5] = b.y; c[6] = b.z; c[7] = b.w;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) c[k] = (i0 + k < B) ? counts[i0 + k] : 0;
        }
        long long v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += c[k];
        long long x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const long long y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        long long pre = carry_s;
        for (int j = 0; j < w; ++j) pre += wsum[j];
        long long run = pre + x - v;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i0 + k < B) offs[i0 + k] = run;
            run += c[k];
        }
        __syncthreads();
        if (tid == 1023) carry_s = pre + x;
        __syncthreads();
    }
    if (tid == 0) offs[B] = carry_s;
}

__global__ __launch_bounds__(256) void s2s_read_offsets_kernel(const long long* __restrict__ offs, const int* __restrict__ read_first,
                                                               int R, long long* __restrict__ out_offsets) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r <= R) out_offsets[r] = offs[read_first[r]];
}

__global__ __launch_bounds__(256) void s2s_compact_kernel(const float* __restrict__ signal, int B, const long long* __restrict__ offs,
                                                          const int* __restrict__ read_first, int R, float* __restrict__ out_pa,
                                                          short* __restrict__ out_dac, long long capacity, float dig, float range,
                                                          float offset, int rna) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    long long pos = offs[b];
    long long r_lo = 0, r_hi = 0;
    if (rna && out_dac) {                      // read that owns chunk b: last r with read_first[r] <= b
        int lo = 0, hi = R;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (read_first[mid] <= b) lo = mid; else hi = mid; }
        r_lo = offs[read_first[lo]];
        r_hi = offs[read_first[lo + 1]];
    }
    for (int t0 = 0; t0 < S2S_T_DEC; t0 += 64) {
        const int t = t0 + lane;
        const float v = (t < S2S_T_DEC) ? signal[(size_t)b * S2S_T_DEC + t] : 0.0f;
        const bool keep = v != 0.0f;
        const unsigned long long mask = __ballot(keep);
        const long long dst = pos + __popcll(mask & ((1ull << lane) - 1ull));
        if (keep && dst < capacity) {
            if (out_pa) out_pa[dst] = v;
            if (out_dac) {
                // signal_io.py:135-138: float32 ops, no contraction; round half-to-even; int16 wrap
                const float raw = rintf(__fsub_rn(__fdiv_rn(__fmul_rn(v, dig), range), offset));
                const short s = (short)(int)fminf(fmaxf(raw, -2147483648.0f), 2147483520.0f);
                out_dac[rna ? (r_lo + (r_hi - 1 - dst)) : dst] = s;
            }
        }
        pos += __popcll(mask);
    }
}

// ---- StreamVByte encoders of the output containers (codecs.py states the formats): one 256-thread workgroup per row.
// VARIANT 32 (slow5 svb-zd): [u32 n][(n+3)/4 control bytes, 2 bits per value][data: 1..4 bytes per value] of the zig-zag
// deltas widened to 32 bits; VARIANT 16 (pod5 VBZ before zstd): [(n+7)/8 control bytes, 1 bit per value][1..2 bytes per value]
// of the zig-zag deltas in 16-bit arithmetic.  A thread owns 8 consecutive values = one (svb16) or two (svb32) control bytes;
// their data bytes are placed by a block-wide prefix sum with a running carry.  WRITE = false only sizes the row.
template <int VARIANT, bool WRITE>
__global__ __launch_bounds__(256) void s2s_svb_kernel(const short* __restrict__ samples, const long long* __restrict__ read_offs,
                                                      const int* __restrict__ row_read, const int* __restrict__ row_index,
                                                      long long row_samples, int* __restrict__ row_bytes,
                                                      const long long* __restrict__ out_offs, unsigned char* __restrict__ out,
                                                      long long capacity) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int rd = row_read[row];
    const long long r0 = read_offs[rd], rn = read_offs[rd + 1] - r0;
    const long long lo = (long long)row_index[row] * row_samples;
    long long n = rn - lo;
    if (n > row_samples) n = row_samples;
    if (n <= 0) {                                            // a candidate row the read turned out not to need
        if (!WRITE && tid == 0) row_bytes[row] = 0;
        return;
    }
    const short* x = samples + r0 + lo;
    const long long nkeys = (VARIANT == 32) ? (n + 3) / 4 : (n + 7) / 8;
    const int hdr = (VARIANT == 32) ? 4 : 0;
    unsigned char* o = nullptr;
    if (WRITE) {
        const long long base = out_offs[row];
        if (out_offs[row + 1] > capacity) return;            // does not fit: skipped, and s2s_svb_check_kernel reports it
        o = out + base;
        if (VARIANT == 32 && tid == 0) { const unsigned nn = (unsigned)n; o[0] = nn; o[1] = nn >> 8; o[2] = nn >> 16; o[3] = nn >> 24; }
    }
    if (tid == 0) carry_s = 0;
    __syncthreads();
    // a thread's nine samples of a step (its eight + the one before them) are loaded one step ahead: the steps of a row are a
    // serial chain (the carry), and without the prefetch each one began with a full memory round trip
    short raw[9];
    auto fetch = [&](long long t0) {
        const long long j0 = t0 + 8 * (long long)tid;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const long long j = j0 + i - 1;
            raw[i] = (j >= 0 && j < n) ? x[j] : (short)0;     // (the delta of a row's first value is against 0)
        }
    };
    fetch(0);
    for (long long t0 = 0; t0 < n; t0 += 2048) {
        const long long j0 = t0 + 8 * (long long)tid;
        unsigned v[8];
        int len[8], tot = 0;
        unsigned key = 0;
        int prev = raw[0];
        short now[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) now[i] = raw[i + 1];
        if (t0 + 2048 < n) fetch(t0 + 2048);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            len[i] = 0; v[i] = 0;
            if (j0 + i < n) {
                const int cur = now[i];
                if (VARIANT == 32) {
                    const int d = cur - prev;
                    v[i] = ((unsigned)d << 1) ^ (unsigned)(d >> 31);
                    len[i] = 1 + (v[i] > 0xFFu) + (v[i] > 0xFFFFu) + (v[i] > 0xFFFFFFu);
                    key |= (unsigned)(len[i] - 1) << (2 * i);
                } else {
                    const short d = (short)((unsigned short)cur - (unsigned short)prev);
                    v[i] = (unsigned short)(((int)d + (int)d) ^ ((int)d >> 15));
                    len[i] = 1 + (v[i] > 0xFFu);
                    key |= (unsigned)(len[i] - 1) << i;
                }
                prev = cur;
                tot += len[i];
            }
        }
        int incl = tot;                                       // block-wide exclusive prefix of `tot`
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off, 64); if (lane >= off) incl += y; }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int pre = carry_s + incl - tot;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        if (WRITE && j0 < n) {
            if (VARIANT == 32) {
                o[hdr + j0 / 4] = (unsigned char)key;
                if (j0 + 4 < n) o[hdr + j0 / 4 + 1] = (unsigned char)(key >> 8);
            } else {
                o[j0 / 8] = (unsigned char)key;
            }
            unsigned char* d = o + hdr + nkeys + pre;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                for (int k = 0; k < len[i]; ++k) *d++ = (unsigned char)(v[i] >> (8 * k));
        }
        __syncthreads();
        if (tid == 255) carry_s = pre + tot;
        __syncthreads();
    }
    if (!WRITE && tid == 0) row_bytes[row] = (int)(hdr + nkeys + carry_s);
}

// s2s_svb_encode's overflow report: rows that do not fit are skipped by the WRITE pass; the total then comes back negative
__global__ void s2s_svb_check_kernel(long long* __restrict__ out_offs, int n, long long capacity) {
    if (threadIdx.x == 0 && blockIdx.x == 0 && out_offs[n] > capacity) out_offs[n] = -out_offs[n];
}

__global__ void s2s_philox_kernel(unsigned seed_lo, unsigned seed_hi, unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                  int n, unsigned* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 r = philox4x32_10(c0 + (unsigned)i, c1, c2, c3, seed_lo, seed_hi);
    out[4 * i + 0] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

// ================================================================================ host side
namespace {

thread_local std::string g_create_error;

struct EventPair { hipEvent_t a, b; int chunks; };

}  // namespace

struct s2s_handle {
    s2s_config cfg;
    int device = 0;
    ModelDev model;
    char* slab = nullptr;             // ONE device allocation made by s2s_create: weights, hand-off slots, first scratch buffers
    size_t slab_bytes = 0;
    float* d_arena = nullptr;
    size_t arena_floats = 0;
    int tile = 0;                     // chunks per launch pair
    int n_wg = 256;                   // decoder grid: persistent workgroups, one per CU
    float* handoff = nullptr;         // frontend -> decoder slots: [n_wg][S2S_MAX_GROUP][S2S_SLOT_FLOATS] (L2-resident)
    int* ws_counts = nullptr;         // export scratch, grown on demand outside of launches
    long long* ws_offs = nullptr;
    int ws_export_cap = 0;
    int* ws_svb = nullptr;            // s2s_svb_encode scratch: bytes per row
    int ws_svb_cap = 0;
    bool profiling = false;
    unsigned long long* d_diag = nullptr;   // S2S_DIAG builds: [8 waves][48] per-phase wave-cycle sums (S2S_TILEHIST: the tile histogram)
    unsigned long long* d_stats = nullptr;  // S2S_STAT_* counters of the predict kernel (inside the slab; s2s_stats_read)
    long long stat_chunks = 0;              // chunks launched since the last s2s_stats_read
    int attn_exact = 0;                     // decoder attention path of the split-f16 modes (s2s_set_attention_path)
    long long stat_exact_chunks = 0;        // ... chunks launched on the exact path since the last s2s_stats_read
    double calib_redo_rate = -1.0;          // share of the calibration launch's softmax runs that overflowed the fast path (-1: not calibrated)
    std::vector<EventPair> events;
    std::string err;
};

namespace {

int fail(s2s_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(h, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((h), S2S_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Scratch that has outgrown its first home inside the handle's slab is an allocation of its own.
void free_scratch(s2s_handle* h, void* p) {
    const char* c = static_cast<const char*>(p);
    if (p && !(h->slab && c >= h->slab && c < h->slab + h->slab_bytes)) (void)hipFree(p);
}

// Makes the handle's device current for the duration of a call and restores the caller's device afterwards.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
        else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

size_t layer_floats() { return 4 * (64 * 64 + 64) + 2 * 64 + (256 * 64 + 256) + (64 * 256 + 64) + 2 * 64; }
size_t mlp_floats() { return 64 * 64 + 64 + 64 + 1; }

const char* check_cfg(const s2s_config* c) {
    if (!c) return "config is NULL";
    if (c->seq_kmer < 1 || c->seq_kmer > 16) return "seq_kmer must be 1..16";
    if (c->max_dna_len != S2S_T_ENC) return "max_dna_len must be 16";
    if (c->max_signal_len != S2S_T_DEC) return "max_signal_len must be 250";
    if (c->dmodel != S2S_DMODEL) return "dmodel must be 64";
    if (c->dff != S2S_DFF) return "dff must be 256";
    if (c->n_heads != S2S_HEADS) return "n_heads must be 8";
    if (c->encoder_layers < 1 || c->encoder_layers > S2S_MAX_LAYERS) return "encoder_layers must be 1..4";
    if (c->decoder_layers < 1 || c->decoder_layers > S2S_MAX_LAYERS) return "decoder_layers must be 1..4";
    if (c->pre_layers < 0 || c->pre_layers > S2S_MAX_LAYERS) return "pre_layers must be 0..4";
    if (c->compute_mode != S2S_MODE_F32 && c->compute_mode != S2S_MODE_F16X3 && c->compute_mode != S2S_MODE_F16)
        return "compute_mode must be S2S_MODE_F32, S2S_MODE_F16X3 or S2S_MODE_F16";
    return nullptr;
}

// Arena builder: every piece starts on a 16-byte boundary (float4 loads).
struct Arena {
    std::vector<float> v;
    int put(const float* p, size_t n) {
        while (v.size() % 4) v.push_back(0.0f);
        const int off = (int)v.size();
        v.insert(v.end(), p, p + n);
        return off;
    }
    // W [M][K] row-major -> A fragments [M/16][K/16][64 lanes][4]; lane (g, i): W[perm(16mt+i)][16kt+4g+r]
    int put_afrag(const float* W, int M, int K, bool qk_perm) {
        std::vector<float> t((size_t)M * K);
        for (int mt = 0; mt < M / 16; ++mt)
            for (int kt = 0; kt < K / 16; ++kt)
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, i = lane & 15;
                    const int row = 16 * mt + (qk_perm ? perm16(i) : i);
                    for (int r = 0; r < 4; ++r)
                        t[(((size_t)mt * (K / 16) + kt) * 64 + lane) * 4 + r] = W[(size_t)row * K + 16 * kt + 4 * g + r];
                }
        return put(t.data(), t.size());
    }
    int put_bias_perm(const float* b, int M) {
        std::vector<float> t(M);
        for (int i = 0; i < M; ++i) t[i] = b[16 * (i / 16) + perm16(i % 16)];
        return put(t.data(), t.size());
    }
    // packed row i = 4g'+r' of a q/k tile takes natural row 8*(r'>>1) + 2g' + (r'&1): after the MFMA,
    // accumulator registers {0,1} hold head 0 (d = 2g, 2g+1) and {2,3} head 1 of the pair.
    static int perm16(int i) { return 8 * ((i & 3) >> 1) + 2 * (i >> 2) + (i & 1); }
};

const float* take(const float*& p, size_t n) { const float* q = p; p += n; return q; }

// w = hi + lo with hi = f16(w), lo = f16(w - hi) (round to nearest even), for a whole matrix at once.  The host compiler turns a
// plain _Float16 cast into a library call unless F16C code generation is on, and s2s_create converts half a million weights:
// the F16C instance does eight per instruction and is picked at run time where the CPU has it.
struct SplitF16 {
    std::vector<_Float16> hi, lo;
};
void split_f16_generic(const float* w, size_t n, _Float16* hi, _Float16* lo) {
    for (size_t i = 0; i < n; ++i) {
        hi[i] = (_Float16)w[i];
        lo[i] = (_Float16)(w[i] - (float)hi[i]);
    }
}
__attribute__((target("avx,f16c"))) void split_f16_f16c(const float* w, size_t n, _Float16* hi, _Float16* lo) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const __m256 x = _mm256_loadu_ps(w + i);
        const __m128i h = _mm256_cvtps_ph(x, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        const __m128i l = _mm256_cvtps_ph(_mm256_sub_ps(x, _mm256_cvtph_ps(h)), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(hi + i), h);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(lo + i), l);
    }
    for (; i < n; ++i) {
        const __m128i h = _mm_cvtps_ph(_mm_set_ss(w[i]), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        const float back = _mm_cvtss_f32(_mm_cvtph_ps(h));
        const __m128i l = _mm_cvtps_ph(_mm_set_ss(w[i] - back), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        const unsigned short hb = (unsigned short)_mm_extract_epi16(h, 0), lb = (unsigned short)_mm_extract_epi16(l, 0);
        std::memcpy(hi + i, &hb, 2);
        std::memcpy(lo + i, &lb, 2);
    }
}
SplitF16 split_f16(const float* w, size_t n) {
    SplitF16 s;
    s.hi.resize(n);
    s.lo.resize(n);
    static const bool fast = __builtin_cpu_supports("f16c") && __builtin_cpu_supports("avx");
    (fast ? split_f16_f16c : split_f16_generic)(w, n, s.hi.data(), s.lo.data());
    return s;
}

LayerOff pack_layer(Arena& A, const float*& p) {
    LayerOff L;
    const float* wq = take(p, 4096); const float* bq = take(p, 64);
    const float* wk = take(p, 4096); const float* bk = take(p, 64);
    const float* wv = take(p, 4096); const float* bv = take(p, 64);
    const float* wfc = take(p, 4096); const float* bfc = take(p, 64);
    const float* ln1g = take(p, 64); const float* ln1b = take(p, 64);
    const float* w1 = take(p, 256 * 64); const float* b1 = take(p, 256);
    const float* w2 = take(p, 64 * 256); const float* b2 = take(p, 64);
    const float* ln2g = take(p, 64); const float* ln2b = take(p, 64);
    // weight stream in the order fft_block consumes it (one unit = 4 fragments = 4 KiB):
    //   K/V phase : per head pair p: Wk rows of p (k-tiles 0..3), Wv rows of p
    //   attention : per pair p: Wq rows of p, Wfc columns of p (m-tiles 0..3)
    //   FFN       : per 64-wide hidden slice hc: W1 m-tiles 4hc..4hc+3, then W2 m-tiles 0..3 x k-tiles 4hc..4hc+3
    std::vector<float> st;
    st.reserve(4 * 4096 + 2 * 256 * 64 + 1024);
    auto frag = [&](const float* Wm, int K, int mt, int kt, bool perm) {
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, i = lane & 15;
            const int row = 16 * mt + (perm ? Arena::perm16(i) : i);
            for (int r = 0; r < 4; ++r) st.push_back(Wm[(size_t)row * K + 16 * kt + 4 * g + r]);
        }
    };
    for (int p_ = 0; p_ < 4; ++p_) {
        for (int kt = 0; kt < 4; ++kt) frag(wk, 64, p_, kt, true);
        for (int kt = 0; kt < 4; ++kt) frag(wv, 64, p_, kt, false);
    }
    for (int p_ = 0; p_ < 4; ++p_) {
        for (int kt = 0; kt < 4; ++kt) frag(wq, 64, p_, kt, true);
        for (int mt = 0; mt < 4; ++mt) frag(wfc, 64, mt, p_, false);
    }
    for (int hc = 0; hc < 4; ++hc) {
        for (int mt = 0; mt < 4; ++mt)
            for (int kt = 0; kt < 4; ++kt) frag(w1, 64, 4 * hc + mt, kt, false);
        for (int mt = 0; mt < 4; ++mt)
            for (int kt = 0; kt < 4; ++kt) frag(w2, 256, mt, 4 * hc + kt, false);
    }
    st.resize(st.size() + 1024, 0.0f);       // the block's last prefetch reads one unit past its stream
    L.stream = A.put(st.data(), st.size());
    // the same units for the split-f16 block: per (m-tile, k-block of 32) a hi and a lo fragment of
    // 64 lanes x 8 halves; lane (g, i), element j: W[16mt + i][kbase + 16(j>>2) + 4g + (j&3)].
    // Wq and bq go in pre-multiplied by log2(e)/sqrt(d_k): Q leaves its GEMM in the units the softmax wants (layers.py:32-33).
    const float c1 = 1.4426950408889634f * 0.35355339059327373f;
    std::vector<float> wq_s(wq, wq + 4096), bq_s(bq, bq + 64);
    for (float& v : wq_s) v *= c1;
    for (float& v : bq_s) v *= c1;
    std::vector<_Float16> sh;
    sh.reserve(2 * (4 * 4096 + 2 * 256 * 64) + 4 * 2048);
    const SplitF16 hk = split_f16(wk, 4096), hv = split_f16(wv, 4096), hq = split_f16(wq_s.data(), 4096), hfc = split_f16(wfc, 4096);
    const SplitF16 h1 = split_f16(w1, 256 * 64), h2 = split_f16(w2, 64 * 256);
    auto frag_h = [&](const SplitF16& Wm, int K, int mt, int kbase, bool lo) {
        const _Float16* src = lo ? Wm.lo.data() : Wm.hi.data();
        for (int lane = 0; lane < 64; ++lane) {
            const int g = lane >> 4, i = lane & 15;
            for (int j = 0; j < 8; ++j) sh.push_back(src[(size_t)(16 * mt + i) * K + kbase + 16 * (j >> 2) + 4 * g + (j & 3)]);
        }
    };
    auto unit_h = [&](const SplitF16& Wm, int K, int mt, int kbase) {      // [kb0 hi][kb0 lo][kb1 hi][kb1 lo]
        for (int kb = 0; kb < 2; ++kb) { frag_h(Wm, K, mt, kbase + 32 * kb, false); frag_h(Wm, K, mt, kbase + 32 * kb, true); }
    };
    for (int p_ = 0; p_ < 4; ++p_) { unit_h(hk, 64, p_, 0); unit_h(hv, 64, p_, 0); }
    for (int u = 0; u < 2; ++u) {
        unit_h(hq, 64, 2 * u, 0);
        unit_h(hq, 64, 2 * u + 1, 0);
        for (int mt = 0; mt < 4; ++mt) { frag_h(hfc, 64, mt, 32 * u, false); frag_h(hfc, 64, mt, 32 * u, true); }
    }
    for (int hc = 0; hc < 4; ++hc) {
        for (int mt = 0; mt < 4; ++mt) unit_h(h1, 64, 4 * hc + mt, 0);
        for (int mt = 0; mt < 4; ++mt) unit_h(h2, 256, mt, 64 * hc);
    }
    sh.resize(sh.size() + 4 * 2048, (_Float16)0.0f);   // the FFN ring runs three units past the end of the stream
    {
        std::vector<float> raw(sh.size() / 2);
        std::memcpy(raw.data(), sh.data(), sh.size() * sizeof(_Float16));
        L.stream_h = A.put(raw.data(), raw.size());
        // hi-only stream for the single-product mode: every even 512-half fragment of the stream above
        std::vector<_Float16> sf;
        for (size_t f0 = 0; f0 + 1024 <= sh.size(); f0 += 1024) sf.insert(sf.end(), sh.begin() + f0, sh.begin() + f0 + 512);
        std::vector<float> rawf(sf.size() / 2);
        std::memcpy(rawf.data(), sf.data(), sf.size() * sizeof(_Float16));
        L.stream_f = A.put(rawf.data(), rawf.size());
    }
    L.bq_nat = A.put(bq_s.data(), 64);         // (f16 blocks only; the f32 block reads L.bq)
    L.bk_nat = A.put(bk, 64);
    L.bq = A.put_bias_perm(bq, 64);
    L.bk = A.put_bias_perm(bk, 64);
    L.bv = A.put(bv, 64);
    L.bfc = A.put(bfc, 64);
    L.b1 = A.put(b1, 256);
    L.b2 = A.put(b2, 64);
    L.ln1g = A.put(ln1g, 64); L.ln1b = A.put(ln1b, 64);
    L.ln2g = A.put(ln2g, 64); L.ln2b = A.put(ln2b, 64);
    return L;
}

// a 64x64 Linear as 4 f16 units (one per m-tile): [kb0 hi][kb0 lo][kb1 hi][kb1 lo], same fragment order as pack_layer
int pack_linear64_h(Arena& A, const float* Wm) {
    const SplitF16 hw = split_f16(Wm, 4096);
    std::vector<_Float16> sh;
    sh.reserve(2 * 4096);
    for (int mt = 0; mt < 4; ++mt)
        for (int kb = 0; kb < 2; ++kb)
            for (int lo = 0; lo < 2; ++lo) {
                const _Float16* src = lo ? hw.lo.data() : hw.hi.data();
                for (int lane = 0; lane < 64; ++lane) {
                    const int g = lane >> 4, i = lane & 15;
                    for (int j = 0; j < 8; ++j) sh.push_back(src[(size_t)(16 * mt + i) * 64 + 32 * kb + 16 * (j >> 2) + 4 * g + (j & 3)]);
                }
            }
    std::vector<float> raw(sh.size() / 2);
    std::memcpy(raw.data(), sh.data(), sh.size() * sizeof(_Float16));
    return A.put(raw.data(), raw.size());
}

MlpOff pack_mlp(Arena& A, const float*& p) {
    MlpOff m;
    const float* w0 = take(p, 4096); const float* b0 = take(p, 64);
    const float* w3 = take(p, 64);   const float* b3 = take(p, 1);
    m.w0 = A.put_afrag(w0, 64, 64, false); m.b0 = A.put(b0, 64);
    m.w0h = pack_linear64_h(A, w0);
    m.w3 = A.put(w3, 64); m.b3 = A.put(b3, 1);
    return m;
}

ParamsDev to_dev(const s2s_params* p) {
    ParamsDev d;
    d.dwell_mean = p->dwell_mean; d.dwell_std = p->dwell_std; d.noise_std = p->noise_std;
    d.min_noise = p->min_noise; d.min_duration = p->min_duration;
    d.noise_sampling = p->noise_sampling; d.duration_sampling = p->duration_sampling;
    d.seed_lo = (unsigned)p->seed; d.seed_hi = (unsigned)(p->seed >> 32);
    return d;
}

}  // namespace

extern "C" {

size_t s2s_blob_floats(const s2s_config* c) {
    if (check_cfg(c)) return 0;
    return (size_t)16 * 64 + (size_t)64 * 5 * c->seq_kmer + 64 + (size_t)c->pre_layers * (4096 + 64) +
           (size_t)(c->encoder_layers + c->decoder_layers) * layer_floats() + 3 * mlp_floats() + (size_t)250 * 64 + 64 + 1;
}

static int predict_impl(s2s_handle* h, void* stream_, const uint8_t* bases, const int64_t* chunk_start, const uint8_t* n_valid, int64_t first_global_chunk,
                        int32_t B, const s2s_params* params, const float* inject_g, const float* inject_zdw,
                        const float* inject_z01, float* out_signal, int32_t* out_dur, const s2s_debug* dbg);

// Which softmax path the split-f16 decoder tries first is a property of the WEIGHTS: one launch of 512 pseudo-random chunks with the
// default samplers on the fast path counts the heads it had to redo (the production counters); above S2S_ATTENTION_REDO_THRESHOLD the handle starts
// every head on the exact path (s2s_device_h.h: the online softmax as its own kernel instance).  A fixed input, so the same weights always get the same
// answer, on any device.  The export scratch inside the slab holds the launch's buffers.
static int calibrate_attention(s2s_handle* h) {
    if (h->cfg.compute_mode == S2S_MODE_F32) return S2S_OK;
    const int B = 512, nb = S2S_T_ENC + h->cfg.seq_kmer - 1;
    std::vector<uint8_t> host((size_t)B * nb + B);
    uint32_t x = 0x9E3779B9u;
    for (size_t i = 0; i < (size_t)B * nb; ++i) { x = x * 1664525u + 1013904223u; host[i] = "ACGT"[x >> 30]; }
    for (int i = 0; i < B; ++i) host[(size_t)B * nb + i] = S2S_T_ENC;
    char* base = reinterpret_cast<char*>(h->ws_offs);               // (5 * 32768 + 1) * 8 bytes
    float* sig = reinterpret_cast<float*>(base);
    int32_t* dur = reinterpret_cast<int32_t*>(base + (size_t)B * S2S_T_DEC * 4);
    uint8_t* d_bases = reinterpret_cast<uint8_t*>(base + (size_t)B * S2S_T_DEC * 4 + (size_t)B * 16 * 4);
    static_assert((size_t)512 * S2S_T_DEC * 4 + 512 * 16 * 4 + 512 * 32 + 512 <= (size_t)(5 * 32768 + 1) * 8, "calibration buffers fit the export scratch");
    HIP_TRY(h, hipMemcpy(d_bases, host.data(), host.size(), hipMemcpyHostToDevice));
    const s2s_params P = {12.5f, 0.0f, 2.0f, 0.0f, 3.0f, 1, 1, 0};
    h->attn_exact = 0;
    // (an empty s2s_debug selects the TEST instance of the kernel -- same arithmetic, same counters: the production instance's
    // first dispatch and its average in a profile stay those of the caller's launches)
    s2s_debug none;
    std::memset(&none, 0, sizeof none);
    const int rc = predict_impl(h, nullptr, d_bases, nullptr, d_bases + (size_t)B * nb, 0, B, &P, nullptr, nullptr, nullptr, sig, dur, &none);
    if (rc != S2S_OK) return rc;
    uint64_t st[10];
    const int rs = s2s_stats_read(h, st);
    if (rs != S2S_OK) return rs;
    h->calib_redo_rate = st[1] ? (double)st[2] / (double)st[1] : 0.0;
    // (measured: a redone head also holds its seven partner waves at the next barrier, so a redo share r costs ~ 235 k x r cycles per
    //  chunk while r is small -- 6.4 %: + 14.7 k -- and 153 k x r once most heads redo; the exact instance costs + 11.4 k whatever the
    //  weights (201.6 k against 190.2 k, profiles/r05/attention_paths.txt): they break even at r = 5 - 5.5 % in cycles; in
    //  chunks per second -- the clock follows the operands -- the two are level at 6.4 %)
    h->attn_exact = h->calib_redo_rate > S2S_ATTENTION_REDO_THRESHOLD ? 1 : 0;
    return S2S_OK;
}

double s2s_attention_redo_threshold(void) { return S2S_ATTENTION_REDO_THRESHOLD; }

int s2s_create(const s2s_config* cfg, const void* blob, size_t blob_bytes, int device, s2s_handle** out) {
    if (!out) return fail(nullptr, S2S_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (const char* m = check_cfg(cfg)) return fail(nullptr, S2S_ERR_ARG, m);
    if (!blob) return fail(nullptr, S2S_ERR_ARG, "weights_blob is NULL");
    if (blob_bytes != s2s_blob_floats(cfg) * sizeof(float)) {
        char buf[160];
        snprintf(buf, sizeof buf, "weight blob is %zu bytes, expected %zu", blob_bytes, s2s_blob_floats(cfg) * sizeof(float));
        return fail(nullptr, S2S_ERR_BLOB, buf);
    }
    int ndev = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(nullptr, S2S_ERR_ARG, "no such HIP device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(nullptr, S2S_ERR_HIP, "hipSetDevice failed");
    std::string arch;
    int n_cu = 0;
    {   // hipGetDeviceProperties is a millisecond-class call: asked once per device and process
        static std::mutex mu;
        static std::map<int, std::pair<std::string, int>> seen;
        std::lock_guard<std::mutex> lock(mu);
        auto it = seen.find(device);
        if (it == seen.end()) {
            hipDeviceProp_t prop;
            HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
            it = seen.emplace(device, std::make_pair(std::string(prop.gcnArchName), prop.multiProcessorCount)).first;
        }
        arch = it->second.first;
        n_cu = it->second.second;
    }
    if (arch.compare(0, 6, "gfx950") != 0)
        return fail(nullptr, S2S_ERR_ARG, "device is " + arch + ", this library is gfx950 only");

    s2s_handle* h = new s2s_handle();
    h->cfg = *cfg;
    h->device = device;
    h->n_wg = n_cu > 0 ? n_cu : 256;
    const int k = cfg->seq_kmer;
    Arena A;
    ModelDev& M = h->model;
    std::memset(&M, 0, sizeof M);
    M.k = k; M.enc_layers = cfg->encoder_layers; M.dec_layers = cfg->decoder_layers; M.pre_layers = cfg->pre_layers;
    M.scale = cfg->scaling_max_value;
    const float* p = static_cast<const float*>(blob);
    M.pe_enc = A.put(take(p, 16 * 64), 16 * 64);
    {   // src_emb.weight [64][5k] -> transposed [5k][64] so a one-hot column is one contiguous row
        const float* w = take(p, (size_t)64 * 5 * k);
        std::vector<float> wt((size_t)5 * k * 64);
        for (int f = 0; f < 64; ++f)
            for (int i = 0; i < 5 * k; ++i) wt[(size_t)i * 64 + f] = w[(size_t)f * 5 * k + i];
        M.emb_wt = A.put(wt.data(), wt.size());
        M.emb_b = A.put(take(p, 64), 64);
    }
    for (int i = 0; i < cfg->pre_layers; ++i) {
        const float* w = take(p, 4096);
        M.pre_w[i] = A.put_afrag(w, 64, 64, false);
        M.pre_wh[i] = pack_linear64_h(A, w);
        M.pre_b[i] = A.put(take(p, 64), 64);
    }
    for (int l = 0; l < cfg->encoder_layers; ++l) M.enc[l] = pack_layer(A, p);
    M.noise = pack_mlp(A, p);
    M.conc = pack_mlp(A, p);
    M.rate = pack_mlp(A, p);
    M.pe_dec = A.put(take(p, 250 * 64), 250 * 64);
    for (int l = 0; l < cfg->decoder_layers; ++l) M.dec[l] = pack_layer(A, p);
    M.out_w = A.put(take(p, 64), 64);
    M.out_b = A.put(take(p, 1), 1);
    while (A.v.size() % 4) A.v.push_back(0.0f);
    h->arena_floats = A.v.size();

    auto bail = [&](hipError_t e, const char* what) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(e);
        s2s_destroy(h);
        return S2S_ERR_HIP;
    };
    hipError_t e;
    // chunks per launch: the kernel needs no per-chunk workspace, so a launch is as long as 32-bit chunk indices allow
    // comfortably (every launch ends in a tail of up to one chunk time per workgroup)
    h->tile = 1 << 20;
    {   // one allocation (each hipMalloc is a driver round trip): weights | hand-off slots | export scratch for the streaming
        // path's usual super-batch, so that its first s2s_export_reads does not have to drain the stream in order to grow it |
        // s2s_svb_encode scratch (rows of a super-batch; POD5: ~6 per 10 kb read)
        const int cap = 5 * 32768 + 1, rows = 16384;   // run_streaming's largest super-batches (131,072 chunks + one read) fit without a re-allocation
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t o_hand = up(h->arena_floats * sizeof(float));
        const size_t o_counts = o_hand + up((size_t)h->n_wg * S2S_MAX_GROUP * S2S_SLOT_FLOATS * sizeof(float));
        const size_t o_offs = o_counts + up((size_t)cap * sizeof(int));
        const size_t o_svb = o_offs + up((size_t)cap * sizeof(long long));
        const size_t o_stats = o_svb + up((size_t)rows * sizeof(int));
        h->slab_bytes = o_stats + 256;
        if ((e = hipMalloc(&h->slab, h->slab_bytes)) != hipSuccess) return bail(e, "hipMalloc(handle)");
        h->d_arena = reinterpret_cast<float*>(h->slab);
        h->handoff = reinterpret_cast<float*>(h->slab + o_hand);
        h->ws_counts = reinterpret_cast<int*>(h->slab + o_counts);
        h->ws_offs = reinterpret_cast<long long*>(h->slab + o_offs);
        h->ws_svb = reinterpret_cast<int*>(h->slab + o_svb);
        h->d_stats = reinterpret_cast<unsigned long long*>(h->slab + o_stats);
        h->ws_export_cap = cap;
        h->ws_svb_cap = rows;
    }
    if ((e = hipMemcpy(h->d_arena, A.v.data(), h->arena_floats * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(arena)");
    if ((e = hipMemset(h->d_stats, 0, 256)) != hipSuccess) return bail(e, "hipMemset(stats)");
    const struct { const void* fn; int bytes; } dyn_lds[] = {
        {reinterpret_cast<const void*>(s2s_fused_kernel<0, false>), Fused<0>::LDS}, {reinterpret_cast<const void*>(s2s_fused_kernel<1, false>), Fused<1>::LDS},
        {reinterpret_cast<const void*>(s2s_fused_kernel<3, false>), Fused<3>::LDS}, {reinterpret_cast<const void*>(s2s_fused_kernel<0, true>), Fused<0>::LDS},
        {reinterpret_cast<const void*>(s2s_fused_kernel<1, true>), Fused<1>::LDS},  {reinterpret_cast<const void*>(s2s_fused_kernel<3, true>), Fused<3>::LDS},
        {reinterpret_cast<const void*>(s2s_fused_kernel<1, false, true>), Fused<1>::LDS}, {reinterpret_cast<const void*>(s2s_fused_kernel<3, false, true>), Fused<3>::LDS},
        {reinterpret_cast<const void*>(s2s_fused_kernel<1, true, true>), Fused<1>::LDS},  {reinterpret_cast<const void*>(s2s_fused_kernel<3, true, true>), Fused<3>::LDS}};
    for (const auto& k : dyn_lds)
        if ((e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, k.bytes)) != hipSuccess)
            return bail(e, "hipFuncSetAttribute(dynamic LDS)");
#if defined(S2S_DIAG) || defined(S2S_TILEHIST)
    if ((e = hipMalloc(&h->d_diag, 8 * 48 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc(diag)");
    if ((e = hipMemset(h->d_diag, 0, 8 * 48 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset(diag)");
#endif
    // S2S_ATTENTION_PATH = fast | exact (any case): that path, no calibration launch; unset, empty or "auto": calibrate;
    // anything else is refused (a typo must not pin the fast path silently)
    std::string want;
    if (const char* env = getenv("S2S_ATTENTION_PATH"))
        for (const char* c = env; *c; ++c) want += (char)std::tolower((unsigned char)*c);
    if (want == "fast" || want == "exact") {
        h->attn_exact = want == "exact" ? 1 : 0;
    } else if (!want.empty() && want != "auto") {
        g_create_error = "S2S_ATTENTION_PATH must be fast, exact or auto (got \"" + want + "\")";
        s2s_destroy(h);
        return S2S_ERR_ARG;
    } else if (calibrate_attention(h) != S2S_OK) {
        g_create_error = "attention-path calibration launch: " + h->err;
        s2s_destroy(h);
        return S2S_ERR_HIP;
    }
    *out = h;
    return S2S_OK;
}

int s2s_set_attention_path(s2s_handle* h, int32_t path) {
    if (!h) return S2S_ERR_ARG;
    if (path != 0 && path != 1) return fail(h, S2S_ERR_ARG, "attention path must be 0 (fast path first) or 1 (exact path)");
    h->attn_exact = path;
    return S2S_OK;
}

int s2s_get_attention_path(const s2s_handle* h, int32_t* path, double* calibration_redo_rate) {
    if (!h) return S2S_ERR_ARG;
    if (path) *path = h->attn_exact;
    if (calibration_redo_rate) *calibration_redo_rate = h->calib_redo_rate;
    return S2S_OK;
}

void s2s_destroy(s2s_handle* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    for (auto& ev : h->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    free_scratch(h, h->ws_counts);
    free_scratch(h, h->ws_offs);
    free_scratch(h, h->ws_svb);
    if (h->slab) (void)hipFree(h->slab);
    if (h->d_diag) (void)hipFree(h->d_diag);
    delete h;
}

const char* s2s_last_error(const s2s_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

static int predict_impl(s2s_handle* h, void* stream_, const uint8_t* bases, const int64_t* chunk_start, const uint8_t* n_valid, int64_t first_global_chunk,
                       int32_t B, const s2s_params* params, const float* inject_g, const float* inject_zdw,
                       const float* inject_z01, float* out_signal, int32_t* out_dur, const s2s_debug* dbg) {
    if (!h) return S2S_ERR_ARG;
    if (B < 0) return fail(h, S2S_ERR_ARG, "B < 0");
    if (B == 0) return S2S_OK;
    if (!bases || !n_valid || !params || !out_signal || !out_dur) return fail(h, S2S_ERR_ARG, "NULL argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(h, S2S_ERR_HIP, "hipSetDevice failed");
    if (!(params->min_duration >= 0.0f)) return fail(h, S2S_ERR_ARG, "min_duration must be >= 0");
    if (!params->duration_sampling && !(params->dwell_std > 0.0f) && !(params->dwell_mean >= 0.0f))
        return fail(h, S2S_ERR_ARG, "dwell_mean must be >= 0");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const ParamsDev P = to_dev(params);
    DebugDev D;
    std::memset(&D, 0, sizeof D);
    if (dbg) {
        D.emb_out = dbg->emb_out; D.enc_out = dbg->enc_out; D.sigma = dbg->sigma; D.conc = dbg->conc;
        D.rate = dbg->rate; D.g = dbg->g; D.y_scaled = dbg->y_scaled; D.z01 = dbg->z01;
        D.emb_in = dbg->emb_in; D.dec_in = dbg->dec_in;
    }
    D.diag = h->d_diag;
    D.stats = h->d_stats;
    h->stat_chunks += B;
    const int nb = S2S_T_ENC + h->cfg.seq_kmer - 1;
    for (int64_t s = 0; s < B; s += h->tile) {
        const int n = (int)((B - s < h->tile) ? (B - s) : h->tile);
        const uint8_t* tb = chunk_start ? bases : bases + (size_t)s * nb;
        const long long* tcs = reinterpret_cast<const long long*>(chunk_start ? chunk_start + s : nullptr);
        const float* tg = inject_g ? inject_g + s * 16 : nullptr;
        const float* tzdw = inject_zdw ? inject_zdw + s * 16 : nullptr;
        const float* tz01 = inject_z01 ? inject_z01 + (size_t)s * S2S_T_DEC : nullptr;
        float* tsig = out_signal + (size_t)s * S2S_T_DEC;
        const long long fc = (long long)(first_global_chunk + s);
        const dim3 grid(n < h->n_wg ? n : h->n_wg), block(DEC_WAVES * 64);
        const int mode = h->cfg.compute_mode;
        EventPair ev{};
        if (h->profiling) {
            HIP_TRY(h, hipEventCreate(&ev.a));
            HIP_TRY(h, hipEventCreate(&ev.b));
            HIP_TRY(h, hipEventRecord(ev.a, stream));
        }
        const bool test = dbg || inject_g || inject_zdw || inject_z01;
        const bool exact = h->attn_exact && mode != S2S_MODE_F32;
        auto fused = exact ? (test ? (mode == S2S_MODE_F16 ? s2s_fused_kernel<3, true, true> : s2s_fused_kernel<1, true, true>)
                                   : (mode == S2S_MODE_F16 ? s2s_fused_kernel<3, false, true> : s2s_fused_kernel<1, false, true>))
                   : test  ? (mode == S2S_MODE_F16 ? s2s_fused_kernel<3, true> : mode == S2S_MODE_F16X3 ? s2s_fused_kernel<1, true> : s2s_fused_kernel<0, true>)
                           : (mode == S2S_MODE_F16 ? s2s_fused_kernel<3, false> : mode == S2S_MODE_F16X3 ? s2s_fused_kernel<1, false> : s2s_fused_kernel<0, false>);
        if (exact) h->stat_exact_chunks += n;
        hipLaunchKernelGGL(fused, grid, block, mode == S2S_MODE_F16 ? Fused<3>::LDS : mode == S2S_MODE_F16X3 ? Fused<1>::LDS : Fused<0>::LDS, stream, h->model, h->d_arena, tb, tcs,
                           n_valid + s, n, fc, P, tg, tzdw, tz01, h->handoff, out_dur + s * 16, tsig, D, (long long)s);
        if (h->profiling) {
            HIP_TRY(h, hipEventRecord(ev.b, stream));
            ev.chunks = n;
            h->events.push_back(ev);
        }
    }
    HIP_TRY(h, hipGetLastError());
    return S2S_OK;
}

int s2s_predict_chunks(s2s_handle* h, void* stream, const uint8_t* bases, const uint8_t* n_valid, int64_t first_global_chunk,
                       int32_t B, const s2s_params* params, const float* inject_g, const float* inject_zdw,
                       const float* inject_z01, float* out_signal, int32_t* out_dur, const s2s_debug* dbg) {
    return predict_impl(h, stream, bases, nullptr, n_valid, first_global_chunk, B, params, inject_g, inject_zdw, inject_z01,
                        out_signal, out_dur, dbg);
}

int s2s_predict_packed(s2s_handle* h, void* stream, const uint8_t* read_bytes, const int64_t* chunk_start,
                       const uint8_t* n_valid, int64_t first_global_chunk, int32_t B, const s2s_params* params,
                       float* out_signal, int32_t* out_dur) {
    if (h && !chunk_start) return fail(h, S2S_ERR_ARG, "chunk_start is NULL");
    return predict_impl(h, stream, read_bytes, chunk_start, n_valid, first_global_chunk, B, params, nullptr, nullptr, nullptr,
                        out_signal, out_dur, nullptr);
}

int s2s_export_reads(s2s_handle* h, void* stream_, const float* signal, int32_t B, const int32_t* read_first, int32_t R,
                     int64_t* out_offsets, float* out_pa, int16_t* out_dac, int64_t capacity, float digitisation,
                     float range, float offset_mean, int32_t rna) {
    if (!h) return S2S_ERR_ARG;
    if (B < 0 || R < 0) return fail(h, S2S_ERR_ARG, "negative size");
    if (!signal || !read_first || !out_offsets) return fail(h, S2S_ERR_ARG, "NULL argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(h, S2S_ERR_HIP, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (B + 1 > h->ws_export_cap) {          // grows outside of the steady state only
        HIP_TRY(h, hipStreamSynchronize(stream));
        free_scratch(h, h->ws_counts);
        free_scratch(h, h->ws_offs);
        h->ws_counts = nullptr; h->ws_offs = nullptr; h->ws_export_cap = 0;
        const int cap = B + 1 + B / 4;
        HIP_TRY(h, hipMalloc(&h->ws_counts, (size_t)cap * sizeof(int)));
        HIP_TRY(h, hipMalloc(&h->ws_offs, (size_t)cap * sizeof(long long)));
        h->ws_export_cap = cap;
    }
    if (B > 0) hipLaunchKernelGGL(s2s_count_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, signal, B, h->ws_counts);
    hipLaunchKernelGGL(s2s_scan_kernel, dim3(1), dim3(1024), 0, stream, h->ws_counts, B, h->ws_offs);
    hipLaunchKernelGGL(s2s_read_offsets_kernel, dim3((R + 256) / 256), dim3(256), 0, stream, h->ws_offs, read_first, R,
                       reinterpret_cast<long long*>(out_offsets));
    if (B > 0 && (out_pa || out_dac))
        hipLaunchKernelGGL(s2s_compact_kernel, dim3((B + 3) / 4), dim3(256), 0, stream, signal, B, h->ws_offs, read_first, R,
                           out_pa, reinterpret_cast<short*>(out_dac), (long long)capacity, digitisation, range, offset_mean,
                           rna);
    HIP_TRY(h, hipGetLastError());
    return S2S_OK;
}

int s2s_svb_encode(s2s_handle* h, void* stream_, const int16_t* samples, const int64_t* read_offsets, const int32_t* row_read,
                   const int32_t* row_index, int32_t N, int64_t row_samples, int32_t variant, uint8_t* out, int64_t capacity,
                   int64_t* out_offsets) {
    if (!h) return S2S_ERR_ARG;
    if (N < 0 || row_samples <= 0 || (variant != 16 && variant != 32)) return fail(h, S2S_ERR_ARG, "bad argument");
    if (!out_offsets || (N > 0 && (!samples || !read_offsets || !row_read || !row_index || !out)))
        return fail(h, S2S_ERR_ARG, "NULL argument");
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(h, S2S_ERR_HIP, "hipSetDevice failed");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (N + 1 > h->ws_svb_cap) {             // grows outside of the steady state only
        HIP_TRY(h, hipStreamSynchronize(stream));
        free_scratch(h, h->ws_svb);
        h->ws_svb = nullptr; h->ws_svb_cap = 0;
        const int cap = N + 1 + N / 4;
        HIP_TRY(h, hipMalloc(&h->ws_svb, (size_t)cap * sizeof(int)));
        h->ws_svb_cap = cap;
    }
    const short* sp = reinterpret_cast<const short*>(samples);
    const long long* ro = reinterpret_cast<const long long*>(read_offsets);
    long long* oo = reinterpret_cast<long long*>(out_offsets);
    if (N > 0) {
        if (variant == 32)
            hipLaunchKernelGGL((s2s_svb_kernel<32, false>), dim3(N), dim3(256), 0, stream, sp, ro, row_read, row_index,
                               (long long)row_samples, h->ws_svb, nullptr, nullptr, 0LL);
        else
            hipLaunchKernelGGL((s2s_svb_kernel<16, false>), dim3(N), dim3(256), 0, stream, sp, ro, row_read, row_index,
                               (long long)row_samples, h->ws_svb, nullptr, nullptr, 0LL);
    }
    hipLaunchKernelGGL(s2s_scan_kernel, dim3(1), dim3(1024), 0, stream, h->ws_svb, N, oo);
    if (N > 0) {
        if (variant == 32)
            hipLaunchKernelGGL((s2s_svb_kernel<32, true>), dim3(N), dim3(256), 0, stream, sp, ro, row_read, row_index,
                               (long long)row_samples, nullptr, oo, out, (long long)capacity);
        else
            hipLaunchKernelGGL((s2s_svb_kernel<16, true>), dim3(N), dim3(256), 0, stream, sp, ro, row_read, row_index,
                               (long long)row_samples, nullptr, oo, out, (long long)capacity);
    }
    hipLaunchKernelGGL(s2s_svb_check_kernel, dim3(1), dim3(64), 0, stream, oo, N, (long long)capacity);
    HIP_TRY(h, hipGetLastError());
    return S2S_OK;
}

int s2s_philox_u32(s2s_handle* h, void* stream_, uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                   int32_t n, uint32_t* out) {
    if (!h) return S2S_ERR_ARG;
    if (n < 0 || !out) return fail(h, S2S_ERR_ARG, "bad argument");
    if (n == 0) return S2S_OK;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(h, S2S_ERR_HIP, "hipSetDevice failed");
    hipLaunchKernelGGL(s2s_philox_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_),
                       (unsigned)seed, (unsigned)(seed >> 32), c0, c1, c2, c3, n, out);
    HIP_TRY(h, hipGetLastError());
    return S2S_OK;
}

int s2s_set_profiling(s2s_handle* h, int32_t enabled) {
    if (!h) return S2S_ERR_ARG;
    h->profiling = enabled != 0;
    return S2S_OK;
}

int s2s_get_kernel_ms(s2s_handle* h, double* ms_total, int64_t* launches, int64_t* chunks) {
    if (!h) return S2S_ERR_ARG;
    double tot = 0.0;
    int64_t nl = 0, nc = 0;
    for (auto& ev : h->events) {
        HIP_TRY(h, hipEventSynchronize(ev.b));
        float ms = 0.0f;
        HIP_TRY(h, hipEventElapsedTime(&ms, ev.a, ev.b));
        tot += ms; nl += 1; nc += ev.chunks;
        (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b);
    }
    h->events.clear();
    if (ms_total) *ms_total = tot;
    if (launches) *launches = nl;
    if (chunks) *chunks = nc;
    return S2S_OK;
}

int s2s_stats_read(s2s_handle* h, uint64_t* out10) {
    if (!h || !out10) return S2S_ERR_ARG;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(h, S2S_ERR_HIP, "hipSetDevice failed");
    HIP_TRY(h, hipDeviceSynchronize());
    unsigned long long raw[8];
    HIP_TRY(h, hipMemcpy(raw, h->d_stats, sizeof raw, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemset(h->d_stats, 0, sizeof raw));
    out10[0] = (uint64_t)h->stat_chunks;
    out10[1] = (uint64_t)h->stat_chunks * DEC_WAVES * S2S_HEADS * (uint64_t)h->cfg.decoder_layers;
    out10[2] = raw[S2S_STAT_REDO];
    out10[3] = raw[S2S_STAT_CYCLES]; out10[4] = raw[S2S_STAT_TICKS]; out10[5] = raw[S2S_STAT_WGS];
    out10[6] = (uint64_t)h->stat_exact_chunks; out10[7] = 0; out10[8] = 0; out10[9] = 0;
    h->stat_chunks = 0; h->stat_exact_chunks = 0;
    return S2S_OK;
}

int s2s_diag_read(s2s_handle* h, uint64_t* out48) {
    if (!h || !out48) return S2S_ERR_ARG;
    if (!h->d_diag) return fail(h, S2S_ERR_ARG, "not a diagnostic (-DS2S_DIAG) build");
    HIP_TRY(h, hipDeviceSynchronize());
    HIP_TRY(h, hipMemcpy(out48, h->d_diag, 8 * 48 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemset(h->d_diag, 0, 8 * 48 * sizeof(uint64_t)));
    return S2S_OK;
}

}  // extern "C"
