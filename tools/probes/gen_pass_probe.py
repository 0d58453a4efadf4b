#!/usr/bin/env python
"""Generates tools/probes/pass_probe.hip: what is one 32-key tile of the decoder's fast softmax worth when its instructions are
issued (C) the way hipcc schedules the two-tile pass today -- LDS reads, wait, four score MFMAs back to back, s_nop, then
exponentials / splits with the P.V MFMAs in between -- and (P) as a software pipeline -- next tile's score MFMAs and the LDS
reads two tiles ahead spread between the current tile's vector instructions, never two MFMAs back to back?  Same instruction
counts in both: per tile 2 score + 4 P.V v_mfma_f32_32x32x16_f16, 16 v_exp_f32, 8 v_cvt_pk_f16_f32, 16 v_fma_mix, 4 ds_read_b128.
Explicit registers, written as one asm block per loop body so that nothing is rescheduled.  Run with 1 and 2 waves per SIMD."""
import sys

O, SC, SN = "v[0:15]", 16, 32           # accumulators: O, scores of the current tile (v16..31), of the next (v32..47)
QB1, QB2, KA, KB, VA0, VA1 = "v[48:51]", "v[52:55]", "v[56:59]", "v[60:63]", "v[64:67]", "v[68:71]"
KA2, KB2, VB0, VB1 = "v[100:103]", "v[104:107]", "v[108:111]", "v[112:115]"
PH, PL = ("v[72:75]", "v[76:79]"), ("v[80:83]", "v[84:87]")
ADDR = "v88"


def valu_half(sc, st, out, nops=True):
    """exp + split of 8 scores (registers sc+8st .. +7) -> P hi in v72+4st.., lo in v80+4st..: 8 exp, 4 cvt, 8 mix.  nops: the order
    hipcc emits today (every cvt directly in front of the two mix instructions that read it, one s_nop 0 between them)."""
    b = sc + 8 * st
    hi, lo = 72 + 4 * st, 80 + 4 * st
    for pair in range(4):
        r0, r1 = b + 2 * pair, b + 2 * pair + 1
        out.append(f"v_exp_f32_e32 v{r0}, v{r0}")
        out.append(f"v_exp_f32_e32 v{r1}, v{r1}")
    if not nops:
        for pair in range(4):
            out.append(f"v_cvt_pk_f16_f32 v{hi + pair}, v{b + 2 * pair}, v{b + 2 * pair + 1}")
    for pair in range(4):
        r0, r1 = b + 2 * pair, b + 2 * pair + 1
        if nops:
            out.append(f"v_cvt_pk_f16_f32 v{hi + pair}, v{r0}, v{r1}")
            out.append("s_nop 0")
        out.append(f"v_fma_mixlo_f16 v{lo + pair}, v{r0}, s20, -v{hi + pair} op_sel_hi:[0,0,1]")
        out.append(f"v_fma_mixhi_f16 v{lo + pair}, v{r1}, s20, -v{hi + pair} op_sel:[0,0,1] op_sel_hi:[0,0,1]")


def mf(dst, a, b, c):
    return f"v_mfma_f32_32x32x16_f16 {dst}, {a}, {b}, {c}"


def body_clustered():
    """two tiles per pass, as the compiler schedules it today"""
    o = []
    for i, (dst, off) in enumerate(((KA, 0), (KB, 64), (VA0, 128), (VA1, 192), (KA2, 256), (KB2, 320), (VB0, 384), (VB1, 448))):
        o.append(f"ds_read_b128 {dst}, {ADDR} offset:{off}")
    o.append("s_waitcnt lgkmcnt(0)")
    o.append(mf("v[16:31]", KA, QB1, "0"))
    o.append(mf("v[32:47]", KA2, QB1, "0"))
    o.append(mf("v[16:31]", KB, QB2, "v[16:31]"))
    o.append(mf("v[32:47]", KB2, QB2, "v[32:47]"))
    o.append("s_nop 8")
    for sc, (va0, va1) in ((SC, (VA0, VA1)), (SN, (VB0, VB1))):
        for st, va in ((0, va0), (1, va1)):
            v = []
            valu_half(sc, st, v)
            o += v
            o.append("s_nop 1")
            o.append(mf(O, va, PH[st], O))
            o.append(mf(O, va, PL[st], O))
    return o


def body_pipelined():
    """two tiles per loop body (same work as body_clustered), each as one pipeline stage"""
    o = []
    for tile, (sc_c, sc_n, ka, kb, ka_ld, kb_ld, va, va_ld) in enumerate((
            (SC, "v[32:47]", KA, KB, KA2, KB2, (VA0, VA1), (VB0, VB1)),
            (SN, "v[16:31]", KA2, KB2, KA, KB, (VB0, VB1), (VA0, VA1)))):
        # LDS reads for two tiles ahead (K) / one ahead (V): land during this stage, first used in the next one
        o.append("s_waitcnt lgkmcnt(0)")                 # (the reads issued a whole stage ago)
        first, second = [], []
        valu_half(sc_c, 0, first)
        valu_half(sc_c, 1, second)
        o.append(mf(sc_n, ka, QB1, "0"))                 # score MFMA 1 of the next tile
        o += first[:6]
        o.append(f"ds_read_b128 {ka_ld}, {ADDR} offset:{256 * tile}")
        o.append(f"ds_read_b128 {kb_ld}, {ADDR} offset:{256 * tile + 64}")
        o.append(mf(sc_n, kb, QB2, sc_n))                # score MFMA 2 (needs the first: >= 32 cycles later)
        o += first[6:]
        o.append(f"ds_read_b128 {va_ld[0]}, {ADDR} offset:{256 * tile + 128}")
        o.append(f"ds_read_b128 {va_ld[1]}, {ADDR} offset:{256 * tile + 192}")
        o.append("s_nop 1")
        o.append(mf(O, va[0], PH[0], O))
        o += second[:7]
        o.append(mf(O, va[0], PL[0], O))
        o += second[7:]
        o.append("s_nop 1")
        o.append(mf(O, va[1], PH[1], O))
        o.append("s_nop 7")                              # (in the real loop: the next stage's first vector instructions)
        o.append(mf(O, va[1], PL[1], O))
    return o


def body_pipelined2(pos=(3, 9, 15, 22, 28), tail_nop=1):
    """like body_pipelined, vector instructions in 8 exp / 4 cvt / 8 mix order (no s_nop 0), and the stage rotated: the previous
    tile's last P.V MFMA sits behind this stage's first exponentials; pos = index of the vector instruction (0..39) in front of
    which [pv1lo(prev), score 1, score 2, pv0hi, pv0lo] are issued; pv1hi closes the stage."""
    o = []
    for tile, (sc_c, sc_n, ka, kb, ka_ld, kb_ld, va, va_ld) in enumerate((
            (SC, "v[32:47]", KA, KB, KA2, KB2, (VA0, VA1), (VB0, VB1)),
            (SN, "v[16:31]", KA2, KB2, KA, KB, (VB0, VB1), (VA0, VA1)))):
        v = []
        valu_half(sc_c, 0, v, nops=False)
        valu_half(sc_c, 1, v, nops=False)
        assert len(v) == 40
        prev_va1 = (VB1, VA1)[tile]                      # the previous stage's second V operand
        m = {pos[0]: [mf(O, prev_va1, PL[1], O)], pos[1]: [mf(sc_n, ka, QB1, "0")], pos[2]: [mf(sc_n, kb, QB2, sc_n)],
             pos[3]: [mf(O, va[0], PH[0], O)], pos[4]: [mf(O, va[0], PL[0], O)]}
        lds = {1: f"ds_read_b128 {ka_ld}, {ADDR} offset:{256 * tile}", 5: f"ds_read_b128 {kb_ld}, {ADDR} offset:{256 * tile + 64}",
               24: f"ds_read_b128 {va_ld[0]}, {ADDR} offset:{256 * tile + 128}", 30: f"ds_read_b128 {va_ld[1]}, {ADDR} offset:{256 * tile + 192}"}
        o.append("s_waitcnt lgkmcnt(0)")
        for i, ins in enumerate(v):
            o += m.get(i, [])
            if i in lds:
                o.append(lds[i])
            o.append(ins)
        o.append(f"s_nop {tail_nop}")
        o.append(mf(O, va[1], PH[1], O))
    return o


def kernel(name, body):
    asm = "\\n\\t".join(body)
    return f'''
__global__ __launch_bounds__(512) void {name}(float seed, float* sink, long long* out, int iters) {{
    extern __shared__ float lds[];                 // (120 KB requested at launch: one workgroup per CU)
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed + i;
    __syncthreads();
    const unsigned addr = (threadIdx.x & 63) * 16;
    float r;
    const long long t0 = __builtin_readcyclecounter();
    asm volatile(
        "v_mov_b32 {ADDR}, %1\\n\\t"
        "s_mov_b32 s20, 1.0\\n\\t"
        "s_mov_b32 s21, %2\\n\\t"
        "1:\\n\\t"
        "{asm}\\n\\t"
        "s_sub_u32 s21, s21, 1\\n\\t"
        "s_cmp_lg_u32 s21, 0\\n\\t"
        "s_cbranch_scc1 1b\\n\\t"
        "s_nop 15\\n\\t"
        "v_mov_b32 %0, v0"
        : "=v"(r) : "v"(addr), "s"(iters)
        : "s20", "s21", "scc", "memory", {", ".join(f'"v{i}"' for i in range(0, 120))});
    const long long t1 = __builtin_readcyclecounter();
    if (r == 123.456f) sink[0] = r;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}}
'''


def main():
    src = ('''// GENERATED by tools/probes/gen_pass_probe.py -- see its docstring.
// hipcc --offload-arch=gfx950 -O3 -w -o pass_probe pass_probe.hip && ./pass_probe
#include <hip/hip_runtime.h>
#include <cstdio>
''' + kernel("probe_clustered", body_clustered()) + kernel("probe_pipelined", body_pipelined())
        + kernel("probe_pipe2a", body_pipelined2()) + kernel("probe_pipe2b", body_pipelined2((2, 8, 14, 21, 30)))
        + kernel("probe_pipe2c", body_pipelined2((4, 10, 16, 24, 32))) + kernel("probe_pipe2d", body_pipelined2((0, 6, 12, 20, 26))) + '''
template <class K> static void run(const char* what, K k, int threads) {
    float* sink; long long* out;
    hipMalloc(&sink, 4); hipMalloc(&out, 256 * 8 * 8);
    const int iters = 20000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 120 << 10);
    k<<<256, threads, 120 << 10>>>(1.0f, sink, out, 100);
    hipDeviceSynchronize();
    k<<<256, threads, 120 << 10>>>(1.0f, sink, out, iters);
    hipDeviceSynchronize();
    long long h[8];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("%-46s", what);
    for (int w = 0; w < threads / 64; ++w) printf(" %8.1f", (double)h[w] / iters);
    printf("   cycles per pass of 2 tiles, per wave\\n");
    hipFree(sink); hipFree(out);
}
int main() {
    run("1 wave/SIMD  clustered (today's schedule)", probe_clustered, 256);
    run("1 wave/SIMD  pipelined", probe_pipelined, 256);
    run("2 waves/SIMD clustered (today's schedule)", probe_clustered, 512);
    run("2 waves/SIMD pipelined", probe_pipelined, 512);
    run("1 wave/SIMD  pipelined, rotated (3,9,15,22,28)", probe_pipe2a, 256);
    run("2 waves/SIMD pipelined, rotated (3,9,15,22,28)", probe_pipe2a, 512);
    run("2 waves/SIMD pipelined, rotated (2,8,14,21,30)", probe_pipe2b, 512);
    run("2 waves/SIMD pipelined, rotated (4,10,16,24,32)", probe_pipe2c, 512);
    run("2 waves/SIMD pipelined, rotated (0,6,12,20,26)", probe_pipe2d, 512);
    return 0;
}
''')
    open(sys.argv[1] if len(sys.argv) > 1 else "tools/probes/pass_probe.hip", "w").write(src)


if __name__ == "__main__":
    main()
