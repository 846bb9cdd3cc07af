// fa_bf16_step.h -- per-step helpers shared by the software-pipelined bf16 kernels (fragment reads, masks, exp + pack,
// lane-local maxima).  gfx950 only.
#pragma once
#include "fa_bf16_common.h"

namespace fa {

// K fragment of sub-tile (stage-local 32-key block kb), k-step ks
template <int D>
__device__ __forceinline__ bf16x8 load_k_frag(const char* k_lds, int k_row_off, int k_g, int kb, int ks)
{
    return *(const bf16x8*)(k_lds + k_row_off + kb * 32 * (2 * D) + (((2 * ks) ^ k_g) * 16));
}
// V^T fragment v = tt * DB + db of stage-local 32-key block kb
template <int D>
__device__ __forceinline__ bf16x8 load_v_frag(const char* v_lds, int v_lane_off, int kb, int v)
{
    constexpr int DB = D / 32;
    const int tt = v / DB, db = v % DB;
    const int off0 = ((kb * 8 + 4 * tt + 0) * (D / 16) + 2 * db) * 128;
    const int off1 = ((kb * 8 + 4 * tt + 2) * (D / 16) + 2 * db) * 128;
    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_lds + v_lane_off + off0));
    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_lds + v_lane_off + off1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
}

// The same fragment through inline asm.  hipcc orders every ds_read_b64_tr_b16 *builtin* behind all LDS-DMA in flight
// (s_waitcnt vmcnt(0) in front of the first one after a global_load_lds), which would expose the whole latency of the
// next stage's DMA once per stage; an asm read is invisible to that pass.  The caller owns the wait: wait_v_frags()
// before the first MFMA that consumes them.  KB / V must be compile-time (immediate offsets).
template <int D, int KB, int V>
__device__ __forceinline__ void load_v_frag_asm(unsigned v_addr, s16x4& lo, s16x4& hi)
{
    constexpr int DB = D / 32;
    constexpr int tt = V / DB, db = V % DB;
    constexpr int off0 = ((KB * 8 + 4 * tt + 0) * (D / 16) + 2 * db) * 128;
    constexpr int off1 = ((KB * 8 + 4 * tt + 2) * (D / 16) + 2 * db) * 128;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(v_addr), "i"(off0));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(v_addr), "i"(off1));
}

__device__ __forceinline__ float rowmax16(const f32x16& s)
{
    float p0 = max3_safe(s[0], s[1], s[2]), p1 = max3_safe(s[3], s[4], s[5]);
    float p2 = max3_safe(s[6], s[7], s[8]), p3 = max3_safe(s[9], s[10], s[11]);
    p0 = max3_safe(p0, s[12], s[13]);
    p1 = max3_safe(p1, s[14], s[15]);
    return xhalf_max(fmaxf(max3_safe(p0, p1, p2), p3));
}
// row minimum over a lane pair's 32 keys (prologue only: decides whether the optimistic mix samples more keys for its reference)
__device__ __forceinline__ float rowmin16(const f32x16& s)
{
    float m = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) m = fminf(m, s[r]);
    return -xhalf_max(-m);
}

__device__ __forceinline__ void mask16(f32x16& s, int key0, int qi, int n, int hi, bool causal)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int key = key0 + 4 * hi + (r & 3) + 8 * (r >> 2);
        if ((key >= n) || (causal && key > qi)) s[r] = -INFINITY;
    }
}

// exp + pack of one block's 16 scores, element range [e0, e1) and pack of fragment(s) whose elements are complete
// (clamp = false: the optimistic mix, where an overflow has to reach the row sum instead of being clamped away)
// PF = 2: half h (0 / 1) of the lo term of P fragment f -- elements 8 f + 4 h .. + 3, two packed registers.  lo = fp16(p - float(hi)):
// one v_fma_mixlo_f16 / v_fma_mixhi_f16 per element computes hi * (-1.0) + p in fp32 (hi: an fp16 half of the packed register, picked
// by op_sel; the difference is exact) and rounds it into the low / high half of the destination.  s holds the fp32 p the pack read.
__device__ __forceinline__ void lo_half(const f32x16& s, int f, int h, const bf16x8& hi, bf16x8& lo)
{
    const u32x4 hv = __builtin_bit_cast(u32x4, hi);
    u32x4 lv = __builtin_bit_cast(u32x4, lo);
#pragma unroll
    for (int r = 2 * h; r < 2 * h + 2; ++r) {
        unsigned d;
        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(d)
                     : "v"(hv[r]), "v"(s[8 * f + 2 * r]), "v"(s[8 * f + 2 * r + 1]));
        lv[r] = d;
    }
    lo = __builtin_bit_cast(bf16x8, lv);
}

// PF = 3: the lo term of P in bf16, lo = bf16(p - float(hi)) with hi = bf16(p) (round to nearest even, v_cvt_pk_bf16_f32).  gfx950 has
// no v_fma_mix with bf16 sources; the exact difference comes from the dot-product unit instead -- one VOP2 instruction per element:
//     v_dot2c_f32_bf16 p, K, hi_pk      p += K.lo * hi_pk.lo + K.hi * hi_pk.hi      with K = (-1, 0) for the low half, (0, -1) for the high
// (products of bf16 values and their sum with p are exact here: profiles/ubench/ubench_dot2.hip checks 4 M values bit for bit, including
// the ends of the optimistic mix's exponent window) -- IN PLACE: the fp32 p is dead once both terms exist, so the residuals need no
// registers.  One v_cvt_pk_bf16_f32 per register then packs them.  Hazard (ours inside asm; LLVM's DotWriteDifferentVALURead): a VALU
// instruction other than the same dot opcode may read a dot result three wait states after it at the earliest -- so the dots and the
// packs of a half fragment are SEPARATE units of the step schedule with at least one other unit between them (xn_schedule_ok), and the
// phase-structured form below runs a whole fragment's eight dots in front of its four packs.
constexpr unsigned kDotNegLo = 0x0000bf80u, kDotNegHi = 0xbf800000u;   // bf16 pairs (-1, 0) and (0, -1)
// dots of half h (0 / 1) of fragment f: elements 8 f + 4 h .. + 3 of s become p - hi
__device__ __forceinline__ void lo_dots_bf16(f32x16& s, int f, int h, const bf16x8& hi)
{
    const u32x4 hv = __builtin_bit_cast(u32x4, hi);
    const int e = 8 * f + 4 * h;
    asm volatile("v_dot2c_f32_bf16 %0, %4, %6\n\tv_dot2c_f32_bf16 %1, %5, %6\n\tv_dot2c_f32_bf16 %2, %4, %7\n\tv_dot2c_f32_bf16 %3, %5, %7"
                 : "+v"(s[e]), "+v"(s[e + 1]), "+v"(s[e + 2]), "+v"(s[e + 3])
                 : "s"(kDotNegLo), "s"(kDotNegHi), "v"(hv[2 * h]), "v"(hv[2 * h + 1]));
}
// packs of the same half: two registers of the lo fragment
__device__ __forceinline__ void lo_packs_bf16(const f32x16& s, int f, int h, bf16x8& lo)
{
    u32x4 lv = __builtin_bit_cast(u32x4, lo);
    const int e = 8 * f + 4 * h;
    unsigned d0, d1;
    asm volatile("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5" : "=&v"(d0), "=&v"(d1) : "v"(s[e]), "v"(s[e + 1]), "v"(s[e + 2]), "v"(s[e + 3]));
    lv[2 * h] = d0;
    lv[2 * h + 1] = d1;
    lo = __builtin_bit_cast(bf16x8, lv);
}
// a whole fragment, phase-structured (prologue / tail stages): eight dots, then four packs -- every pack at least three instructions
// behind the last dot it reads
__device__ __forceinline__ void lo_frag_bf16(f32x16& s, int f, const bf16x8& hi, bf16x8& lo)
{
    const u32x4 hv = __builtin_bit_cast(u32x4, hi);
    const int e = 8 * f;
    u32x4 lv;
    asm volatile("v_dot2c_f32_bf16 %4, %12, %14\n\tv_dot2c_f32_bf16 %5, %13, %14\n\tv_dot2c_f32_bf16 %6, %12, %15\n\tv_dot2c_f32_bf16 %7, %13, %15\n\t"
                 "v_dot2c_f32_bf16 %8, %12, %16\n\tv_dot2c_f32_bf16 %9, %13, %16\n\tv_dot2c_f32_bf16 %10, %12, %17\n\tv_dot2c_f32_bf16 %11, %13, %17\n\t"
                 "v_cvt_pk_bf16_f32 %0, %4, %5\n\tv_cvt_pk_bf16_f32 %1, %6, %7\n\tv_cvt_pk_bf16_f32 %2, %8, %9\n\tv_cvt_pk_bf16_f32 %3, %10, %11"
                 : "=&v"(lv[0]), "=&v"(lv[1]), "=&v"(lv[2]), "=&v"(lv[3]), "+v"(s[e]), "+v"(s[e + 1]), "+v"(s[e + 2]), "+v"(s[e + 3]), "+v"(s[e + 4]),
                   "+v"(s[e + 5]), "+v"(s[e + 6]), "+v"(s[e + 7])
                 : "s"(kDotNegLo), "s"(kDotNegHi), "v"(hv[0]), "v"(hv[1]), "v"(hv[2]), "v"(hv[3]));
    lo = __builtin_bit_cast(bf16x8, lv);
}

template <int PF = 0>
__device__ __forceinline__ void exp_range(f32x16& s, bf16x8 (&pf)[2], float c, float off, int e0, int e1, bool clamp = true)
{
#pragma unroll
    for (int e = 0; e < 16; ++e)
        if (e >= e0 && e < e1) s[e] = clamp ? exp2_clamp01(fmaf(s[e], c, -off)) : fast_exp2(fmaf(s[e], c, -off));
#pragma unroll
    for (int f = 0; f < 2; ++f)
        if (e0 < 8 * (f + 1) && e1 >= 8 * (f + 1)) {  // this range completes fragment f
            pf[f] = pack_p16x8<PF>(s, 8 * f);
            asm volatile("" : "+v"(pf[f]));
        }
}

// cycle stamps for the in-kernel phase profile (PROF builds only)
__device__ __forceinline__ unsigned long long stamp() { return __builtin_readcyclecounter(); }

// lane-local (no cross-half exchange) maximum of 16 scores: three micro-steps u = 0, 1, 2.  One asm statement per
// micro-step (hipcc pads a conservative s_nop between two asm statements that touch the same register).  The operands are
// accumulators of MFMAs that retired a phase ago (see max3_raw).
__device__ __forceinline__ void lanemax_step(int u, const f32x16& sx, float (&pm)[4], float& out)
{
    if (u == 0) {
        asm volatile("v_max3_f32 %0, %3, %4, %5\n\tv_max3_f32 %1, %6, %7, %8\n\tv_max3_f32 %2, %9, %10, %11"
                     : "=&v"(pm[0]), "=&v"(pm[1]), "=&v"(pm[2])
                     : "v"(sx[0]), "v"(sx[1]), "v"(sx[2]), "v"(sx[3]), "v"(sx[4]), "v"(sx[5]), "v"(sx[6]), "v"(sx[7]), "v"(sx[8]));
    } else if (u == 1) {
        asm volatile("v_max3_f32 %0, %3, %4, %5\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %8, %9"
                     : "=&v"(pm[3]), "+v"(pm[0]), "+v"(pm[1])
                     : "v"(sx[9]), "v"(sx[10]), "v"(sx[11]), "v"(sx[12]), "v"(sx[13]), "v"(sx[14]), "v"(sx[15]));
    } else {
        asm volatile("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %4" : "=&v"(out) : "v"(pm[0]), "v"(pm[1]), "v"(pm[2]), "v"(pm[3]));
    }
}


}  // namespace fa
