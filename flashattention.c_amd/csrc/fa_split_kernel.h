// fa_split_kernel.h -- fused flash-attention forward on the 16-bit matrix pipe with split (hi + lo) operands (gfx950): the kernel
// template instantiated per (dtype, head dim) in fa_split_{f32,bf16}_d{32,64,128}.hip; dispatch in fa_fwd_f32_split.hip (fp32 tensors) and
// fa_fwd_bf16_split.hip (bf16 tensors, accurate mode).
//
// Same contract as fa_fwd_f32.hip (fp32 Q/K/V in, fp32 O out; replaces flash_tiled_coarse{,_causal},
// /root/reference/src/flashattention.cu:139-579), but both contractions run as THREE 16-bit matrix instructions on two-term splits of
// their fp32 operands, a.b ~= a_hi.b_hi + a_lo.b_hi + a_hi.b_lo, with fp32 accumulation in the matrix core:
//
//     S = Q'.K^T   FP16 terms (round 5):  x_hi = f16(x),  x_lo = f16(x - x_hi)       22 significant bits; v_mfma_f32_32x32x16_f16
//     O += P.V     BF16 terms:            x_hi = bf16(x), x_lo = bf16(x - x_hi)      16 significant bits, fp32's exponent range (P = 2^(s - m0 - B)
//                                                                                    lives near 2^-96); v_mfma_f32_32x32x16_bf16
//
// The 16-bit pipes are 16x the fp32 one on this chip (2.5 PF vs 157 TF dense), so three products still beat v_mfma_f32_32x32x2_f32 by 5x
// on the matrix pipe.  ERROR of a logit from the operand terms: <= 3 * 2^-22 * sum |q'_i k_i| -- below the rounding bound of the
// reference's own fp32 FMA chain, d * 2^-24 * sum |q'_i k_i|, for every d >= 12, whatever the data (rounds 1-4 carried K and Q' as two BF16
// terms: 2^-16.4 * sum |q'_i k_i| in the worst case, 6e-2 in O on coherent inputs that the RMS-model guard of those rounds let through:
// VERDICT r04).  What is left on wide logits is fp32 ACCUMULATION: every matrix instruction rounds (truncates) its partial sum at the
// magnitude it has then, so the hi.hi products of ALL k-steps come first -- the accumulator starts at -(m0 + B), as large as the row's widest
// logit, and walks down to the score's own small magnitude -- and the cross terms are added last, where an ulp is small (scores(), qk(),
// the slot schedule).  And the keys are CENTRED before they are split (k_j - kbar, kbar the coordinate-wise median of three keys of the
// share; the row constant q'.kbar goes back into the log-sum-exp): a magnitude all keys share never reaches a rounded sum -- see KEY
// CENTERING in the kernel.  O's terms add <= 3 * 2^-17 * max|v|.  Measured: <= 1e-4 on unit-variance data at scale 1, <= 1.5e-5 at
// 1/sqrt(d), <= 2.7e-4 on the coherent-rounding family of tests/adversarial.py (the reference's own FMA chain: up to 5.9e-3).
// RANGE GUARD (fp32 tensors under FA_KERNEL_AUTO, FwdParams::flag_mode = 4): fp16 terms hold |x| < 65520 (beyond: hi = inf, lo = -inf, NaN
// scores) and give elements below 2^-3 a subnormal lo term (absolute error <= 2^-25, times the partner element).  Every workgroup sees all
// keys of its slab and its own query rows; it tracks max |k| element-wise while converting K (one v_max3_f32 per four values) and the
// 2-norms of its Q' rows, and when the first attempt produced a NaN or  D * max|k| + sqrt(D) * max|q'|_2  exceeds kSubnormalBudget (8192:
// the subnormal terms stay below 2^-12 per logit in the worst case) the workgroup redoes its own rows in exact fp32 arithmetic before it exits
// (f32_exact_rows, fa_f32_exact.h: the body of the exact kernel, in the LDS this kernel is done with) and sets the caller's report word.
// Callers who know better select FA_KERNEL_SPLIT (no guard) or FA_KERNEL_MFMA.  fp32 range is kept for V, P (bf16 exponent) and, through
// the redo of rows whose accumulators come out tiny, for O.
//
//   workgroup   NWAVES waves x QB blocks of 32 query rows; K/V tiles of 32 keys.
//   HBM -> LDS  fp32 K/V rows are loaded into registers (two 16-byte loads per 8 values), split there, and written as FOUR
//               16-bit images per tile (K_hi, K_lo in fp16; V_hi, V_lo in bf16) in the layouts of fa_bf16_common.h: K row-major with
//               XOR-swizzled 16-byte slots (ds_read_b128 A fragments), V as [key/4][col/16][4][16] sub-tiles
//               (ds_read_b64_tr_b16 hands out V^T fragments).  Rows past the end of the slab are zeros.
//   S^T = K Q'^T   Q' = Q * scale*log2(e) in fp32 (one rounding per element), then split: scores arrive in the exp2 domain.
//   O^T += V^T P^T   same key permutation trick as the bf16 kernels: P never leaves its registers (split there into hi/lo).
//   softmax     optimistic, in two flavours selected per shape by choose_split():
//     run_fast  reference-free: p = exp2(s), nothing between the matrix core and v_exp_f32; software pipelined two tiles
//               deep with a static slot schedule (one matrix instruction, then its share of the vector work) so that the
//               single wave of a SIMD keeps both pipes busy -- see the comment at run_fast.
//     run_tile<OPT>  p = exp2(s - m0) with m0 the row maximum of the first tile folded into the accumulator the first
//               product starts from; phases in sequence, tiles above a causal wave's diagonal skipped (short causal rows).
//               fp32 P, l and O have 2^127 of head room either way: a row whose sum stays inside (2^-100, 2^100) with finite
//               outputs provably lost nothing.  Any other row sends its workgroup to
//     run_tile<!OPT>  the textbook running maximum (p <= 1), correct for every input; results are stored before the vote, a
//               rejected tile is simply overwritten.
#pragma once
#include "fa_bf16_common.h"
#include "fa_f32_exact.h"
#include "fa_kernels.h"
#include <type_traits>
#include <utility>

#ifndef FA_SPLIT_REF
#define FA_SPLIT_REF 1   // 0: experiment switch -- the fast pass runs reference-free (p = exp2(s), round 2)
#endif
#ifndef FA_SPLIT_STAMPS
#define FA_SPLIT_STAMPS 0   // 1: experiment switch (ablation builds) -- cycle stamps of the pipelined pass into the lse buffer, 8 floats per wave
#endif
#ifndef FA_SPLIT_CENTER
#define FA_SPLIT_CENTER 1  // 0: experiment switch -- keys are split as they come (rounds 1-4) instead of relative to a reference key
#endif
#ifndef FA_SPLIT_NOCVT
#define FA_SPLIT_NOCVT 0   // 1: timing-only ablation (ablation library, VERDICT r05 #2: the kill test of a one-time K / V split pre-pass) -- the conversion of
                           // a K / V piece costs 1 VALU per element instead of 3.5: hi = the rounded value, lo = hi with its exponent cleared (finite, tiny), no
                           // centring, no guard maximum.  Results are those of one-term operands (~1e-3), the verification still passes.
#endif
#ifndef FA_SPLIT_QK16
#define FA_SPLIT_QK16 1  // 0: experiment switch -- K and Q' of fp32 tensors as two BF16 terms (16 bits: rounds 1-4) instead of two FP16 terms
#endif

namespace fa {

constexpr int kKvSplit = 32;         // keys per tile
constexpr float kSplitLimit = 0x1p100f;  // optimistic pass: a row sum below this proves that no term overflowed
constexpr float kSplitTinyAcc = 0x1p-116f;   // sum of a row's unnormalised accumulators below this: products near (or below) the subnormals
// |q|_2 * |k|_inf * scale above which 16-bit operand terms no longer hold 1e-3.  100 until the fallback became per workgroup (round 4): a
// launch whose widest rows raised the old flag was redone as a whole, rows just under the limit included -- now those stay on the bf16 pipe,
// and rows at 95 .. 100 read up to 1.19e-3 in the LSE (soak seed 101 case 297; sweep over 126 launches, worst |O| / |lse| error of rows under
// L: 60 -> 3.6e-4 / 5.0e-4, 75 -> 5.9e-4 / 7.1e-4, 100 -> 7.0e-4 / 8.2e-4, 125 -> 8.4e-4 / 1.05e-3: profiles/r04_experiments.txt part 9)
// 75 would put unit-variance data at d = 128, scale 1 (up to 78) on the fallback: 90.
constexpr float kGuardLimit = 90.0f;   // (FA_SPLIT_QK16 = 0 builds only: the bf16-term form of rounds 1-4)
// fp16-term form (round 5): D * max|k| + sqrt(D) * max|q'|_2 above which the subnormal lo terms of small elements could add more than
// 2^-25 * 8192 = 2^-12 (1.7e-4 nat) to a logit -- in the WORST case: every element of q' below 2^-3 with the same sign of residual against keys
// that all sit at max|k - kbar|; random data stays 30 .. 50 times below its bound.  Unit-variance data: ~600 at d = 64, ~1000 at d = 128 (scale 1);
// keys with elements up to 60 at d = 128 (large activations in a few channels) still pass.  (2048 until the values were measured against
// LLM-like key magnitudes: max|k| = 20 at d = 128 would have sent every workgroup to the three times slower fp32 rows.)
constexpr float kSubnormalBudget = 8192.0f;

// running maximum of |a|, |b|: one instruction (abs as source modifiers)
__device__ __forceinline__ void absmax2(float& m, float a, float b)
{
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
}

template <int D>
struct SplitCfg {
    static constexpr int kRowBytes = 2 * D;
    static constexpr int kImageBytes = kKvSplit * kRowBytes;  // one bf16 image (hi or lo) of a K or V tile
    static constexpr int kStageBytes = 4 * kImageBytes;       // K_hi, K_lo, V_hi, V_lo
    static constexpr int kGroups = kKvSplit * D / 8;          // 8-element (32-byte fp32) groups per tile
};

// two-term bf16 split of a pair of fp32 values: hi = bf16(x) (round to nearest even), lo = bf16(x - hi).  The two
// conversions stay visible to hipcc (it pads the hazards around them: transcendental result -> VALU, VALU -> MFMA operand);
// the four instructions in between are asm so that they stay scalar: v_pk_add_f32 blocks the matrix pipe's issue.
__device__ __forceinline__ void split2(float a, float b, bf16x2& hi, bf16x2& lo)
{
    hi[0] = (__bf16)a;
    hi[1] = (__bf16)b;
    float la, lb;
    asm("v_lshlrev_b32 %0, 16, %2\n\tv_and_b32 %1, 0xffff0000, %2\n\tv_sub_f32 %0, %3, %0\n\tv_sub_f32 %1, %4, %1"
        : "=&v"(la), "=&v"(lb)
        : "v"(__builtin_bit_cast(unsigned, hi)), "v"(a), "v"(b));
    lo[0] = (__bf16)la;
    lo[1] = (__bf16)lb;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>), in order: compile-time indices for the slot schedule
template <class F, int... Is>
__device__ __forceinline__ void for_each_index(F&& f, std::integer_sequence<int, Is...>)
{
    (f(std::integral_constant<int, Is>{}), ...);
}

// the same split in plain C++ (this file is compiled without the SLP vectoriser, so the subtractions stay scalar): every
// instruction visible to hipcc's scheduler and hazard padding -- used by the software-pipelined fast pass
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ void split2c(float a, float b, bf16x2& hi, bf16x2& lo)
{
    const f32x2_t ab = {a, b};
    hi = __builtin_convertvector(ab, bf16x2);
    const f32x2_t hf = __builtin_convertvector(hi, f32x2_t);
    const f32x2_t lv = {a - hf[0], b - hf[1]};
    lo = __builtin_convertvector(lv, bf16x2);
}
// Two-term FP16 split of a pair of fp32 values (the operands of K.Q'^T for fp32 tensors, round 5): hi = f16(x) to nearest even, lo =
// f16(x - hi) -- 22 significant bits where two bf16 terms hold 16, on the same matrix pipe at the same rate (v_mfma_f32_32x32x16_f16).
// x - hi is ONE instruction: v_fma_mix_f32 reads the packed f16 half directly (fma(hi, -1, x), exact); v_cvt_pk_f16_f32 rounds to nearest
// even and keeps fp16 subnormals.  The pair travels in a bf16x2-typed register (bits only; the matrix instruction gives them their meaning).
// Range: |x| >= 65520 makes hi = inf and lo = -inf, whose products are NaN -- caught by the guard (see the kernel); |x - hi| < 2^-14 makes
// lo a subnormal with an absolute error <= 2^-25, bounded by the guard as well.
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ void split2h(float a, float b, bf16x2& hi, bf16x2& lo)
{
    const f32x2_t ab = {a, b};
    const f16x2_t h = __builtin_convertvector(ab, f16x2_t);
    const unsigned hp = __builtin_bit_cast(unsigned, h);
    f32x2_t lv;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lv[0]) : "v"(hp), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lv[1]) : "v"(hp), "v"(b));
    const f16x2_t l = __builtin_convertvector(lv, f16x2_t);
    hi = __builtin_bit_cast(bf16x2, h);
    lo = __builtin_bit_cast(bf16x2, l);
}
// x -= float(h) for four values against two packed fp16 pairs: v_fma_mix_f32 reads the fp16 half directly (fma(h, -1, x): one rounding, like
// v_sub_f32).  The reference rows of the key / value centring live in registers this way: half the registers of fp32 copies, the same
// instruction count (any fixed vector is a valid reference, so rounding it to fp16 costs nothing as long as the same values are added back).
__device__ __forceinline__ void sub_f16x4(f32x4& x, unsigned h01, unsigned h23)
{
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x[0]) : "v"(h01));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[1]) : "v"(h01));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x[2]) : "v"(h23));
    asm("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x[3]) : "v"(h23));
}
__device__ __forceinline__ unsigned pack_f16(float a, float b)
{
    const f32x2_t ab = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(ab, f16x2_t));
}

// F16 = the fp16 split (K and Q' of fp32 tensors), else the bf16 split (V, P; everything for bf16 tensors); ASM = the form whose four
// middle instructions are asm (the phase-structured pass), else plain C++ (the slot-scheduled pass)
template <bool F16, bool ASM>
__device__ __forceinline__ void split2x(float a, float b, bf16x2& hi, bf16x2& lo)
{
    if constexpr (FA_SPLIT_NOCVT != 0) {
        const f32x2_t ab = {a, b};
        unsigned h;
        if constexpr (F16) h = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, f16x2_t));
        else h = __builtin_bit_cast(unsigned, __builtin_convertvector(ab, bf16x2));
        hi = __builtin_bit_cast(bf16x2, h);
        lo = __builtin_bit_cast(bf16x2, h & (F16 ? 0x83ff83ffu : 0x807f807fu));
        return;
    }
    if constexpr (F16) split2h(a, b, hi, lo);
    else if constexpr (ASM) split2(a, b, hi, lo);
    else split2c(a, b, hi, lo);
}
template <bool F16, bool ASM>
__device__ __forceinline__ void split8x(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo)
{
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        bf16x2 h, l;
        split2x<F16, ASM>(a[i], a[i + 1], h, l);
        hi[i] = h[0], hi[i + 1] = h[1], lo[i] = l[0], lo[i + 1] = l[1];
        split2x<F16, ASM>(b[i], b[i + 1], h, l);
        hi[i + 4] = h[0], hi[i + 5] = h[1], lo[i + 4] = l[0], lo[i + 5] = l[1];
    }
}
__device__ __forceinline__ void split8c(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) { split8x<false, false>(a, b, hi, lo); }
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) { split8x<false, true>(a, b, hi, lo); }

__device__ __forceinline__ void split_p(const f32x16& s, int base, bf16x8& hi, bf16x8& lo)
{
#pragma unroll
    for (int i = 0; i < 8; i += 2) {
        bf16x2 h, l;
        split2(s[base + i], s[base + i + 1], h, l);
        hi[i] = h[0], hi[i + 1] = h[1], lo[i] = l[0], lo[i + 1] = l[1];
    }
}

// The K.Q'^T chain is asm: its first product reads the accumulator from registers DISTINCT from its destination (the
// builtin insists on D == C and copies the 16 registers of C first), and hipcc must not be tempted to park the scores in
// AGPRs between products.  hipcc knows nothing about the inside of an asm statement, hence
//   s_nop 1   in front of every product: an operand may have been copied into place (v_accvgpr_read_b32, v_mov_b32) by
//             the instruction before, and VALU write -> MFMA read needs two wait states;
//   scores_retire() after the chain, before any VALU instruction may read the scores.
// F16: the operand registers hold fp16 pairs (split2h) and the product is v_mfma_f32_32x32x16_f16 -- same shape, same rate
template <bool F16 = false>
__device__ __forceinline__ void mfma_from(f32x16& d, const bf16x8& a, const bf16x8& b, const f32x16& c)
{
    if constexpr (F16) asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    else asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
}
template <bool F16 = false>
__device__ __forceinline__ void mfma_from_zero(f32x16& d, const bf16x8& a, const bf16x8& b)
{
    if constexpr (F16) asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    else asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
}
template <bool F16 = false>
__device__ __forceinline__ void mfma_acc(f32x16& d, const bf16x8& a, const bf16x8& b)
{
    if constexpr (F16) asm("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
    else asm("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}
// the builtin form (slot-scheduled pass)
template <bool F16>
__device__ __forceinline__ f32x16 mfma_qk(const bf16x8& a, const bf16x8& b, const f32x16& c)
{
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// four consecutive output values: fp32, or bf16 when the caller's O is bf16 (p.o_is_bf16, bf16 tensors only)
__device__ __forceinline__ void store4(const FwdParams& p, int64_t off, const f32x4& v)
{
    if (p.o_is_bf16) {
        bf16x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = (__bf16)v[e];
        *(bf16x4*)((__bf16*)p.o + off) = b;
    } else {
        *(f32x4*)((float*)p.o + off) = v;
    }
}

// IN_BF16: bf16 tensors through the same machinery ("accurate" bf16 mode, FA_KERNEL_SPLIT with a bf16 dtype): K and V are
// exact in one bf16 term, so only Q' = Q*scale*log2(e) and P are split -- two products per contraction instead of three, no
// conversion work -- and the result is as close to the fp64 oracle as with fp32 tensors (P is not rounded to 8 bits).
template <int D, int NWAVES, int QB, bool CAUSAL, int MINBLOCKS, bool PIPE, bool IN_BF16>
__global__ __launch_bounds__(NWAVES* kWave, MINBLOCKS) void fa_fwd_f32_split_kernel(FwdParams p)
{
    using C = SplitCfg<D>;
    using T = std::conditional_t<IN_BF16, __bf16, float>;   // element type of Q, K, V
    constexpr int NPROD = IN_BF16 ? 2 : 3;                  // matrix products per contraction
    constexpr bool QK16 = !IN_BF16 && FA_SPLIT_QK16;       // K and Q' as two FP16 terms (22 bits), K.Q'^T on v_mfma_f32_32x32x16_f16
    constexpr int KS = D / 16;   // k-steps of S^T = K Q^T
    constexpr int DB = D / 32;   // 32-wide blocks of the head dim in O^T
    constexpr int NT = NWAVES * kWave;
    constexpr int BM = NWAVES * QB * 32;
    constexpr int GPT = (C::kGroups + NT - 1) / NT;  // groups per thread and tile

    __shared__ __attribute__((aligned(1024))) char smem[2 * C::kStageBytes];
    __shared__ __attribute__((aligned(16))) float s_kref[D];   // key centering, eight-wave tiling only (no registers to spare): the reference key
    __shared__ __attribute__((aligned(16))) float s_vref[D];   // ... and the reference value row
    __shared__ unsigned s_kmax;   // range guard: max |k| over the keys this workgroup reads, as the bits of a non-negative float

    if (flag_says_skip(p)) return;   // conditional fallback of a launch chain (bf16 tensors behind the fp16-P kernel)
    unsigned long long st_in = 0, st_req = 0, st_loop0 = 0, st_loop1 = 0;
    if constexpr (FA_SPLIT_STAMPS != 0) st_in = __builtin_readcyclecounter();

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane & 31, hi = lane >> 5;
    constexpr bool GUARD = !IN_BF16 && !FA_SPLIT_NOCVT;   // tracked for every fp32 launch (4 VALU per K piece); acted upon under flag_mode 4 (FA_KERNEL_AUTO: the
                                       // workgroup redoes its rows in fp32 arithmetic on the spot) and 3 (ablation chains: raise the word)
    float kmax = 0.0f;
    if (GUARD && tid == 0) s_kmax = 0u;   // ordered before the first atomic by the barriers of the main loop

    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles;
    int qt = w % p.q_tiles;
    if (CAUSAL) qt = causal_tile(p, qt);  // longest (latest) q tiles first, or heavy + light pairs (FwdParams::alt_order)
    const int n = p.n;
    const int q0 = qt * BM + wave * (QB * 32);

    const int b = slab / p.heads, h = slab % p.heads;
    const T* qg = (const T*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
    const T* kg = (const T*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const T* vg = (const T*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    const int64_t o_slab = b * p.o_batch_stride + h * p.o_head_stride;   // elements (fp32 or bf16 output, p.o_is_bf16)

    // keys of this workgroup: all n, or (key-split launches, fa_launch.cpp) its share [h * n_kv, min((h + 1) * n_kv, n_kv_total)) --
    // kv_head_stride carries the offset, nk bounds the LOCAL key indices (local key i is key kbeg + i of the slab; causal: local key <=
    // local row q - kbeg).  Causal shares are multiples of the tile height: a share starts at or below the tile's first row (every row sees
    // its first key) or lies entirely above the tile -- an empty share, which stores lse = -inf for its rows (the combine gives it weight
    // 0 and never reads its O) and is done.
    const int kbeg = p.n_kv > 0 ? h * p.n_kv : 0;
    const int nk = p.n_kv > 0 ? min(p.n_kv, p.n_kv_total - kbeg) : n;
    int kv_end = nk;
    if (CAUSAL) kv_end = min(nk, qt * BM + BM - kbeg);
    if (CAUSAL && kv_end <= 0) {   // workgroup-uniform
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int qi = q0 + qb * 32 + lq;
            if (qi < n && p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = -INFINITY;
        }
        return;
    }
    const int q0l = q0 - kbeg;   // first row of this wave in local key coordinates
    const int nt = (kv_end + kKvSplit - 1) / kKvSplit;

    // ---- this thread's pieces of a tile: group g = tid + i * NT -> (row, 8-column slot) of K and of V
    // The K piece and the V piece of a thread are DIFFERENT (row, slot) pairs: each image wants the eight lanes of one ds_write_b128
    // group on eight different 16-byte bank groups.  K image (row-major, slots swizzled): consecutive lanes take consecutive slots
    // of one row (128 contiguous bytes).  V image ([key/4][col/16][4][16] sub-tiles of 128 bytes): consecutive lanes take the four
    // rows of a sub-tile and its two 8-column halves -- with the K mapping the four lanes of a row that share (col/8) & 1 landed on the
    // same banks (4-way conflicts on every V write: SQ_LDS_BANK_CONFLICT 25 M against 14.7 M LDS instructions in round 1).
    int g_krow[GPT], g_vrow[GPT], g_ksrc[GPT], g_vsrc[GPT], g_kdst[GPT], g_vdst[GPT];
    bool g_on[GPT];
#pragma unroll
    for (int i = 0; i < GPT; ++i) {
        const int g = (tid + i * NT) % C::kGroups;   // d = 32: the upper half of the workgroup duplicates the lower half's pieces
        g_on[i] = true;
        const int row = g / (D / 8), c8 = g % (D / 8);
        g_krow[i] = row;
        g_ksrc[i] = row * p.kv_row_stride + c8 * 8;
        g_kdst[i] = row * C::kRowBytes + ((c8 ^ k_swizzle<D>(row)) * 16);
        const int r4 = g & 3, half = (g >> 2) & 1, c16 = (g >> 3) % (D / 16), rq = g / (4 * (D / 8));
        const int vrow = rq * 4 + r4, vc8 = c16 * 2 + half;
        g_vrow[i] = vrow;
        g_vsrc[i] = vrow * p.kv_row_stride + vc8 * 8;
        g_vdst[i] = 2 * C::kImageBytes + (rq * (D / 16) + c16) * 128 + r4 * 32 + half * 16;
    }
    // KEY CENTERING (fp32 tensors, round 5).  softmax(q.k_j) = softmax(q.(k_j - kbar)) for ANY fixed vector kbar: the row constant q.kbar
    // cancels in O and is added back to the log-sum-exp.  What it buys: fp32 accumulation rounds (the matrix core truncates) every partial
    // sum at the magnitude it has, so a logit of ~1500 carries ~1e-3 of error whatever the operand terms hold -- and when that magnitude is
    // COMMON MODE (constant-component rows, a broadcast token, inputs with an offset: every key close to every other, the family of
    // tests/adversarial.py) the softmax only needs the DIFFERENCES, which k_j - kbar delivers exactly (one fp32 subtraction, exact for
    // nearly equal values) before anything is rounded to 22 bits or accumulated.  kbar = the coordinate-wise MEDIAN of three keys of the
    // share (first, middle, last): an outlier key (an attention sink in position 0) cannot become the reference, |kbar_c| never exceeds the
    // second largest of three actual values (centred magnitudes are at most twice the uncentred ones), and it costs three row loads and
    // one v_med3_f32 per column per workgroup + one v_sub_f32 per converted key element.  Coherent family at d = 128, causal: 3.5e-3 ->
    // below 1e-4 (profiles/r05_family_centered.txt); the reference's own fp32 FMA chain reads 5.7e-3 there.
    constexpr bool CENTER = !IN_BF16 && FA_SPLIT_CENTER && !FA_SPLIT_NOCVT;
    // (local key indices, among the keys THIS workgroup reads: a causal tile takes them below its own horizon -- until round 6 they were
    // rows 0, nk / 2, nk - 1 of the share, i.e. future tokens or padding for most causal tiles, and two outliers among rows no row of the
    // tile attends to could become the reference of everything it does attend to: ADVICE r05)
    const int kref_r1 = kv_end >> 1, kref_r2 = kv_end - 1;
    // The three rows of each reference are REQUESTED here (threads 0 .. D/4 - 1 the key's columns, the next D/4 the value's) and reduced to their
    // median further down, behind the requests for K(0) and Q: one memory round trip for all three instead of two in a row (round 6: the
    // reference rows used to be waited for, written to LDS and synchronised on before anything else was asked for -- c2 spends two rounds of
    // workgroups of 16 tiles each, and every round paid that latency; profiles/r06_exp13_prologue.txt)
    f32x4 ref_a = {0.0f, 0.0f, 0.0f, 0.0f}, ref_b = ref_a, ref_c = ref_a;
    auto request_reference_rows = [&]() {
        if constexpr (CENTER) {
            if (tid < D / 2) {
                const float* src = (const float*)(tid < D / 4 ? (const void*)kg : (const void*)vg) + (tid % (D / 4)) * 4;
                ref_a = *(const f32x4*)src;
                ref_b = *(const f32x4*)(src + (int64_t)kref_r1 * p.kv_row_stride);
                ref_c = *(const f32x4*)(src + (int64_t)kref_r2 * p.kv_row_stride);
            }
        }
    };
    // this thread's K pieces all sit in the same 8 columns (NT is a multiple of D / 8, or kGroups divides it)
    const int kref_col = ((tid % C::kGroups) % (D / 8)) * 8;
    // KREF_REG: the thread's eight reference values live in registers; the eight-wave D = 128 tiling (256 registers per lane, all in use)
    // reads them from LDS in front of every conversion instead -- behind the V piece's conversion, which hides the latency
    constexpr bool KREF_REG = !(D == 128 && NWAVES == 8);
    // (the pipelined pass subtracts the references from registers only: an instantiation without them would skip the centring silently and
    // still add vbar back to O -- ADVICE r05)
    static_assert(!PIPE || KREF_REG || !CENTER, "the pipelined pass needs the reference rows in registers");
    unsigned krefp[4] = {0u, 0u, 0u, 0u}, vrefp[4] = {0u, 0u, 0u, 0u};   // the thread's eight reference values each, as packed fp16 pairs
    // VALUE CENTERING (round 5).  sum_j w_j v_j = vbar + sum_j w_j (v_j - vbar) for softmax weights (they add up to one): the kernel splits
    // v_j - vbar into its two bf16 terms, so the 16 bits cover the SPREAD of V and not an offset all values share -- V = 100 + N(0, 1) under a
    // peaked softmax read 1.6e-3 (the terms' 3 * 2^-17 * max|v|) where the reference's own fp32 recurrence reads 1.4e-4, V = 1000 + N(0, 1)
    // 1.6e-2 (profiles/r05_v_offset.txt) -- and vbar, the same median of three rows as for the keys, is added back to O in the epilogue.
    // A V that is CONSTANT over the share is all zeros after centring: zero accumulators, which the workgroup tells from underflow by looking at
    // its share of V (see the vote behind the first attempt).
    // this thread's V pieces all sit in the same 8 columns: (c16 * 2 + half) * 8 of its group index
    const int vref_col = ((((tid % C::kGroups) >> 3) % (D / 16)) * 2 + (((tid % C::kGroups) >> 2) & 1)) * 8;
    f32x4 kst[GPT][2], vst[GPT][2];
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            const bool kok = g_on[i] && (kv0 + g_krow[i] < nk), vok = g_on[i] && (kv0 + g_vrow[i] < nk);
            const int64_t koff = (int64_t)kv0 * p.kv_row_stride + g_ksrc[i], voff = (int64_t)kv0 * p.kv_row_stride + g_vsrc[i];
            const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            if constexpr (IN_BF16) {   // eight bf16 values = one 16-byte register group, passed through unchanged
                kst[i][0] = kok ? *(const f32x4*)(kg + koff) : z;
                vst[i][0] = vok ? *(const f32x4*)(vg + voff) : z;
            } else {
                kst[i][0] = kok ? *(const f32x4*)(kg + koff) : z;
                kst[i][1] = kok ? *(const f32x4*)(kg + koff + 4) : z;
                vst[i][0] = vok ? *(const f32x4*)(vg + voff) : z;
                vst[i][1] = vok ? *(const f32x4*)(vg + voff + 4) : z;
            }
        }
    };
    auto store_tile = [&](char* stage, int kv0) {   // kv0: first key of the tile load_tile fetched
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            if (!g_on[i]) continue;
            if constexpr (IN_BF16) {
                *(f32x4*)(stage + g_kdst[i]) = kst[i][0];
                *(f32x4*)(stage + g_vdst[i]) = vst[i][0];
            } else {
                bf16x8 h8, l8;
                f32x4 r0 = {0.0f, 0.0f, 0.0f, 0.0f}, r1 = r0;
                if constexpr (CENTER && !KREF_REG) r0 = *(const f32x4*)&s_kref[kref_col], r1 = *(const f32x4*)&s_kref[kref_col + 4];
                if constexpr (CENTER && KREF_REG) sub_f16x4(vst[i][0], vrefp[0], vrefp[1]), sub_f16x4(vst[i][1], vrefp[2], vrefp[3]);
                if constexpr (CENTER && !KREF_REG) vst[i][0] -= *(const f32x4*)&s_vref[vref_col], vst[i][1] -= *(const f32x4*)&s_vref[vref_col + 4];
                split8(vst[i][0], vst[i][1], h8, l8);
                *(bf16x8*)(stage + g_vdst[i]) = h8;
                *(bf16x8*)(stage + C::kImageBytes + g_vdst[i]) = l8;
                if constexpr (CENTER && KREF_REG) sub_f16x4(kst[i][0], krefp[0], krefp[1]), sub_f16x4(kst[i][1], krefp[2], krefp[3]);
                if constexpr (CENTER && !KREF_REG) kst[i][0] -= r0, kst[i][1] -= r1;
                if constexpr (GUARD) {
                    // (key rows past the share were loaded as zeros and are now -kbar: masked in the scores, and kept out of the range guard's max |k|)
                    if (!CENTER || kv0 + g_krow[i] < nk) {
#pragma unroll
                        for (int e = 0; e < 4; e += 2) absmax2(kmax, kst[i][0][e], kst[i][0][e + 1]), absmax2(kmax, kst[i][1][e], kst[i][1][e + 1]);
                    }
                }
                split8x<QK16, true>(kst[i][0], kst[i][1], h8, l8);
                *(bf16x8*)(stage + g_kdst[i]) = h8;
                *(bf16x8*)(stage + C::kImageBytes + g_kdst[i]) = l8;
            }
        }
    };

    // loads of tile t through buffer descriptors: rows past the end of the slab come back as zeros from the bounds check,
    // so there is no branch (a branch inside the loop body lets LLVM sink vector work out of its slot)
    constexpr unsigned ES = sizeof(T);
    const unsigned slab_bytes = ((unsigned)(nk - 1) * (unsigned)p.kv_row_stride + D) * ES;
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kg, 0, slab_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vg, 0, slab_bytes, 0x00020000);
    const unsigned tile_step = (unsigned)kKvSplit * (unsigned)p.kv_row_stride * ES;
    // key centring: rows past the share (the ragged tail of its last tile, the tiles the pipeline reads ahead of the end) are read as
    // the share's LAST key instead of the descriptor's zeros -- they are masked or never used, but zeros would leave the centring as
    // -kbar and raise the range guard's max |k| for nothing.  Offsets grow with the row for a fixed column: one add + one min per piece.
    unsigned k_last[GPT];
#pragma unroll
    for (int i = 0; i < GPT; ++i) k_last[i] = ((unsigned)(nk - 1) * (unsigned)p.kv_row_stride + (unsigned)kref_col) * ES;
    auto load_k = [&](int t) {
        const unsigned soff = (unsigned)t * tile_step;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            if constexpr (CENTER) {
                const unsigned off = min(g_ksrc[i] * ES + soff, k_last[i]);
                kst[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, off, 0, 0));
                kst[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, off + 16, 0, 0));
            } else {
                kst[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, g_ksrc[i] * ES, soff, 0));
                if constexpr (!IN_BF16)
                    kst[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, g_ksrc[i] * ES + 16, soff, 0));
            }
        }
    };
    auto load_v = [&](int t) {
        const unsigned soff = (unsigned)t * tile_step;
#pragma unroll
        for (int i = 0; i < GPT; ++i) {
            vst[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, g_vsrc[i] * ES, soff, 0));
            if constexpr (!IN_BF16)
                vst[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, g_vsrc[i] * ES + 16, soff, 0));
        }
    };
    // ---- everything the prologue needs from memory is requested before anything is waited for: the reference rows, K(0) (pipelined pass), Q
    request_reference_rows();
    if constexpr (PIPE) load_k(0);

    // ---- Q' fragments (B operand of S^T = K Q'^T), hi and lo: lane (lq, hi) holds Q'[q][16*ks + 8*hi .. +7]
    bf16x8 qh[QB][KS], ql[QB][KS];
    float qn2 = 0.0f;   // guard: largest squared 2-norm of Q' among this lane's rows (its half of each row; halves are added below)
    const bool want_crow = CENTER && p.lse != nullptr && FA_SPLIT_STAMPS == 0;   // (uniform; key-split launches always carry an lse)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        float qs = 0.0f;
        const int qrow = min(q0 + qb * 32 + lq, n - 1);
        const T* qr = qg + (int64_t)qrow * p.q_row_stride + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f32x4 a, c;
            if constexpr (IN_BF16) {
                const bf16x8 q8 = *(const bf16x8*)(qr + ks * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (float)q8[e] * p.scale_log2e, c[e] = (float)q8[e + 4] * p.scale_log2e;
            } else {
                a = *(const f32x4*)(qr + ks * 16) * p.scale_log2e;
                c = *(const f32x4*)(qr + ks * 16 + 4) * p.scale_log2e;
            }
            if constexpr (GUARD) {
#pragma unroll
                for (int e = 0; e < 4; ++e) qs = fmaf(a[e], a[e], fmaf(c[e], c[e], qs));
            }
            split8x<QK16, true>(a, c, qh[qb][ks], ql[qb][ks]);
        }
        if constexpr (GUARD) qn2 = fmaxf(qn2, xhalf_sum(qs));
    }
    if constexpr (CENTER) {
        // both reference rows are computed ONCE per workgroup (threads 0 .. D/4 - 1 the key's, the next D/4 the value's: three row loads and
        // four v_med3_f32 each) and live in LDS: the conversions take their eight columns from there (into registers, or -- eight-wave
        // tiling -- in front of every use), the epilogue its add-back and the row constant of the log-sum-exp
        if (tid < D / 2) {   // (rounded to fp16 and clamped to its range: ANY fixed vector is a valid reference; see sub_f16x4)
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = (float)(_Float16)__builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(ref_a[e], ref_b[e], ref_c[e]), -65504.0f, 65504.0f);
            *(f32x4*)&(tid < D / 4 ? s_kref : s_vref)[(tid % (D / 4)) * 4] = r;
        }
        __syncthreads();
        if constexpr (KREF_REG) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                krefp[e] = pack_f16(s_kref[kref_col + 2 * e], s_kref[kref_col + 2 * e + 1]);
                vrefp[e] = pack_f16(s_vref[vref_col + 2 * e], s_vref[vref_col + 2 * e + 1]);
            }
        }
    }
    // key centering: q'.kbar of this lane's row (exp2 domain), added back to the log-sum-exp in the epilogues -- with the operand the
    // matrix core sees (hi + lo, exact in fp32), accumulated in fp64: once per row and tile, and only when an lse is asked for
    auto crow_of = [&](auto qbc) -> float {   // (compile-time indices: a runtime index into the Q' fragments would put them in scratch)
        constexpr int qb = decltype(qbc)::value;
        if (!want_crow) return 0.0f;
        double cacc = 0.0;
        for_each_index([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            const f32x4 r0 = *(const f32x4*)&s_kref[ks * 16 + hi * 8], r1 = *(const f32x4*)&s_kref[ks * 16 + hi * 8 + 4];
            typedef __attribute__((ext_vector_type(8))) float f32x8_t;
            f32x8_t qe;   // the operand the matrix core sees: hi + lo, exact in fp32
            if constexpr (QK16) {
                qe = __builtin_convertvector(__builtin_bit_cast(f16x8, qh[qb][ks]), f32x8_t) + __builtin_convertvector(__builtin_bit_cast(f16x8, ql[qb][ks]), f32x8_t);
            } else {
                qe = __builtin_convertvector(qh[qb][ks], f32x8_t) + __builtin_convertvector(ql[qb][ks], f32x8_t);
            }
            for_each_index([&](auto ec) {
                constexpr int e = decltype(ec)::value;
                cacc += (double)qe[e] * (double)r0[e] + (double)qe[e + 4] * (double)r1[e];
            }, std::make_integer_sequence<int, 4>{});
        }, std::make_integer_sequence<int, KS>{});
        return xhalf_sum((float)cacc);
    };

    const int k_row_off = lq * C::kRowBytes;
    const int k_g = hi ^ k_swizzle<D>(lq);
    int k_off[KS];   // per-lane byte offset of the K fragment of k-step ks inside an image (swizzle resolved once)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_off[ks] = k_row_off + (((2 * ks) ^ k_g) * 16);
    const int li = lane & 15;
    const int v_lane_off = (hi * (D / 16) + ((lane >> 4) & 1)) * 128 + (li >> 2) * 32 + (li & 3) * 8;

    // The whole tile, OPT = optimistic pass (fixed reference m0) or the textbook redo.  Returns whether this lane's rows
    // came out inside the range the optimistic pass can prove.
    // guard: a NaN in a row SUM of the first attempt (the running maxima above drop NaNs: v_max3_f32 and fmaxf return the other operand) -- a
    // NaN or inf in Q or K, or an fp16 term out of range, reaches it -- sends the workgroup to fp32 arithmetic.  (Round 5: a NaN OUTPUT alone
    // no longer does: rows that outgrow the optimistic window have l = +inf and O = inf / inf, and belong to the textbook redo on the same
    // pipes, not to the three times slower fallback; a NaN in V gives the NaN it must give on either path.)
    bool saw_nan = false;
    bool hard_fail = false;   // fast pass: some row failed its verification for another reason than accumulators that are EXACTLY zero
    auto run_tile = [&](auto opt_c) -> bool {
        constexpr bool OPT = decltype(opt_c)::value;
        // MREG: -m_ref of a row lives in a 16-register tuple, the accumulator the first product of every tile starts from (the
        // subtraction is free).  The eight-wave D = 128 tiling has no 16 registers to spare (it spilled 16-24 bytes per lane): there
        // the first product starts from zero and m_ref is subtracted by the VALU -- 16 instructions per tile in a loop that is
        // bound by its 48 matrix instructions per tile.
        constexpr bool MREG = !(D == 128 && NWAVES == 8);
        f32x16 o[QB][DB];
        f32x16 minit[MREG ? QB : 1];   // -m_ref of this lane's row in all 16 registers
        float m[QB], l[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            m[qb] = 0.0f;
            l[qb] = 0.0f;
            if constexpr (MREG) {
#pragma unroll
                for (int r = 0; r < 16; ++r) minit[qb][r] = 0.0f;
            }
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.0f;
        }

        // scores of one tile for all QB blocks.  ORDER (round 5): the accumulator starts at -m_ref (as large as the row's widest logit) and
        // every matrix instruction rounds its result at the magnitude the partial sum has THEN -- so the hi.hi products of all k-steps come
        // first (the partial sum walks from -m_ref down to the score's own small magnitude) and the cross terms, 2^-11 of them, are added
        // last, where an ulp is small.  Interleaved per k-step (rounds 1-4) two thirds of the roundings happened at full magnitude: the
        // coherent-input family of tests/adversarial.py read 1.9e-3 at d = 128 from that alone.  Without MREG the chain starts from zero
        // and grows: there the cross terms go first.
        auto scores = [&](const char* kh_lds, f32x16 (&s)[QB]) {
            auto hh_pass = [&](bool first_pass) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kfh = *(const bf16x8*)(kh_lds + k_off[ks]);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        if (first_pass && ks == 0) {
                            if constexpr (MREG) mfma_from<QK16>(s[qb], kfh, qh[qb][ks], minit[qb]);
                            else mfma_from_zero<QK16>(s[qb], kfh, qh[qb][ks]);
                        } else {
                            mfma_acc<QK16>(s[qb], kfh, qh[qb][ks]);
                        }
                    }
                }
            };
            auto cross_pass = [&](bool first_pass) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8 kfh = *(const bf16x8*)(kh_lds + k_off[ks]);
                    if constexpr (IN_BF16) {   // K exact in one term: K . Q'_lo
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) {
                            if (first_pass && ks == 0) mfma_from_zero(s[qb], kfh, ql[qb][ks]);
                            else mfma_acc(s[qb], kfh, ql[qb][ks]);
                        }
                    } else {
                        const bf16x8 kfl = *(const bf16x8*)(kh_lds + C::kImageBytes + k_off[ks]);
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) {
                            if (first_pass && ks == 0) mfma_from_zero<QK16>(s[qb], kfl, qh[qb][ks]);
                            else mfma_acc<QK16>(s[qb], kfl, qh[qb][ks]);
                            mfma_acc<QK16>(s[qb], kfh, ql[qb][ks]);
                        }
                    }
                }
            };
            if constexpr (MREG) {
                hh_pass(true);
                cross_pass(false);
            } else {
                cross_pass(true);
                hh_pass(false);
            }
            // let the last product retire (19 wait states cover its 8 passes), tied to the registers the chain writes
            if constexpr (QB == 1) asm volatile("s_nop 15\n\ts_nop 2" : "+v"(s[0]));
            else asm volatile("s_nop 15\n\ts_nop 2" : "+v"(s[0]), "+v"(s[QB - 1]));
            if constexpr (!MREG) {
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[qb][r] -= m[qb];
            }
        };
        auto mask = [&](f32x16& sq, int kv0, int qi) {
            asm volatile("; mask" ::: "memory");  // not speculatable: keeps the caller's wave-uniform `if` a real branch
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + 4 * hi + (r & 3) + 8 * (r >> 2);
                if ((key >= nk) || (CAUSAL && key > qi)) sq[r] = -INFINITY;
            }
        };
        auto row_max = [&](const f32x16& sq) {
            float mx = fmaxf(sq[0], sq[1]);
#pragma unroll
            for (int r = 2; r < 16; r += 2) mx = max3_safe(mx, sq[r], sq[r + 1]);
            return xhalf_max(mx);
        };

        load_tile(0);
        store_tile(smem, 0);
        __syncthreads();

        if constexpr (OPT) {
            // reference of each row: the maximum over the first tile (every row sees key 0, so it is finite for finite inputs)
            f32x16 s[QB];
            scores(smem, s);
            const bool need_mask = (kKvSplit > nk) || (CAUSAL && (kKvSplit - 1 > q0l));
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (need_mask) mask(s[qb], 0, q0l + qb * 32 + lq);
                m[qb] = row_max(s[qb]);
                if constexpr (MREG) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) minit[qb][r] = -m[qb];
                }
            }
        }

        // one K/V tile; STG (compile-time) is the LDS stage it lives in, so every fragment read is base register + immediate
        auto step = [&](auto stg_c, int j) {
            constexpr int STG = decltype(stg_c)::value;
            const bool more = j + 1 < nt;
            if (more) load_tile((j + 1) * kKvSplit);
            const int kv0 = j * kKvSplit;
            const char* kh_lds = smem + STG * C::kStageBytes;
            const char* vh_lds = kh_lds + 2 * C::kImageBytes;

            if (!(CAUSAL && kv0 > q0l + QB * 32 - 1)) {   // else: tile entirely above this wave's diagonal
                f32x16 s[QB];
                scores(kh_lds, s);   // S'^T = K Q'^T - m_ref

                // ================= softmax (registers only) =================
                const bool need_mask = (kv0 + kKvSplit > nk) || (CAUSAL && (kv0 + kKvSplit - 1 > q0l));
                bf16x8 ph[QB][2], pl[QB][2];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    if (need_mask) mask(s[qb], kv0, q0l + qb * 32 + lq);
                    if constexpr (!OPT) {
                        const float mx = row_max(s[qb]);                  // row maximum relative to m_ref
                        const bool grow = (j == 0) || (mx > 0.0f);        // tile 0 sets the reference (m_ref starts at 0)
                        if (__builtin_amdgcn_ballot_w64(grow) != 0) {     // wave-uniform
                            asm volatile("; rescale" ::: "memory");
                            const float delta = grow ? mx : 0.0f;
                            const float alpha = (j == 0) ? 0.0f : fast_exp2(-delta);
                            m[qb] += delta;
                            l[qb] *= alpha;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                s[qb][r] -= delta;
                                if constexpr (MREG) minit[qb][r] = -m[qb];
                            }
#pragma unroll
                            for (int db = 0; db < DB; ++db)
#pragma unroll
                                for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
                        }
                    }
                    float rs0 = 0.0f, rs1 = 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        s[qb][r] = fast_exp2(s[qb][r]);
                        s[qb][r + 1] = fast_exp2(s[qb][r + 1]);
                        // scalar adds (v_pk_add_f32 blocks the matrix pipe's issue); the s_nop is the wait state a
                        // transcendental result needs before a plain VALU instruction may read it -- invisible to hipcc here
                        asm("s_nop 0\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3"
                            : "+v"(rs0), "+v"(rs1) : "v"(s[qb][r]), "v"(s[qb][r + 1]));
                    }
                    l[qb] += rs0 + rs1;
                    split_p(s[qb], 0, ph[qb][0], pl[qb][0]);
                    split_p(s[qb], 8, ph[qb][1], pl[qb][1]);
                }

                // ================= O^T += V^T P^T, three products =================
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const int off0 = ((4 * t + 0) * (D / 16) + 2 * db) * 128;
                        const int off1 = ((4 * t + 2) * (D / 16) + 2 * db) * 128;
                        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vh_lds + v_lane_off + off0));
                        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vh_lds + v_lane_off + off1));
                        const bf16x8 vfh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                        if constexpr (!IN_BF16) {
                            const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vh_lds + C::kImageBytes + v_lane_off + off0));
                            const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vh_lds + C::kImageBytes + v_lane_off + off1));
                            const bf16x8 vfl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                            for (int qb = 0; qb < QB; ++qb) o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph[qb][t], o[qb][db], 0, 0, 0);
                        }
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) {
                            o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl[qb][t], o[qb][db], 0, 0, 0);
                            o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph[qb][t], o[qb][db], 0, 0, 0);
                        }
                    }
            }

            // stage STG^1 was last read in step j-1; every wave has passed the barrier that ended that step
            if (more) store_tile(smem + (STG ^ 1) * C::kStageBytes, (j + 1) * kKvSplit);
            __syncthreads();
        };

        int j = 0;
        for (; j + 1 < nt; j += 2) {
            step(std::integral_constant<int, 0>{}, j);
            step(std::integral_constant<int, 1>{}, j + 1);
        }
        if (j < nt) step(std::integral_constant<int, 0>{}, j);

        // ================= epilogue: O / l, store =================
        mfma_drain();  // the loop exit is a branch: the last P.V MFMAs may still be in flight
        float crows[QB];
        for_each_index([&](auto qbc) { crows[decltype(qbc)::value] = crow_of(qbc); }, std::make_integer_sequence<int, QB>{});
        bool ok = true;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float lt = xhalf_sum(l[qb]);
            const float inv = 1.0f / lt;
            const int qi = q0 + qb * 32 + lq;
            const float crow = crows[qb];
            float mag = 0.0f;
            if (qi < n) {
                const int64_t o_off = o_slab + (int64_t)qi * p.o_row_stride + 4 * hi;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            pk[e] = o[qb][db][4 * g + e] * inv;
                            if (OPT) mag += fabsf(pk[e]);
                        }
                        if constexpr (CENTER) pk += *(const f32x4*)&s_vref[db * 32 + 8 * g + 4 * hi];   // value centering: the reference row back in
                        store4(p, o_off + db * 32 + 8 * g, pk);
                    }
                if (FA_SPLIT_STAMPS == 0 && p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = ((m[qb] + __builtin_amdgcn_logf(lt)) + crow) * kLn2;
                if (OPT) ok = ok && (lt < kSplitLimit) && (mag < INFINITY);   // false for NaN as well
                if (GUARD) saw_nan = saw_nan || (lt != lt);   // (a NaN row SUM: NaN logits -- an overflowed window gives +inf, which the redo handles)
            }
        }
        return ok;
    };

    // -----------------------------------------------------------------------------------------------------------------
    // The fast pass: reference-free optimistic softmax, p = exp2(s) -- no maximum, no subtraction, nothing between the
    // matrix core and v_exp_f32 -- software pipelined two tiles deep.  One wave cannot overlap its own dependent phases, and
    // a second wave on the SIMD does not help either (a wave waiting to issue an MFMA holds the vector issue port), so
    // iteration j gives the matrix pipe two jobs that do not depend on this iteration's VALU work:
    //      matrix pipe   S(j+1) = K(j+1) Q'^T          and    O += V(j-1)^T P(j-1)^T
    //      VALU          P(j) = split(exp2(S(j)))      and    the fp32 -> hi/lo conversion of K(j+2), V(j)
    // in ONE basic block of plain builtins, which hipcc's scheduler interleaves (sched_group_barrier pins the rhythm).
    // K(j+2) replaces K(j) and V(j) replaces V(j-2) in their two-stage LDS rings at the end of the iteration; one barrier per
    // tile.  A causal wave does not skip the (at most 2*QB*NWAVES - 1) tiles above its diagonal, it masks them: no branches.
    // The exponent reference of round 3 puts every P near 2^-B (B = 109 - ceil(log2 n): 96 at n = 8192): a row whose sum ends up in
    // (2^-(B + 2), 2^100) with finite outputs lost nothing to RANGE at the top; at the bottom the accumulators hold ~2^-B l |O|, so the
    // products of the terms that matter stay normal fp32 numbers only while |v| is above ~2^-30 -- a row whose unnormalised accumulators
    // are all tiny (kSplitTinyAcc) or zero is sent to the textbook redo as well, like any row outside the window.
    // -----------------------------------------------------------------------------------------------------------------
    // the fast pass's exponent bias B = 109 - ceil(log2 n) (set_reference) and the smallest healthy row sum, 2^-(B + 2)
    const float fast_bias = 109.0f - (float)(32 - __builtin_clz((unsigned)max(n - 1, 1)));
    const float lt_floor = __builtin_amdgcn_exp2f(-(fast_bias + 2.0f));
    auto run_fast = [&]() -> bool {
        f32x16 o[QB][DB];
        float la[QB], lb[QB];   // two partial row sums per block (even / odd score registers)
        // Exponent reference of the fast pass (round 3).  It is still "nothing between the matrix core and v_exp_f32": -(m0 + B) of a
        // row sits in the accumulator every K.Q'^T product STARTS from, m0 = the row maximum over the first one or two tiles, and
        // p = 2^(s - m0 - B) with B = 109 - ceil(log2 n): every term more than 2^-(17 + log2 n) below the m0 level underflows the bf16 hi
        // AND lo terms to exact zeros (together they drop at most 2^-17 of a row's mass: two orders below this path's own error).
        // Zero operands cost the matrix core less power, and this kernel runs at the chip's power limit (c3: all-zero V -14 % time;
        // DESIGN.md 4.6).  Rows may still outgrow m0 by 2^(100 + B) before the verification below fails them.
        f32x16 minit[QB];
        float mref[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            mref[qb] = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) minit[qb][r] = 0.0f;
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            la[qb] = lb[qb] = 0.0f;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.0f;
        }
        auto store_k = [&](char* stage) {
#pragma unroll
            for (int i = 0; i < GPT; ++i) {
                if (!g_on[i]) continue;
                if constexpr (IN_BF16) {
                    *(f32x4*)(stage + g_kdst[i]) = kst[i][0];
                } else {
                    bf16x8 h8, l8;
                    if constexpr (CENTER) sub_f16x4(kst[i][0], krefp[0], krefp[1]), sub_f16x4(kst[i][1], krefp[2], krefp[3]);
                    if constexpr (GUARD) {
#pragma unroll
                        for (int e = 0; e < 4; e += 2) absmax2(kmax, kst[i][0][e], kst[i][0][e + 1]), absmax2(kmax, kst[i][1][e], kst[i][1][e + 1]);
                    }
                    split8x<QK16, false>(kst[i][0], kst[i][1], h8, l8);
                    *(bf16x8*)(stage + g_kdst[i]) = h8;
                    *(bf16x8*)(stage + C::kImageBytes + g_kdst[i]) = l8;
                }
            }
        };
        auto store_v = [&](char* stage) {
#pragma unroll
            for (int i = 0; i < GPT; ++i) {
                if (!g_on[i]) continue;
                if constexpr (IN_BF16) {
                    *(f32x4*)(stage + g_vdst[i]) = vst[i][0];
                } else {
                    bf16x8 h8, l8;
                    if constexpr (CENTER) sub_f16x4(vst[i][0], vrefp[0], vrefp[1]), sub_f16x4(vst[i][1], vrefp[2], vrefp[3]);
                    split8c(vst[i][0], vst[i][1], h8, l8);
                    *(bf16x8*)(stage + g_vdst[i]) = h8;
                    *(bf16x8*)(stage + C::kImageBytes + g_vdst[i]) = l8;
                }
            }
        };
        auto qk = [&](const char* kh_lds, f32x16 (&s)[QB]) {   // hi.hi products of every k-step first, cross terms last: see scores()
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kfh = *(const bf16x8*)(kh_lds + k_off[ks]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) s[qb] = mfma_qk<QK16>(kfh, qh[qb][ks], ks == 0 ? minit[qb] : s[qb]);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kfh = *(const bf16x8*)(kh_lds + k_off[ks]);
                if constexpr (IN_BF16) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) s[qb] = mfma_qk<false>(kfh, ql[qb][ks], s[qb]);
                } else {
                    const bf16x8 kfl = *(const bf16x8*)(kh_lds + C::kImageBytes + k_off[ks]);
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        s[qb] = mfma_qk<QK16>(kfl, qh[qb][ks], s[qb]);
                        s[qb] = mfma_qk<QK16>(kfh, ql[qb][ks], s[qb]);
                    }
                }
            }
        };
        // the reference of every row from the scores of tile 0 (in `c0`) and, when it needs no mask, tile 1 (`c1`): both were computed
        // from a zero accumulator and are shifted here, once; every later product starts from minit
        auto set_reference = [&](f32x16 (&c0)[QB], f32x16 (&c1)[QB], bool mask0, bool have1, bool use1) {
            const float bias = fast_bias;   // B = 109 - ceil(log2 n)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (mask0) {
                    asm volatile("; mask" ::: "memory");
                    const int qi = q0l + qb * 32 + lq;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = 4 * hi + (r & 3) + 8 * (r >> 2);
                        if ((key >= nk) || (CAUSAL && key > qi)) c0[qb][r] = -INFINITY;
                    }
                }
                auto lane_max = [](const f32x16& sq) {   // (compiler-visible: hipcc pads the MFMA -> VALU hazard itself)
                    float m = fmaxf(sq[0], sq[1]);
#pragma unroll
                    for (int r = 2; r < 16; r += 2) m = max3_safe(m, sq[r], sq[r + 1]);
                    return m;
                };
                float mx = lane_max(c0[qb]);
                if (use1) mx = fmaxf(mx, lane_max(c1[qb]));
                mx = xhalf_max(mx);
                mref[qb] = mx + bias;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    minit[qb][r] = -mref[qb];
                    c0[qb][r] -= mref[qb];
                    if (have1) c1[qb][r] -= mref[qb];
                }
            }
        };
        // P(j) = split(exp2(S(j))), row sums
        auto softmax = [&](f32x16 (&s)[QB], bf16x8 (&ph)[QB][2], bf16x8 (&pl)[QB][2], bool need_mask, int kv0) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (need_mask) {
                    asm volatile("; mask" ::: "memory");  // not speculatable: keeps the wave-uniform `if` a real branch
                    const int qi = q0l + qb * 32 + lq;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + 4 * hi + (r & 3) + 8 * (r >> 2);
                        if ((key >= nk) || (CAUSAL && key > qi)) s[qb][r] = -INFINITY;
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        const float p0 = fast_exp2(s[qb][8 * t + i]), p1 = fast_exp2(s[qb][8 * t + i + 1]);
                        la[qb] += p0;
                        lb[qb] += p1;
                        bf16x2 h2, l2;
                        split2c(p0, p1, h2, l2);
                        ph[qb][t][i] = h2[0], ph[qb][t][i + 1] = h2[1], pl[qb][t][i] = l2[0], pl[qb][t][i + 1] = l2[1];
                    }
            }
        };
        auto pv = [&](const bf16x8 (&ph)[QB][2], const bf16x8 (&pl)[QB][2], const char* vh_lds) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int off0 = ((4 * t + 0) * (D / 16) + 2 * db) * 128;
                    const int off1 = ((4 * t + 2) * (D / 16) + 2 * db) * 128;
                    const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vh_lds + v_lane_off + off0));
                    const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vh_lds + v_lane_off + off1));
                    const bf16x8 vfh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                    if constexpr (!IN_BF16) {
                        const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vh_lds + C::kImageBytes + v_lane_off + off0));
                        const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vh_lds + C::kImageBytes + v_lane_off + off1));
                        const bf16x8 vfl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                        for (int qb = 0; qb < QB; ++qb) o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfl, ph[qb][t], o[qb][db], 0, 0, 0);
                    }
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, pl[qb][t], o[qb][db], 0, 0, 0);
                        o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfh, ph[qb][t], o[qb][db], 0, 0, 0);
                    }
                }
        };
        auto needs_mask = [&](int kv0) { return (kv0 + kKvSplit > nk) || (CAUSAL && (kv0 + kKvSplit - 1 > q0l)); };

        if constexpr (FA_SPLIT_STAMPS != 0) st_req = __builtin_readcyclecounter();   // Q converted, references in registers (everything requested has landed)
        // ---- prologue: K(0) (requested at the top of the kernel) staged, K(1) in flight under the scores of tile 0
        store_k(smem);
        load_k(1);
        __syncthreads();
        f32x16 sa[QB], sb[QB];
        bf16x8 pha[QB][2], pla[QB][2], phb[QB][2], plb[QB][2];
        qk(smem, sa);
        store_k(smem + C::kStageBytes);
        __syncthreads();   // K(1) visible; and iteration 0 overwrites K(0) with K(2): every wave must have read its K(0) fragments first
        if constexpr (FA_SPLIT_STAMPS != 0) st_loop0 = __builtin_readcyclecounter();

        // Iteration j, tile j in stage STG = j & 1.  In: scores `cur` of tile j, P `pprev` of tile j-1.  Out: scores `next`
        // of tile j+1, P `pcur` of tile j.  FIRST has no P.V, LAST no K.Q^T; only a LAST or causal iteration can need masks.
        auto step = [&](auto stg_c, auto first_c, auto last_c, int j, f32x16 (&cur)[QB], f32x16 (&next)[QB],
                        bf16x8 (&phc)[QB][2], bf16x8 (&plc)[QB][2], bf16x8 (&php)[QB][2], bf16x8 (&plp)[QB][2]) {
            constexpr int STG = decltype(stg_c)::value;
            constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
            char* st_cur = smem + STG * C::kStageBytes;
            char* st_oth = smem + (STG ^ 1) * C::kStageBytes;
            if (!LAST) load_k(j + 2);
            load_v(j);
            const int kv0 = j * kKvSplit;
            if constexpr (FIRST || LAST) {   // executed once each: plain phases
                if (!LAST) qk(st_oth, next);                                   // K(j+1)
                if constexpr (FIRST && FA_SPLIT_REF) {
                    mfma_drain();   // the reference reads scores the matrix core may still be writing (hipcc pads its own MFMA -> VALU
                                    // hazards, but `next` is only read here when tile 1 is sampled -- keep the drain unconditional)
                    set_reference(cur, next, needs_mask(0), !LAST, !LAST && !needs_mask(kKvSplit));
                }
                softmax(cur, phc, plc, (LAST || CAUSAL) && needs_mask(kv0), kv0);
                if (!FIRST) pv(php, plp, st_oth + 2 * C::kImageBytes);          // V(j-1)
                if (!LAST) store_k(st_cur);                                     // K(j+2) over K(j)
                store_v(st_cur);                                                // V(j) over V(j-2)
            } else {
                // ---- the steady state: a static slot schedule.  Slot I issues ONE matrix instruction -- group g = I / MPG
                // is a k-step of K(j+1).Q'^T (g < KS) or a (key half, head-dim block) of V(j-1)^T.P(j-1)^T -- preceded
                // by the fragment reads of the NEXT group and followed by its share of the vector work (a unit = one
                // pair of scores: 2 exp, 2 adds, split; or four fp32 values of the K(j+2) / V(j) pieces of this thread:
                // split, and the LDS writes once a piece is complete).  sched_barrier pins the slots; inside a slot hipcc
                // orders (and pads) as it likes.
                if (CAUSAL && needs_mask(kv0)) {
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb) {
                        asm volatile("; mask" ::: "memory");
                        const int qi = q0l + qb * 32 + lq;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int key = kv0 + 4 * hi + (r & 3) + 8 * (r >> 2);
                            if (key > qi) cur[qb][r] = -INFINITY;
                        }
                    }
                }
                // Groups (round 5: the hi.hi products of ALL k-steps first, the cross terms behind them -- see scores() for why):
                //   [0, KS)          hi.hi of k-step G, one product per block                        (fragment: K_hi(G))
                //   [KS, 2 KS)       the cross terms of k-step G - KS, NPROD - 1 per block            (K_hi and, fp32 tensors, K_lo: read again)
                //   [2 KS, 2 KS + 2 DB)  a (key half, head-dim block) of V^T.P^T, NPROD per block     (V_hi, V_lo)
                // The fragments of a group are requested TWO groups ahead (three register sets): the hi.hi groups are only QB slots long.
                constexpr int NQ1 = QB, NQ2 = (NPROD - 1) * QB, NPV = NPROD * QB;
                constexpr int NG = 2 * KS + 2 * DB, NSLOT = KS * (NQ1 + NQ2) + 2 * DB * NPV;
                constexpr auto group_start = [](int G) constexpr {
                    return G < KS ? G * NQ1 : G < 2 * KS ? KS * NQ1 + (G - KS) * NQ2 : KS * (NQ1 + NQ2) + (G - 2 * KS) * NPV;
                };
                constexpr auto group_of = [group_start](int I) constexpr {
                    int G = 0;
                    while (G + 1 < NG && group_start(G + 1) <= I) ++G;
                    return G;
                };
                constexpr int NU_S = 8 * QB;                                     // score pairs
                constexpr int NU_C = IN_BF16 ? 2 * GPT : 8 * GPT;               // K/V pieces: plain stores, or half units of the split
                constexpr int NU = 2 * NU_S + NU_C;                              // half units
                bf16x8 fh[3], fl[3];       // fragments of the current group and the two behind it
                bf16x8 ch[2][GPT], cl[2][GPT];   // converted pieces (K, V) being assembled
                const char* k_img = st_oth;                          // K(j+1) hi (lo at + kImageBytes)
                const char* v_img = st_oth + 2 * C::kImageBytes;     // V(j-1) hi
                auto load_frags = [&](auto gc) {
                    constexpr int G = decltype(gc)::value;
                    if constexpr (G < KS) {
                        fh[G % 3] = *(const bf16x8*)(k_img + k_off[G]);
                    } else if constexpr (G < 2 * KS) {
                        fh[G % 3] = *(const bf16x8*)(k_img + k_off[G - KS]);
                        if constexpr (!IN_BF16) fl[G % 3] = *(const bf16x8*)(k_img + C::kImageBytes + k_off[G - KS]);
                    } else {
                        constexpr int t = (G - 2 * KS) / DB, db = (G - 2 * KS) % DB;
                        constexpr int off0 = ((4 * t + 0) * (D / 16) + 2 * db) * 128;
                        constexpr int off1 = ((4 * t + 2) * (D / 16) + 2 * db) * 128;
                        const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_img + v_lane_off + off0));
                        const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_img + v_lane_off + off1));
                        fh[G % 3] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
                        if constexpr (!IN_BF16) {
                            const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_img + C::kImageBytes + v_lane_off + off0));
                            const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_img + C::kImageBytes + v_lane_off + off1));
                            fl[G % 3] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                        }
                    }
                };
                // half units: (2u) = exp + row sums of a score pair / split of the first two values of a piece,
                //             (2u+1) = hi/lo split of that pair / split of the other two values (+ LDS writes of a whole piece)
                auto unit = [&](auto uc) {
                    constexpr int X = decltype(uc)::value, U = X / 2, H = X % 2;
                    if constexpr (X >= 2 * NU_S && IN_BF16) {   // bf16 tensors: a piece is stored as it came
                        constexpr int c = X - 2 * NU_S, gi = c / 2, which = c % 2;
                        *(f32x4*)(st_cur + (which ? g_vdst[gi] : g_kdst[gi])) = which ? vst[gi][0] : kst[gi][0];
                    } else if constexpr (U < NU_S) {
                        constexpr int qb = U / 8, t = (U % 8) / 4, i = 2 * (U % 4);
                        if constexpr (H == 0) {
                            const float p0 = fast_exp2(cur[qb][8 * t + i]), p1 = fast_exp2(cur[qb][8 * t + i + 1]);
                            la[qb] += p0;
                            lb[qb] += p1;
                            cur[qb][8 * t + i] = p0;        // the scores are dead: keep P in their registers until the split
                            cur[qb][8 * t + i + 1] = p1;
                        } else {
                            bf16x2 h2, l2;
                            split2c(cur[qb][8 * t + i], cur[qb][8 * t + i + 1], h2, l2);
                            phc[qb][t][i] = h2[0], phc[qb][t][i + 1] = h2[1], plc[qb][t][i] = l2[0], plc[qb][t][i + 1] = l2[1];
                        }
                    } else {
                        constexpr int c = U - NU_S, gi = c / 4, which = (c % 4) / 2, half = c % 2;
                        f32x4 x = which ? vst[gi][half] : kst[gi][half];
                        if constexpr (CENTER && which == 0) sub_f16x4(x, krefp[2 * half], krefp[2 * half + 1]);
                        if constexpr (CENTER && which == 1) sub_f16x4(x, vrefp[2 * half], vrefp[2 * half + 1]);
                        if constexpr (GUARD && which == 0) absmax2(kmax, x[2 * H], x[2 * H + 1]);
                        bf16x2 h2, l2;
                        split2x<QK16 && which == 0, false>(x[2 * H], x[2 * H + 1], h2, l2);   // K pieces: fp16 terms (fp32 tensors)
                        ch[which][gi][4 * half + 2 * H] = h2[0], ch[which][gi][4 * half + 2 * H + 1] = h2[1];
                        cl[which][gi][4 * half + 2 * H] = l2[0], cl[which][gi][4 * half + 2 * H + 1] = l2[1];
                        if constexpr (half == 1 && H == 1) {   // piece complete: K(j+2) over K(j), V(j) over V(j-2), both in this tile's stage
                            char* dst = st_cur + (which ? g_vdst[gi] : g_kdst[gi]);
                            *(bf16x8*)dst = ch[which][gi];
                            *(bf16x8*)(dst + C::kImageBytes) = cl[which][gi];
                        }
                    }
                };
                load_frags(std::integral_constant<int, 0>{});
                load_frags(std::integral_constant<int, 1>{});
                auto slot = [&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    constexpr int G = group_of(I), M = I - group_start(G);
                    if constexpr (M == 0 && G + 2 < NG) load_frags(std::integral_constant<int, G + 2>{});
                    if constexpr (G < KS) {                    // hi.hi of k-step G; the first one starts from -(m0 + B)
                        constexpr int qb = M;
                        if constexpr (G == 0) next[qb] = mfma_qk<QK16>(fh[G % 3], qh[qb][G], minit[qb]);
                        else next[qb] = mfma_qk<QK16>(fh[G % 3], qh[qb][G], next[qb]);
                    } else if constexpr (G < 2 * KS) {         // cross terms -- fp32 tensors: K_lo.Q'_hi, K_hi.Q'_lo; bf16 tensors: K.Q'_lo
                        constexpr int ks = G - KS, term = M / QB, qb = M % QB;
                        constexpr bool a_lo = !IN_BF16 && term == 0;
                        const bf16x8& a = a_lo ? fl[G % 3] : fh[G % 3];
                        const bf16x8& bq = a_lo ? qh[qb][ks] : ql[qb][ks];
                        next[qb] = mfma_qk<QK16>(a, bq, next[qb]);
                    } else {                                   // fp32 tensors: lo.hi, hi.lo, hi.hi; bf16 tensors (V exact in one term): hi.lo, hi.hi
                        constexpr int t = (G - 2 * KS) / DB, db = (G - 2 * KS) % DB, term = M / QB, qb = M % QB;
                        constexpr bool a_lo = !IN_BF16 && term == 0, b_lo = IN_BF16 ? term == 0 : term == 1;
                        const bf16x8& a = a_lo ? fl[G % 3] : fh[G % 3];
                        const bf16x8& bp = b_lo ? plp[qb][t] : php[qb][t];
                        o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bp, o[qb][db], 0, 0, 0);
                    }
                    // units whose slot this is: unit u sits in slot floor(u * NSLOT / NU)
                    constexpr int u_lo = (I * NU + NSLOT - 1) / NSLOT, u_hi = ((I + 1) * NU + NSLOT - 1) / NSLOT;
                    for_each_index([&](auto k) { unit(std::integral_constant<int, u_lo + decltype(k)::value>{}); },
                                   std::make_integer_sequence<int, (u_hi > u_lo ? u_hi - u_lo : 0)>{});
                    __builtin_amdgcn_sched_barrier(0);
                };
                for_each_index(slot, std::make_integer_sequence<int, NSLOT>{});
            }
            __syncthreads();
        };
        constexpr std::integral_constant<int, 0> S0{};
        constexpr std::integral_constant<int, 1> S1{};
        constexpr std::true_type YES{};
        constexpr std::false_type NO{};
        if (nt == 1) {
            step(S0, YES, YES, 0, sa, sb, pha, pla, phb, plb);
            pv(pha, pla, smem + 2 * C::kImageBytes);
        } else {
            step(S0, YES, NO, 0, sa, sb, pha, pla, phb, plb);       // P(0) -> a
            int j = 1;
            for (; j + 2 < nt; j += 2) {                          // j odd here
                step(S1, NO, NO, j, sb, sa, phb, plb, pha, pla);     // P(j) -> b, consumes a
                step(S0, NO, NO, j + 1, sa, sb, pha, pla, phb, plb); // P(j+1) -> a, consumes b
            }
            if (nt - j == 2) {
                step(S1, NO, NO, j, sb, sa, phb, plb, pha, pla);
                step(S0, NO, YES, j + 1, sa, sb, pha, pla, phb, plb);
                pv(pha, pla, smem + 2 * C::kImageBytes);           // V(nt-1), nt-1 even: stage 0
            } else {
                step(S1, NO, YES, j, sb, sa, phb, plb, pha, pla);
                pv(phb, plb, smem + C::kStageBytes + 2 * C::kImageBytes);   // nt-1 odd: stage 1
            }
        }

        // ================= epilogue: O / l, store =================
        mfma_drain();  // the last P.V MFMAs may still be in flight
        if constexpr (FA_SPLIT_STAMPS != 0) st_loop1 = __builtin_readcyclecounter();
        float crows[QB];
        for_each_index([&](auto qbc) { crows[decltype(qbc)::value] = crow_of(qbc); }, std::make_integer_sequence<int, QB>{});
        bool ok = true;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const float lt = xhalf_sum(la[qb] + lb[qb]);
            const float inv = 1.0f / lt;
            const int qi = q0 + qb * 32 + lq;
            const float crow = crows[qb];
            float mag = 0.0f;
            if (qi < n) {
                const int64_t o_off = o_slab + (int64_t)qi * p.o_row_stride + 4 * hi;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            pk[e] = o[qb][db][4 * g + e] * inv;
                            mag += fabsf(pk[e]);
                        }
                        if constexpr (CENTER) pk += *(const f32x4*)&s_vref[db * 32 + 8 * g + 4 * hi];   // value centering: the reference row back in
                        store4(p, o_off + db * 32 + 8 * g, pk);
                    }
                if (FA_SPLIT_STAMPS == 0 && p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = ((mref[qb] + __builtin_amdgcn_logf(lt)) + crow) * kLn2;
                // lt: the row's own reference term 2^-B is in the sum, so a healthy row sum never falls below 2^-(B + 1) (B = 101 for n <= 256:
                // round 3 tested against a fixed 2^-100 there and redid every tile whose reference key held most of a row's mass);
                // mag * lt = the sum of the unnormalised accumulators: tiny or zero means the products p v of the terms that matter
                // were near (or below) fp32's subnormal range (|v| below ~2^-30): the textbook redo (p <= 1) takes those -- an all-zero V too
                const bool in_range = (lt > lt_floor) && (lt < kSplitLimit) && (mag < INFINITY);   // false for NaN as well
                const float acc = mag * lt;
                ok = ok && in_range && !(acc < kSplitTinyAcc);
                if constexpr (CENTER) hard_fail = hard_fail || !in_range || (acc != 0.0f && acc < kSplitTinyAcc);
                if (GUARD) saw_nan = saw_nan || (lt != lt);   // (a NaN row SUM: NaN logits -- an overflowed window gives +inf, which the redo handles)
            }
        }
        return ok;
    };

    // results are stored before the vote (a rejected tile is simply overwritten by the redo)
    bool ok;
    if constexpr (PIPE) ok = run_fast();
    else ok = run_tile(std::true_type{});
    if constexpr (GUARD) {   // every thread converted its share of every K tile: fold the shares into the workgroup's maximum
        if (p.flag_mode >= 3) atomicMax(&s_kmax, __float_as_uint(kmax));   // non-negative floats order like their bit patterns
    }
    bool redo = __syncthreads_or(!ok);
    if constexpr (CENTER && PIPE) {
        // Accumulators that are exactly zero: either every centred value the rows saw is exactly zero -- a V that is constant over the share
        // at a value fp16 holds (zeros, ones: padding heads, sanity checks), and then vbar + 0, already stored, IS the result -- or products
        // that underflowed as a whole (|v - vbar| below ~2^-53), which need the redo.  The rare path can afford to look: one pass over the
        // share's V (L2 hits, a tenth of a tile's time) instead of recomputing the tile -- until this, a constant V cost fp32 tensors twice
        // the time (profiles/r05_redo_rate.txt).
        if (redo && !__syncthreads_or(hard_fail)) {
            const float* vf = (const float*)vg;
            int any = 0;   // (bitwise, no short circuit: the loads of an unrolled group are in flight together)
#pragma unroll 4
            for (int i = tid; i < kv_end * (D / 4); i += NWAVES * kWave) {
                const int row = i / (D / 4), c4 = (i % (D / 4)) * 4;
                const f32x4 x = *(const f32x4*)(vf + (int64_t)row * p.kv_row_stride + c4);
                const f32x4 r = *(const f32x4*)&s_vref[c4];
                any |= (int)(x[0] != r[0]) | (int)(x[1] != r[1]) | (int)(x[2] != r[2]) | (int)(x[3] != r[3]);   // (a NaN differs from everything)
            }
            if (!__syncthreads_or(any)) redo = false;
        }
    }
    if constexpr (GUARD) {
        // |q'|_2 carries scale * log2(e); +-inf on either side fails the comparison, a NaN is caught through the first attempt's
        // results (saw_nan): either way the output is then produced in fp32 arithmetic --
        //   flag_mode 4 (FA_KERNEL_AUTO, round 4): by THIS workgroup, for its own rows, right here (fa_f32_exact.h: the body of the exact
        //               kernel over the same keys, in the LDS this kernel is done with); the word, if the caller has one, only reports it;
        //   flag_mode 3 (the ablation library's chains): by the exact kernel queued behind this launch, for the whole grid.
        bool wide;
        if constexpr (QK16) {
            // fp16 terms: what the guard bounds is the RANGE of fp16, not the logit width (3 * 2^-22 * sum |q'_i k_i| is below the rounding
            // bound of the fp32 FMA chain itself, d * 2^-24 * sum |q'_i k_i|, at every width).  (1) |x| >= 65520 -> hi = inf, lo = -inf ->
            // NaN scores -> saw_nan (as for NaN / inf inputs).  (2) an element below 2^-3 has a subnormal lo term, absolute error <= 2^-25,
            // multiplied by its partner: sum <= 2^-25 (|k|_1 + |q'|_1) <= 2^-25 (D |k|_inf + sqrt(D) |q'|_2) per logit; kept <= 2^-12.
            wide = p.flag_mode >= 3 && (saw_nan || !((float)D * __uint_as_float(s_kmax) + sqrtf((float)D * qn2) <= kSubnormalBudget));
        } else {
            wide = p.flag_mode >= 3 && (saw_nan || !(sqrtf(qn2) * __uint_as_float(s_kmax) <= kGuardLimit * kLog2e));
        }
        if (p.flag_mode == 3) {
            if (wide) __hip_atomic_store(p.flag, p.flag_serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (p.flag_mode == 4) {   // (uniform)
            if (__syncthreads_or(wide)) {   // ... which is also the barrier behind the last LDS read of the attempt
                if (p.flag != nullptr && tid == 0) __hip_atomic_store(p.flag, p.flag_serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
                for (int qb = 0; qb < QB; ++qb) {
                    if (qb > 0) __syncthreads();   // every wave is out of the previous block's last tile
                    f32_exact_rows<D, NWAVES, CAUSAL, false>(p, smem, qg, kg, vg, o_slab, slab, q0 + qb * 32, kbeg, nk, kv_end, wave, lane);
                }
                count_cliff(p, 1);
                return;
            }
        }
    }
    if (redo) {
        run_tile(std::false_type{});
        count_cliff(p, 0);   // (behind the redo: nothing is live here -- in front of it the causal 128-row tilings spilled 68 bytes)
    }
    if constexpr (FA_SPLIT_STAMPS != 0 && PIPE) {   // fa_driver_ablation --mode prof4 --dtype f32: the layout of the bf16 kernels' stamps
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_out = __builtin_readcyclecounter();
        if (lane == 0 && p.lse != nullptr) {
            float* dst = p.lse + ((int64_t)blockIdx.x * NWAVES + wave) * 8;
            dst[0] = (float)(st_loop1 - st_loop0);   // the tiles of the pipelined pass
            dst[1] = 0.0f;
            dst[2] = (float)nt;
            dst[3] = (float)(st_out - st_in);
            dst[4] = (float)(st_req - st_in);        // references, Q', K(0) requested and landed; Q' converted
            dst[5] = (float)(st_loop0 - st_req);     // K(0), K(1) converted and staged, scores of tile 0
            dst[6] = 0.0f;
            dst[7] = (float)(st_out - st_loop1);     // O / l, stores, verification votes, range guard
        }
    }
}

template <int D, int NWAVES, int QB, int MINBLOCKS, bool PIPE, bool IN_BF16 = false>
static hipError_t launch_split(const FwdParams& p0, int causal, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = NWAVES * QB * 32;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)total), block(NWAVES * kWave);
    // The one-block-per-wave tilings at d <= 64 run two workgroups per CU.  A causal grid that is resident as a whole (at most two
    // workgroups per CU) then lasts as long as its heaviest PAIR of tiles: deal them heavy + light (causal_tile, fa_common.h).
    // Measured, fp32 tensors, ms: 16 x 4096 d = 64 0.162 -> 0.125, 32 x 2048 0.089 -> 0.074, 16 x 4096 d = 32 0.122 -> 0.094, and with
    // slabs cut by XCD boundaries 12 x 4096 0.149 -> 0.134, 20 x 2048 0.079 -> 0.071, 12 x 4096 d = 32 0.119 -> 0.104; the
    // two-block tilings (one workgroup per CU, heavy tiles first) must keep their order: 16 x 8192 0.397 -> 0.531 when paired.
    // (larger grids: +-3 % either way on seven shapes -- they keep the plain order)
    p.alt_order = (causal && QB == 1 && NWAVES == 4 && D <= 64 && total <= 2 * 256) ? 1 : 0;
    if (causal)
        hipLaunchKernelGGL((fa_fwd_f32_split_kernel<D, NWAVES, QB, true, MINBLOCKS, PIPE, IN_BF16>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_f32_split_kernel<D, NWAVES, QB, false, MINBLOCKS, PIPE, IN_BF16>), grid, block, 0, stream, p);
    return hipGetLastError();
}

// mode 0 = the product choice; 5 (d = 128) = mode 1 with eight waves per workgroup; 1 / 2 = first-tile-reference pass with one / two 32-row blocks per wave (tiles above a causal
// wave's diagonal are skipped), 3 / 4 = software-pipelined reference-free pass with one / two blocks per wave (slot-pinned
// steady state; causal tiles above the diagonal are masked, not skipped).  Measured on MI355X (one box, interleaved, ms):
//   d=64  BH=16  N=8192   non-causal m1 0.757  m3 0.706  m4 0.655 | causal m1 0.473  m3 0.410  m4 0.376   (exact: 2.06 | 1.66)
//   d=64  BH=128 N=1024   non-causal m1 0.107  m3 0.100  m4 0.100 | causal m1 0.084  m3 0.085  m4 0.095   (0.265 | 0.287)
//   d=64  BH=32  N=4096   causal m1 0.265  m3 0.235  m4 0.283;  BH=32 N=2048 causal m1 0.095  m3 0.083  m4 0.085
//   d=64  BH=8   N=4096   non-causal m1 0.123  m3 0.103  m4 0.136      (128 tiles of 256 rows: half the CUs idle with m4)
//   d=128 BH=16  N=8192   non-causal m1 1.775  m3 1.942  m5 1.300 (pipelined 8-wave: 1.790) | causal m1 1.080  m3 0.796  m5 0.702   (4.03 | 2.47)
//   d=128 BH=128 N=1024   non-causal m1 0.253  m3 0.244  m5 0.186 | causal m1 0.243  m3 0.168  m5 0.157
//   (8-wave workgroups at d=64 / d=32 -- tried as m5..m7 -- lose to m4: 0.79 / 0.70 / 0.84 vs 0.66 at BH=16 N=8192 d=64)
//   d=32  BH=16  N=8192   non-causal m2 0.476  m3 0.474  m4 0.450 | causal m1 0.328  m3 0.295  m4 0.271   (1.14 | 0.86)
static bool split_addressable(const FwdParams& p, int d, unsigned elem_size)
{
    return ((uint64_t)(p.n - 1) * (uint64_t)p.kv_row_stride + (uint64_t)d) * elem_size < (1ull << 32);
}

// Short rows (round 3, profiles/r03_short_rows.txt; ms at BH x N = 131072 rows, m1 / m3 / m4-or-m5): the pipelined pass fills and
// drains over four tiles and wants whole 256-row workgroups --
//   d=64  non-causal  N=128 0.034 / 0.052 / 0.092   256 0.047 / 0.076 / 0.088   512 0.070 / 0.075 / 0.094   640 0.081 / 0.071 / 0.095
//                     896 0.104 / 0.094 / 0.121     1024 0.115 / 0.102 / 0.104  (16 x 256: 0.013 / 0.021 / 0.036, 16 x 512: 0.022 / 0.016 / 0.025)
//         causal      N=512 0.059 / 0.064 / 0.114   768 0.075 / 0.067 / 0.084   1024 0.095 / 0.086 / 0.100
//   d=32  non-causal  N=128 0.022 / 0.035 / 0.065   256 0.031 / 0.053 / 0.059   384 0.042 / 0.039 / 0.049   512 0.050 / 0.047 / 0.046
//         causal      N=256 0.028 / 0.052 / 0.070   512 0.044 / 0.042 / 0.070   768 0.058 / 0.052 / 0.062
//   d=128 non-causal  N=128 0.072 / 0.117 / 0.081   640 0.176 / 0.146 / 0.171   896 0.223 / 0.184 / 0.227 (m5's 256-row tiles: 1024 rows of work)
static int choose_split(const FwdParams& p, int d, int causal, unsigned elem_size)
{
    // the pipelined pass addresses K/V through 32-bit buffer offsets
    const bool addressable = split_addressable(p, d, elem_size);
    if (!addressable) return 1;
    const int64_t tiles256 = (int64_t)p.bh * ((p.n + 255) / 256);
    const int64_t tiles128 = (int64_t)p.bh * ((p.n + 127) / 128);
    // 256-row workgroups compute whole tiles: rows past N are wasted, and with N mod 256 in (0, 128] the 128-row tiling wastes a tile less
    const bool fits256 = p.n % 256 == 0 || p.n % 256 > 128 || p.n >= 4096;
    // 256-row workgroups run one per CU: a round that is filled to between a quarter and three quarters costs a whole one (32 x 3072,
    // 384 tiles: m3 0.213 / m4 0.239 at d = 64, 0.161 / 0.180 at d = 32)
    const int64_t part = tiles256 % 256;
    const bool rounds256 = tiles256 >= 1024 || part == 0 || part >= 192;
    // Round 5 (fp16 terms, hi.hi first, centred keys; profiles/r05_sweep_f32short.txt, 131072 rows per launch, ms m1 / m3 / m4-or-m5): the
    // phase-structured pass lost its edge on short rows of BIG grids -- d=64 N=128 0.0355 / 0.0322, 256 0.0467 / 0.0422, 512 0.0692 / 0.0644
    // (causal 512 0.0614 / 0.0555); d=128 N=128 0.0740 / 0.0672, N=384 0.1264 / 0.1101 / 0.1260 (round 3: m5 0.124, m3 0.161) --, so from two
    // full rounds of 128-row workgroups on those go to the pipelined 128-row tiling; small grids keep the pass without a pipeline to fill.
    const bool big = tiles128 >= 1024;
    if (d == 128) {
        if (p.n <= 128) return big ? 3 : 1;
        // two blocks per wave do not fit the register file; EIGHT waves of one block each (256-row workgroups, two waves per
        // SIMD, phases in sequence) halve the K/V conversion work and the L2 traffic per row
        // causal, ms m3 / m5: 64 x 2048 0.257 / 0.291, 32 x 4096 0.412 / 0.511, 8 x 8192 0.387 / 0.560; 16 x 8192 0.796 / 0.702, 128 x 1024 0.168 / 0.157
        if (causal && p.n > 1536 && (p.n < 8192 || tiles256 < 512)) return 3;
        return (tiles256 >= 256 && fits256) ? 5 : 3;   // small grids (BH=4 N=4096: m1 0.231, m3 0.158, m5 0.274 ms): 128-row workgroups, pipelined
    }
    // rows of a few tiles: the first-tile-reference pass (no pipeline to fill); at d = 64 up to 512 keys once the grid is two rounds deep
    if (d == 64 && big) {
        if (causal && p.n <= 384) return 1;                                // (ties with the pipelined tiling: 0.0499 / 0.0482 at N = 384)
    } else if (p.n <= 256) {
        return 1;
    }
    if (causal) {
        if (p.n <= (d == 64 ? (big ? 384 : 512) : 384)) return 1;         // short rows: skipping tiles beats masking them
        // 256-row tiles only pay on long rows, and from two rounds on: one round of them lasts as long as its heaviest tile, alone on its
        // CU (8 x 8192: m3 0.220 / m4 0.309 at d = 64, 0.173 / 0.238 at d = 32; 16 x 8192: 0.410 / 0.376)
        return (p.n >= 8192 && tiles256 >= 512) ? 4 : 3;
    }
    if (d == 64 && p.n <= 1024) return 3;
    return (tiles256 >= 256 && fits256 && rounds256) ? 4 : 3;              // small grids: 128-row workgroups fill more CUs
}

// every tiling instantiated for head dim D (one translation unit per (dtype, D): fa_split_{f32,bf16}_d{32,64,128}.hip, so the
// instantiations compile in parallel)
template <int D, bool IN_BF16>
static hipError_t launch_split_modes(const FwdParams& p, int causal, int mode, hipStream_t stream)
{
    if constexpr (D == 128) {
        if (mode == 1) return launch_split<128, 4, 1, 1, false, IN_BF16>(p, causal, stream);
        if (mode == 3) return launch_split<128, 4, 1, 1, true, IN_BF16>(p, causal, stream);
        if (mode == 5) return launch_split<128, 8, 1, 1, false, IN_BF16>(p, causal, stream);   // 8 waves: 256-row workgroups
    } else {
        if (mode == 1) return launch_split<D, 4, 1, 2, false, IN_BF16>(p, causal, stream);
#if FA_ABLATION
        if constexpr (!IN_BF16)
            if (mode == 2) return launch_split<D, 4, 2, 1, false, false>(p, causal, stream);   // two blocks per wave, not pipelined: superseded by mode 4
#endif
        if (mode == 3) return launch_split<D, 4, 1, 2, true, IN_BF16>(p, causal, stream);
        if (mode == 4) return launch_split<D, 4, 2, 1, true, IN_BF16>(p, causal, stream);
    }
    return hipErrorInvalidValue;
}

// defined in fa_split_{f32,bf16}_d{32,64,128}.hip
hipError_t split_launch_f32_d32(const FwdParams& p, int causal, int mode, hipStream_t stream);
hipError_t split_launch_f32_d64(const FwdParams& p, int causal, int mode, hipStream_t stream);
hipError_t split_launch_f32_d128(const FwdParams& p, int causal, int mode, hipStream_t stream);
hipError_t split_launch_bf16_d32(const FwdParams& p, int causal, int mode, hipStream_t stream);
hipError_t split_launch_bf16_d64(const FwdParams& p, int causal, int mode, hipStream_t stream);
hipError_t split_launch_bf16_d128(const FwdParams& p, int causal, int mode, hipStream_t stream);

}  // namespace fa
