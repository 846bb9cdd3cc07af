// fa_bf16_xn_kernel.h -- the one-wave-per-SIMD bf16 kernel with explicit register files, templated on the head dimension D and on
// the number NB of 32-row blocks a wave owns.  ONE source for what used to be two near-copies:
//     NB = 4, D = 64          "x4": 128 query rows per wave, 512-row workgroups   (fa_fwd_bf16_x4*.hip: large non-causal grids, c4 / c5)
//     NB = 2, D = 32/64/128   "x2":  64 query rows per wave, 256-row workgroups   (fa_fwd_bf16_x2.hip: d = 128, d = 32, causal and small grids)
// and, through the PF switch, the fp16-P forms: one fp16 term of P (PF = 1: fa_fwd_bf16_x4_p16.hip, fa_fwd_bf16_x2_p16_d*.hip; explicit
// choice only) and two (PF = 2: fa_fwd_bf16_x2_p16x2_d*.hip -- the accurate path FA_KERNEL_AUTO gives an fp32 output).
// Replaces the hot loop of flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:214-354, :434,480-484).
//
// Why one wave per SIMD: the loop is bound by the number of instructions issued per MFMA AND by the chip's power budget (a second
// resident wave does add issue capacity -- 1330 cycles per step for each of two NB = 2 workgroups against 1057 for one alone -- but on
// a fully loaded chip the clock gives most of it back: DESIGN.md 4.6), and both budgets are spent best by the stream with the
// fewest instructions per FLOP.  More blocks per wave share every K / V^T fragment read, barrier, DMA and piece of
// per-step bookkeeping among more FLOP; the single resident wave needs every latency hidden by the software pipeline itself:
//
//   step t (32 keys), NB = 4:   K.Q^T of sub-tile t+1 for blocks A,B | P.V + row sums of A | K.Q^T (t+1) for C,D, first half
//                               | P.V B | K.Q^T C,D second half | P.V C | P.V D                                  (40 MFMA slots)
//   step t (32 keys), NB = 2:   K.Q^T of sub-tile t+1 for A and B (2 KS slots) | P.V + row sums of A (NV + 2) | P.V + row sums of B
//   VALU work (exp + pack of every block for sub-tile t; rescaled mixes: the lane maxima of sub-tile t+1 and the rescale test) is
//   cut into ~4-instruction units and dealt out over the MFMA slots by measured issue cost, which by construction finishes exp(X)
//   before P.V(X) and starts max(X) only after K.Q^T(X) has retired (xn_schedule_ok checks every dependency of every instantiated
//   schedule at compile time; profiles/r03_xn_schedule_check.py is the offline model used to search the dealing limits).  V^T fragments of the step are read in its first NV slots, the K fragments of the next step
//   right after the step's last K.Q^T slot.
//
// D = 128 (NB = 2): a 32x32-key block has 18 MFMAs for the same 40 (48) VALU instructions as at D = 64 -- the step is bound by the
//   matrix pipe, not by instruction issue (1254 TFLOP/s at BH = 16, N = 8192; registers: O 128 + Q 64 + row sums in AGPRs).
// D = 64, NB = 2: fewer rows per wave than NB = 4 (more LDS reads and bookkeeping per FLOP), but 256-row tiles that run as two
//   rounds of workgroups per CU, heavy tiles first: the causal case is no longer bound by its heaviest tile.
#pragma once
#include <utility>
#include "fa_bf16_step.h"
#include "fa_kernels.h"

#ifndef FA_OPT_RECENTRE
#define FA_OPT_RECENTRE 1   // 0: experiment switch -- the optimistic mix keeps its sampled exponent reference for the whole tile (rounds 2-5)
#endif
#ifndef FA_XN_TAKE_TURNS
#define FA_XN_TAKE_TURNS 1   // 0: experiment switch -- two workgroups sharing a CU leave the issue priority to the arbiter (oldest wave first)
#endif
#ifndef FA_OPT_SAMPLE_SUBTILES
#define FA_OPT_SAMPLE_SUBTILES 99   // experiment switch: at most this many sub-tiles BEYOND the first go into the sampled reference (0: none)
#endif
#ifndef FA_OPT_SAMPLE_WHEN_RECENTRED
#define FA_OPT_SAMPLE_WHEN_RECENTRED 0   // 1: experiment switch -- tiles that re-centre their reference sample it in the prologue as well (round 6 until its last day)
#endif
#ifndef FA_OPT_RC_FIRST
#define FA_OPT_RC_FIRST 8           // experiment switch: stages into a long tile at which the reference is re-centred for the first time (then x 3)
#endif
#ifndef FA_OPT_SAMPLE
#define FA_OPT_SAMPLE 1   // 0: experiment switch -- the optimistic mix takes its exponent reference from the first sub-tile only (round 2)
#endif
namespace fa {
// ---- the two-term-P (PF = 3) step schedules: what round 4 measured its way to (profiles/r04_experiments.txt, r04_pb2_ab.txt) ----------
// A v_dot2c waits until the MFMA issued before it has LEFT the matrix pipe -- ~50 cycles after that MFMA's issue
// (profiles/r04_ubench_dot2b.txt: the first dot behind an MFMA costs 23 cycles, every further one 4) -- so:
//   * a slot's dots are emitted behind its other VALU work (pb2_dots_last; D = 32 has 42 units in 20 slots -- no dealing keeps every
//     pack out of its dots' slot -- and keeps list order);
//   * a block's sixteen dots are ONE unit, behind the second fragment's pack, and the block's P.V group runs (hi, hi, lo, lo) so that
//     the lo fragments may complete later: every group of dots pays the wait at most once (c4: 4 dots per unit 0.3755 ms, 8: 0.3689;
//     another box 8: 0.3578, 16: 0.3550);
//   * a slot that holds dots takes kDotSlotExtra more non-dot units from the slot behind it (c4 +0.5 %, through NB = 2 +2.6 %);
//   * issuing the MFMA of a dot slot a slot early (70 cycles of VALU between it and the dots) bought nothing: the wave then waits at the
//     second of the two back-to-back MFMAs instead ("a wave waiting to issue an MFMA holds the vector issue port").
constexpr bool pb2_dots_last(int D) { return D >= 64; }
constexpr int kDotSlotExtra = 1;
// workgroups of the NB = 2 two-term kernel a CU is asked to hold (register budget 512 / this per lane).  Asked for two at d = 64, hipcc
// spills: 17 registers to scratch in the causal instantiation, ~20 VGPRs into AGPRs inside the loop in the non-causal one (659
// v_accvgpr_read in the binary); asked for one they come out at 252 - 256 registers without either and fit a CU twice anyway -- like the
// bf16-P NB = 2 kernel, see xn_launch_order; tests/test_code_objects.py pins that on the binary.
constexpr int pb2_wgs_per_cu(int D) { return D == 32 ? 2 : 1; }
}  // namespace fa

namespace fa {

// PF: the format of P (and of the V image) in the second contraction -- 0 = bf16, 1 = fp16 (11 significant bits), 2 = fp16 hi + fp16 lo
// (two products per P.V and per row sum: P to ~22 bits; see XSoft), 3 = bf16 hi + bf16 lo (round 4: P to ~17 bits with bf16's exponent
// range, so V stays as it is -- no copy, no scratch, no launch chain -- and the optimistic mix applies)
template <int D, int NB, int PF = 0>
struct XShape {
    static_assert((NB == 4 && D == 64) || (NB == 2 && (D == 32 || D == 64 || D == 128)), "instantiated shapes");
    static_assert(PF >= 0 && PF <= 3, "P formats");
    static constexpr int KS = D / 16;          // k-steps of K.Q^T
    static constexpr int DB = D / 32;          // 32-column blocks of O
    static constexpr int NV = 2 * DB;          // V^T fragments per 32-key sub-tile
    static constexpr int NT = pf_terms(PF);    // terms of P
    static constexpr int GRP = NT * (NV + 2);  // slots of one P.V + row-sum group
    static constexpr int kSlots = NB * (KS + GRP);
    static constexpr int kFirstPv = 2 * KS;    // first slot that needs the V^T fragments (both layouts start with K.Q^T of A, B)
    // NB = 4: the V^T fragments are waited for one by one in the P.V slots of block A that consume them (the youngest read gets
    // nine slots instead of five to land); NB = 2: one wait in front of the first P.V slot
    static constexpr bool kVWaitPerFrag = NB == 4;
    // K fragments of the next step: KS slots starting here (after the step's last K.Q^T slot)
    static constexpr int kKLoad = NB == 4 ? 16 + 2 * GRP : 2 * KS + 1;
    // VALU units per block: 16 exponentials + 2 packs; the lo term of P adds 4 half fragments (PF = 2: one v_fma_mix per element packs
    // as it goes) or 4 + 4 (PF = 3: the dots of a half fragment and its packs are separate units, see lo_dots_bf16); rescaled mixes:
    // 3 lane-max micro-steps per block and the test
    static constexpr int kLoUnits = PF == 2 ? 4 : PF == 3 ? 3 : 0;
    static constexpr int kUnitsOpt = NB * (18 + kLoUnits), kUnitsRsc = NB * (21 + kLoUnits) + 1;
    // VALU units are dealt out over the first kWend half-slots of the step (largest values that put every pack in front of the first MFMA
    // reading it, found offline -- profiles/r04_xn_schedule_check.py -- and re-checked at compile time: xn_schedule_ok); NB = 4 with one
    // term of P: see weight_end()
    static constexpr int kWendOpt2 = PF == 3 ? (D == 128 ? 89 : D == 64 ? 47 : 25) : (D == 128 ? 58 : D == 64 ? 30 : 16);
    static constexpr int kWendRsc2 = PF == 3   ? (D == 128 ? 92 : D == 64 ? 49 : 27)
                                     : PF == 2 ? (D == 128 ? 99 : D == 64 ? 52 : 29)
                                               : (D == 128 ? 64 : D == 64 ? 34 : 18);
    static constexpr int kWendOpt4 = 96, kWendRsc4 = 107;   // NB = 4, PF = 3
};
// ---- matrix instructions with explicit register files ------------------------------------------------------------------
// With one wave per SIMD the wave owns 256 architectural VGPRs and 256 accumulation registers (AGPRs).  VALU instructions
// only reach the former, MFMA operands may sit in either.  hipcc picks ONE form for every MFMA of a function (all
// accumulators in AGPRs: the scores then need a v_accvgpr_read per element; or all in VGPRs: the Q fragments and
// output accumulators are shuttled through v_accvgpr copies) -- either way hundreds of extra issue slots per step.  So
// the placement is stated per instruction:   scores S -> VGPRs (the softmax reads them),   O, row sums -> AGPRs
// (only MFMAs touch them in the loop),   Q fragments -> AGPRs (B operand),   K / V^T / P fragments -> VGPRs.
// The price: hipcc does not see an MFMA inside an asm statement, so the hazards are ours --
//   * MFMA result -> VALU read: the static schedule reads scores >= 7 slots after their last MFMA; every cold path
//     (tail, rescale branch, epilogue) drains the matrix pipe first, with the registers concerned tied to the drain;
//   * MFMA -> dependent MFMA (same accumulator): at least one independent 32x32x16 MFMA (32 cycles) sits between them;
//   * VALU result -> MFMA operand: two wait states, ours as well (the loop packs P a whole slot ahead; the tail pads);
//     LDS result -> MFMA operand: the explicit lgkmcnt waits.
__device__ __forceinline__ void mfma_s_first(f32x16& s, const bf16x8& kf, const bf16x8& q)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(s) : "v"(kf), "a"(q));
}
__device__ __forceinline__ void mfma_s(f32x16& s, const bf16x8& kf, const bf16x8& q)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(s) : "v"(kf), "a"(q));
}
// PF (here and below): P and the V image are fp16 instead of bf16 (the accurate mode, see XSoft); the 16-bit fragments travel in
// the same register types either way.
template <int PF = 0>
__device__ __forceinline__ void mfma_o(f32x16& o, const bf16x8& vf, const bf16x8& pfrag)
{
    if constexpr (pf_f16(PF)) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(o) : "v"(vf), "v"(pfrag));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o) : "v"(vf), "v"(pfrag));
}
template <int PF = 0>
__device__ __forceinline__ void mfma_l(f32x4_t& l, const bf16x8& ones, const bf16x8& pfrag)
{
    if constexpr (pf_f16(PF)) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(l) : "v"(ones), "v"(pfrag));
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(l) : "v"(ones), "v"(pfrag));
}

// Exponent bookkeeping of the three softmax mixes: p = 2^(c*s - m - kBias), and (rescaled mixes) the reference m of a wave moves
// when some score of the next sub-tile has c*s - m - kBias > kThr.
//   optimistic, bf16 P   m fixed after the first sub-tile, 2^100 of room either way, verified at the end (fa_bf16_common.h)
//   rescaled,   bf16 P   p <= 1 with the row maximum anywhere in 2^-64 .. 1 between two moves of the reference
//   rescaled,   fp16 P   fp16 spans 2^-14 .. 2^16 (normal): the row maximum is put at 2^0 when the reference moves and may grow to
//                        2^15.9 before it moves again.  The LOW end of the window is what decides the accuracy: below 2^-14 an fp16
//                        carries an absolute error of 2^-25, and on a long flat row (N = 16 384, scale 0.5) most of the weight sits
//                        in thousands of entries 2^-10 .. 2^-20 below the maximum.  With the maximum at 2^-5 (the first choice, 19
//                        bits of headroom) those were subnormal and the output error reached 1.9e-3 (a CPU simulation of the
//                        rounding alone: 2.6e-3 at 2^-5, 7.7e-4 at 2^-3, flat from 2^-1 upwards); with the maximum at 2^0 the
//                        same launch is at 6.2e-4.  The price is headroom, i.e. more moves of the reference on unit-variance
//                        data at scale 1 (c4, same box: 2^-5 0.327 ms, 2^-1 0.334, 2^0 ~0.338, 2^2 0.344, 2^6 0.373); at 1/sqrt(d)
//                        the reference never moves after the first sub-tile and the time is the same.
//   rescaled,   fp16 hi + lo   the same window; lo = fp16(p - hi) carries the next 11 bits of p (exact difference, one
//                        v_fma_mixlo/hi_f16 per score) while hi >= 2^-3 or so and fades out below (its own subnormals): P is good to
//                        ~2^-22 relative at the row maximum and never worse than one-term fp16 P.  Twice the P.V and row-sum MFMAs.
// optimistic mix, sampled exponent reference (xn_tile prologue): tiles of at least this many 64-key stages, first sub-tile spread (binades)
constexpr int kSampleMinStages = 64;
constexpr float kSampleSpread = 28.0f;

//   bf16 hi + lo (PF = 3)      the bf16 mixes as they are (optimistic first, rescaled redo): lo = bf16(p - hi) has bf16's exponent range,
//                        so nothing about the window changes; P is good to 2^-17 relative wherever lo is a normal number (p > 2^-117:
//                        17 binades below the optimistic reference; further down the terms fade to one-term accuracy and then to the
//                        exact zeros the clock likes)
template <bool OPT, int PF>
struct XSoft {
    static_assert(!(OPT && pf_f16(PF)), "fp16 P has no room for a fixed exponent reference");
    static constexpr float kBias = OPT ? kOptBias : pf_f16(PF) ? 0.0f : kLazyThr;
    static constexpr float kThr = pf_f16(PF) ? 15.9f : 0.0f;   // fp16: p < 2^15.9 = 61 147 (fp16 ends at 65 504)
};

// mfma_drain() with the registers it protects as operands: the drain is an asm statement without a data dependence of its
// own, and hipcc is free to schedule the VALU consumers of an asm MFMA's result in front of it (it did: the optimistic
// mix has no branch between the tail's K.Q^T and the exponentials, and they were hoisted above the bare drain).
template <int NB>
__device__ __forceinline__ void drain_scores(f32x16 (&s)[NB])
{
    if constexpr (NB == 4) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]));
    else asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(s[0]), "+v"(s[1]));
}
template <int NB, int DB>
__device__ __forceinline__ void drain_accumulators(f32x16 (&o)[NB][DB], BlockState (&st)[NB])
{
    if constexpr (NB == 4) {
        static_assert(DB == 2, "four blocks per wave exist at D = 64 only");
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                     : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[2][0]), "+a"(o[2][1]), "+a"(o[3][0]), "+a"(o[3][1]),
                       "+a"(st[0].lacc), "+a"(st[1].lacc), "+a"(st[2].lacc), "+a"(st[3].lacc));
    } else if constexpr (DB == 4) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                     : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[0][2]), "+a"(o[0][3]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(o[1][2]), "+a"(o[1][3]),
                       "+a"(st[0].lacc), "+a"(st[1].lacc));
    } else if constexpr (DB == 2) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                     : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]), "+a"(st[0].lacc), "+a"(st[1].lacc));
    } else {
        static_assert(DB == 1, "drain written for D = 32, 64, 128");
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[1][0]), "+a"(st[0].lacc), "+a"(st[1].lacc));
    }
}

// (rare, wave-uniform) move the exponent references of all blocks; everything still at the old reference is scaled once
template <int NB, int DB, int PF = 0>
__device__ __forceinline__ void xn_rescale(const float (&mx)[NB], float c, BlockState (&st)[NB], f32x16 (&o)[NB][DB], float (&off)[NB])
{
    using SM = XSoft<false, PF>;
    bool any = false;
    bool move[NB];   // wave-uniform: only the blocks that need it move (fp16 P, 16 binades of headroom, moves every few hundred keys on
                     // unscaled unit-variance data: -2.5 % on c4; and the branch keeps 36 VGPRs less alive than moving all blocks did,
                     // which is what lets the NB = 2 kernels at d = 64 fit a CU twice: 124 - 136 + 104 registers instead of 160 + 104)
    float mc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        mc[b] = mx[b] * c;
        mc[b] = fmaf(-fabsf(mc[b]), 0x1p-23f, mc[b]);
        const bool need = mc[b] - st[b].m > SM::kBias + SM::kThr;
        move[b] = __any(need);
        any = any || move[b];
    }
    if (__builtin_expect(any, 0)) {
        asm volatile("; lazy rescale" ::: "memory");
        drain_accumulators<NB, DB>(o, st);  // the accumulators rescaled below may have an MFMA in flight (hazard not padded across the branch)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (!move[b]) continue;
            const float nm = fmaxf(st[b].m, mc[b]);
            const float a = fast_exp2(st[b].m - nm);
            st[b].m = nm;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][db][r] *= a;
#pragma unroll
            for (int r = 0; r < 4; ++r) st[b].lacc[r] *= a;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) off[b] = st[b].m + SM::kBias;
}

// Optimistic mix, round 6: RE-CENTRE the exponent reference on the row sum a few times per long tile.  The reference of a row starts as a
// sampled score (192 keys: ~2.7 sigma on unit-variance data where the row maximum over 8192 keys sits at ~3.9 sigma -- 14 binades higher), and
// every term more than T binades below the reference is an exact zero, which is what the power budget likes.  After K1 keys
// the row sum l1 itself is a better anchor: the new reference is 2^(floor(log2 l1) - 1) <= l1 / 2 -- the dropped mass of the remaining keys is
// then bounded by nk * 2^-T * l1 / 2 <= 2^-(T + 1 - log2 nk) of the FINAL row sum (2^-11 for bf16 P, 2^-14 for two-term P: the same bound the
// sampled reference gives relative to the largest term, stated against the row sum, which is what O / l divides by).  The shift is an exact power
// of two in O, l and every later P: nothing is rounded.  Wave-uniform call between two groups of stages; per lane: its own row's shift.
template <int NB, int DB>
__device__ __forceinline__ void xn_recentre(BlockState (&st)[NB], f32x16 (&o)[NB][DB], float (&off)[NB], int bias)
{
    asm volatile("; optimistic mix: re-centre on the row sum" ::: "memory");
    drain_accumulators<NB, DB>(o, st);   // the accumulators read and scaled below may have an MFMA in flight
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const float lt = st[b].lacc[0];
        // P lives near 2^-bias (the reference term is exactly 2^-bias): the row sum is re-anchored at 2^(1 - bias) .. 2^(2 - bias)
        const int e = (int)((__float_as_uint(lt) >> 23) & 0xffu) - 127;   // floor(log2 lt) for a normal lt
        const int sh = e + bias - 1;
        const int shift = (lt > 0.0f && e < 0 && sh > 0) ? min(sh, 64) : 0;   // (zero, NaN, inf, a window about to overflow: left for the tile's verification)
        // block by block (like xn_rescale: fewer registers alive, which is what lets the NB = 2 kernels at d = 64 fit a CU twice); a block whose
        // rows already have their reference at their sum (scaled logits: 1 / sqrt(d)) has nothing to gain and skips the D-wide multiply
        if (!__any(shift >= 3)) continue;
        const float a = __uint_as_float((unsigned)(127 - shift) << 23);   // 2^-shift
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[b][db][r] *= a;
#pragma unroll
        for (int r = 0; r < 4; ++r) st[b].lacc[r] *= a;
        st[b].m += (float)shift;
        off[b] += (float)shift;
    }
}

// ---- static schedule of one step -------------------------------------------------------------------------------------
// MFMA slot i -> what it is
struct XSlot {
    int kind;  // 0 = K.Q^T, 1 = P.V, 2 = row sum
    int blk, idx;
    int term;  // P.V / row sum: 0 = P (hi), 1 = the lo term of P (PF = 2)
};
// P.V + row sums of one block, GRP slots, no two dependent MFMAs adjacent: P.V index p reads V^T fragment p = tt * DB + db,
// accumulates into o[blk][p % DB] and takes P fragment p / DB.  One term of P: P.V(tt = 0) x DB | row sum 0 | P.V(tt = 1) x DB | row
// sum 1.  Two terms (PF = 2): each of the two halves is run for hi, then for lo -- an accumulator is touched every DB + 1 slots.
template <int D, int NB, int PF = 0>
__device__ __host__ constexpr XSlot xn_pv_group(int blk, int j)
{
    using S = XShape<D, NB, PF>;
    constexpr int H = S::DB + 1;                 // slots of one (term, tt) run: DB P.V + one row sum
    const int run = j / H, w = j % H;            // run = tt * NT + term
    // (PF = 3: a whole block's dots are one unit behind the second fragment's pack: both hi runs first -- the lo fragments are complete later)
    const bool hi_first = PF == 3;
    const int tt = hi_first ? run % 2 : run / S::NT, term = hi_first ? run / 2 : run % S::NT;
    if (w == S::DB) return {2, blk, tt, term};
    return {1, blk, tt * S::DB + w, term};
}
template <int D, int NB, int PF = 0>
__device__ __host__ constexpr XSlot xn_slot(int i)
{
    using S = XShape<D, NB, PF>;
    constexpr int G = S::GRP;
    if constexpr (NB == 4) {
        if (i < 8) return {0, i % 2, i / 2, 0};                                       // K.Q^T A,B   k-step i/2
        if (i < 8 + G) return xn_pv_group<D, NB, PF>(0, i - 8);                       // P.V + row sums A
        if (i < 12 + G) return {0, 2 + (i - 8 - G) % 2, (i - 8 - G) / 2, 0};          // K.Q^T C,D   k-steps 0,1
        if (i < 12 + 2 * G) return xn_pv_group<D, NB, PF>(1, i - 12 - G);             // P.V + row sums B
        if (i < 16 + 2 * G) return {0, 2 + (i - 12 - 2 * G) % 2, 2 + (i - 12 - 2 * G) / 2, 0};   // K.Q^T C,D   k-steps 2,3
        if (i < 16 + 3 * G) return xn_pv_group<D, NB, PF>(2, i - 16 - 2 * G);         // P.V + row sums C
        return xn_pv_group<D, NB, PF>(3, i - 16 - 3 * G);                             // P.V + row sums D
    } else {
        if (i < 2 * S::KS) return {0, i % 2, i / 2, 0};                               // K.Q^T A,B   k-step i/2
        return xn_pv_group<D, NB, PF>((i - 2 * S::KS) / G, (i - 2 * S::KS) % G);
    }
}
// Instruction mixes that share the slot sequence:
//   OPT = false  the lazily rescaled softmax: exp + pack of sub-tile t, lane maxima of sub-tile t+1, rescale test (21 NB + 1 units;
//                PF = 2: four more per block for the lo term of P)
//   OPT = true   the optimistic softmax: the exponent reference of a row is fixed after its first sub-tile (with 2^100 of
//                headroom either way, see kOptBias), so the loop has no maxima, no test and no branch (18 NB units); the tile
//                is verified at the end and redone with OPT = false if any row left the safe range.
template <int D, int NB, int PF = 0>
constexpr int xn_num_units(bool opt) { return opt ? XShape<D, NB, PF>::kUnitsOpt : XShape<D, NB, PF>::kUnitsRsc; }
template <int D, int NB, int PF = 0>
__device__ __host__ constexpr int xn_weight_before(int i)  // in half-slots: a 32x32x16 slot = 2, a 16x16x32 slot = 1
{
    int w = 0;
    for (int k = 0; k < i; ++k) w += (xn_slot<D, NB, PF>(k).kind == 2) ? 1 : 2;
    return w;
}
// NB = 4: the VALU work of the optimistic mix has to be finished before the k-step-1 MFMAs of P.V D (slot 37; the fragment of its
// k-step 0 is packed early enough by the dealing rule), the rescaled mixes use the whole step; NB = 2: the offline-found limits of
// XShape.  Every dependency of the resulting tables is checked at compile time (xn_schedule_ok).
template <int D, int NB, int PF = 0>
constexpr int xn_weight_end(bool opt)
{
    using S = XShape<D, NB, PF>;
    if constexpr (NB == 4 && PF == 3) return opt ? S::kWendOpt4 : S::kWendRsc4;
    else if constexpr (NB == 4) return opt ? xn_weight_before<D, NB, PF>(16 + 3 * S::GRP + S::DB + 1) : xn_weight_before<D, NB, PF>(S::kSlots);
    else return opt ? S::kWendOpt2 : S::kWendRsc2;
}
// VALU units are dealt out by ISSUE COST, not by count: measured beside MFMAs (profiles/ubench/ubench_clock.hip) a plain
// VALU instruction occupies the wave's issue for 4 cycles and a v_exp_f32 for 8, so an exp element (fma + exp) costs 12, a
// pack (4 cvt) 16, half a lo fragment (4 v_fma_mix) 16, the max micro-steps 12 / 12 / 8, the test ~20 -- 1044 (896) cycles per
// NB = 4 step.  Slot i receives the units whose cumulative cost fits its share of the step's half-slots (24..36 cycles per full slot).
struct XTable {
    int ub[72];  // VALU units dealt out before slot i (kSlots + 1 entries used)
};
// The unit sequence.  A pack (v_cvt_pk) is scheduled two exp units after the last exponential it consumes: a VALU
// instruction that reads the result of a transcendental issued just before it costs a wait state (hipcc pads an s_nop).
// PF = 2: the two halves of a fragment's lo term follow its pack, one exp unit apart (they read the pack's result and the
// exponentials the pack read).
// PF = 3: the sixteen dots of a block (kind 4, one unit behind the second fragment's pack) and the four packs of each lo fragment (kind 5)
// are separate units; idx carries the number of the last half fragment a unit covers (3; 1 and 3).
struct XUnit {
    int kind;  // 0 = exp of one element, 1 = pack of one fragment, 2 = lane-max micro-step, 3 = rescale test, 4 = PF = 2: half a lo fragment
               // (complete), PF = 3: the block's dots, 5 = the packs of one lo fragment (PF = 3)
    int blk, idx, cost;
};
struct XUnitList {
    XUnit u[128];
};
template <int NB, int PF = 0>
__device__ __host__ constexpr XUnitList xn_make_units(bool opt)
{
    XUnitList l{};
    int n = 0;
    int npend = 0;
    XUnit pend[5] = {};
    for (int b = 0; b < NB; ++b) {
        for (int e = 0; e < 16; ++e) {
            l.u[n++] = {0, b, e, 12};
            if (npend > 0 && e >= 1 && e <= 5) {   // the previous block's second fragment: pack, then (two terms) its lo halves
                l.u[n++] = pend[0];
                for (int k = 0; k + 1 < 5; ++k) pend[k] = pend[k + 1];
                --npend;
            }
            if (e == 9) l.u[n++] = {1, b, 0, 16};
            if constexpr (PF == 2) {   // fp16 hi + lo (ablation library): half a lo fragment per unit, one v_fma_mix per element packs as it goes
                if (e == 10) l.u[n++] = {4, b, 0, 16};
                if (e == 11) l.u[n++] = {4, b, 1, 16};
            }
            // (PF = 3: the block's sixteen dots are ONE unit, behind the second fragment's pack -- in the next block's first exponentials)
        }
        pend[0] = {1, b, 1, 16};
        npend = 1;
        if constexpr (PF == 3) {   // dots of both fragments (the unit carries the index of the block's last half), then the packs of each lo fragment
            pend[1] = {4, b, 3, 64};
            pend[2] = {5, b, 1, 16};
            pend[3] = {5, b, 3, 16};
            npend = 4;
        } else if constexpr (PF == 2) {
            pend[1] = {4, b, 2, 16};
            pend[2] = {4, b, 3, 16};
            npend = 3;
        }
    }
    if (opt) {
        for (int k = 0; k < npend; ++k) l.u[n++] = pend[k];
        return l;
    }
    int used = 0;
    for (int b = 0; b < NB; ++b)
        for (int m = 0; m < 3; ++m) {
            l.u[n++] = {2, b, m, m == 2 ? 8 : 12};
            if (m >= 1 && used < npend) l.u[n++] = pend[used++];   // one term: after max(0, 1); two terms: the lo halves after max(0, 2), max(1, 1)
        }
    for (; used < npend; ++used) l.u[n++] = pend[used];   // (NB = 2, PF = 3: the last packs; not reached otherwise)
    l.u[n++] = {3, 0, 0, 20};
    return l;
}
template <int D, int NB, int PF = 0>
__device__ __host__ constexpr XTable xn_make_table(bool opt)
{
    constexpr int kS = XShape<D, NB, PF>::kSlots;
    static_assert(kS + 1 <= 72, "XTable::ub");
    const XUnitList l = xn_make_units<NB, PF>(opt);
    const int nu = xn_num_units<D, NB, PF>(opt), wend = xn_weight_end<D, NB, PF>(opt);
    XTable t{};
    int total = 0;
    for (int u = 0; u < nu; ++u) total += l.u[u].cost;
    int n = 0, cum_next = l.u[0].cost;  // cum_next: cost of units 0..n inclusive
    for (int i = 0; i <= kS; ++i) {
        const int wb = xn_weight_before<D, NB, PF>(i) < wend ? xn_weight_before<D, NB, PF>(i) : wend;
        const int target = total * wb / wend + 6;
        while (n < nu && cum_next <= target) {
            ++n;
            if (n < nu) cum_next += l.u[n].cost;
        }
        t.ub[i] = n;
    }
    t.ub[kS] = nu;
    // PF = 3: a slot that holds a dot unit takes kDotSlotExtra more (non-dot) units from the slot behind it.  The dots of a slot
    // are emitted last and need the matrix pipe EMPTY (~50 cycles after the slot's MFMA); the dot unit alone fills a slot's share of the
    // VALU work, so without this the dots start ~35 cycles behind the MFMA and wait out the rest, eight times per step.
    if (PF == 3 && pb2_dots_last(D)) {
        for (int i = 0; i + 2 <= kS; ++i) {
            bool dots = false;
            for (int u = t.ub[i]; u < t.ub[i + 1]; ++u) dots = dots || l.u[u].kind == 4;
            if (!dots) continue;
            for (int k = 0; k < kDotSlotExtra; ++k) {
                const int u = t.ub[i + 1];
                if (u >= t.ub[i + 2] || u >= nu || l.u[u].kind == 4 || l.u[u].kind == 5) break;   // nothing left / never another dot or lo-pack unit
                ++t.ub[i + 1];
            }
        }
    }
    return t;
}

// the unit list and the dealing table of an instantiated schedule, evaluated once (every slot and every unit of a step refers to them:
// re-deriving them per use made the two-term schedules a minute of constant evaluation per translation unit)
template <int NB, int PF, bool OPT>
inline constexpr XUnitList xn_units_v = xn_make_units<NB, PF>(OPT);
template <int D, int NB, int PF, bool OPT>
inline constexpr XTable xn_table_v = xn_make_table<D, NB, PF>(OPT);

// index of the unit that completes the P fragment MFMA slot sl reads: the pack of fragment tt (hi term), the second lo half (lo term;
// PF = 3: its packs)
template <int D, int NB, int PF = 0>
__device__ __host__ constexpr int xn_producer_unit(XSlot sl, bool opt)
{
    const int frag = sl.kind == 1 ? sl.idx / XShape<D, NB, PF>::DB : sl.idx;
    const XUnitList l = xn_make_units<NB, PF>(opt);
    for (int u = 0; u < xn_num_units<D, NB, PF>(opt); ++u) {
        if (sl.term == 0 && l.u[u].kind == 1 && l.u[u].blk == sl.blk && l.u[u].idx == frag) return u;
        if (sl.term == 1 && l.u[u].kind == (PF == 3 ? 5 : 4) && l.u[u].blk == sl.blk && l.u[u].idx == 2 * frag + 1) return u;
    }
    return 1 << 20;
}
// A VALU result needs two wait states before an MFMA reads it, and an asm MFMA is not padded by hipcc: when the dealing rule
// puts the last VALU instruction of a P fragment into the slot right in front of the first MFMA that reads it, that MFMA gets an s_nop 1.
template <int D, int NB, int PF, bool OPT>
__device__ __host__ constexpr bool xn_needs_pad(int i)
{
    const XSlot sl = xn_slot<D, NB, PF>(i);
    if (sl.kind == 0 || i == 0) return false;
    return xn_producer_unit<D, NB, PF>(sl, OPT) >= xn_table_v<D, NB, PF, OPT>.ub[i - 1];
}
// Compile-time check of every dependency of a step's static schedule (round 1 and 2 re-derived the tables offline,
// profiles/r01_x4_schedule_check.py):
//   * the unit that completes a P fragment is dealt out before the first MFMA slot that reads the fragment;
//   * the lane maxima of a block's next sub-tile start at least five slots after the block's last K.Q^T slot (MFMA result -> VALU read);
//   * the unit list has exactly the advertised number of units, and lo halves follow their pack, packs their exponentials.
template <int D, int NB, int PF = 0>
__device__ __host__ constexpr bool xn_schedule_ok(bool opt)
{
    using S = XShape<D, NB, PF>;
    const XUnitList l = xn_make_units<NB, PF>(opt);
    const XTable t = xn_make_table<D, NB, PF>(opt);
    const int nu = xn_num_units<D, NB, PF>(opt);
    // unit count: the list is zero-initialised past its end (cost 0) and every real unit has a cost
    for (int u = 0; u < nu; ++u)
        if (l.u[u].cost == 0) return false;
    if (nu < 128 && l.u[nu].cost != 0) return false;
    int last_qk[NB] = {};
    for (int i = 0; i < S::kSlots; ++i) {
        const XSlot sl = xn_slot<D, NB, PF>(i);
        if (sl.kind == 0) {
            last_qk[sl.blk] = i;
            continue;
        }
        if (sl.term >= S::NT) return false;
        if (xn_producer_unit<D, NB, PF>(sl, opt) >= t.ub[i]) return false;       // fragment not complete before its MFMA
    }
    int pos[NB][8] = {};   // per block: unit index of exp 7, exp 15, pack 0, pack 1, (PF = 3) dots of lo half 0 .. 3 (order checks)
    for (int u = 0; u < nu; ++u) {
        const XUnit un = l.u[u];
        if (un.kind == 0 && un.idx == 7) pos[un.blk][0] = u;
        if (un.kind == 0 && un.idx == 15) pos[un.blk][1] = u;
        if (un.kind == 1) pos[un.blk][2 + un.idx] = u;
        if (un.kind == 4) {
            if (u <= pos[un.blk][2 + un.idx / 2] || pos[un.blk][2 + un.idx / 2] == 0) return false;   // lo half in front of its pack
            pos[un.blk][4 + un.idx] = u;
        }
        if (un.kind == 5) {   // packs of a lo half: behind its dots, with a whole unit between (a dot result is readable three wait states
                              // later) and in a later slot (a slot's dots are emitted behind its other units: xn_units)
            const int ud = pos[un.blk][4 + 3];   // the block's one dot unit
            // (they may be neighbours: a pack reads dots that have the block's later dots and the earlier packs between -- three at least)
            if (PF != 3 || ud == 0 || u < ud + 1) return false;
            int sd = 0, sp = 0;
            while (sd < S::kSlots && !(t.ub[sd] <= ud && ud < t.ub[sd + 1])) ++sd;
            while (sp < S::kSlots && !(t.ub[sp] <= u && u < t.ub[sp + 1])) ++sp;
            if (pb2_dots_last(D) && sp <= sd) return false;
        }
        if (un.kind == 2 && un.idx == 0) {
            int s0 = 0;
            while (s0 < S::kSlots && !(t.ub[s0] <= u && u < t.ub[s0 + 1])) ++s0;
            if (s0 < last_qk[un.blk] + 5) return false;                           // maxima read scores still in the matrix pipe
        }
    }
    for (int b = 0; b < NB; ++b)
        if (pos[b][2] <= pos[b][0] || pos[b][3] <= pos[b][1]) return false;        // pack in front of an exponential it consumes
    return true;
}

// One step.  sc: scores of sub-tile t (consumed), sn: scores of sub-tile t+1 (produced); kf: K fragments of sub-tile t+1 on
// entry, of sub-tile t+2 (read from k_nxt / block kb_n2) on exit.  Returns the lane's rescale test for sub-tile t+1.
// Every slot / unit index is a template parameter (fold expressions over integer sequences): nothing here relies on the
// optimiser unrolling a 40 x 53 loop nest to resolve the register arrays.
template <int D, int NB>
struct XCtx {
    using S = XShape<D, NB>;   // (KS, DB, NV do not depend on the P format)
    const bf16x8& ones_a;
    const bf16x8 (&qf)[NB][S::KS];
    f32x16 (&sc)[NB];
    f32x16 (&sn)[NB];
    f32x16 (&o)[NB][S::DB];
    BlockState (&st)[NB];
    const float (&off)[NB];
    bf16x8 (&kf)[S::KS];
    float (&lm)[NB];
    float c;
    const char* k_nxt;
    int kb_n2, k_row_off, k_g;
    unsigned v_addr;
    s16x4 vlo[S::NV], vhi[S::NV];
    bf16x8 vf[S::NV];
    bf16x8 pf[NB][2];
    bf16x8 pl[NB][2];   // PF = 2: the lo term of P
    float pm[4];
    bool need;
};

template <int D, int NB, bool OPT, int U, int ABL = 0, int PF = 0>
__device__ __forceinline__ void xn_unit(XCtx<D, NB>& x)
{
    constexpr XUnit un = xn_units_v<NB, PF, OPT>.u[U];
    if constexpr (un.kind == 0 && (ABL & 256)) {
        // timing-only ablation: the exponential replaced by a plain VALU instruction of the same data flow
        const float t = fmaf(x.sc[un.blk][un.idx], x.c, -x.off[un.blk]);
        float r;
        asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r) : "v"(t));
        x.sc[un.blk][un.idx] = r;
    } else if constexpr (un.kind == 0) {
        // optimistic mix: no clamp -- an overflow has to reach the row sum (as a huge value or +inf): that is what the final
        // check reads.  fp16 P: no clamp either -- p ranges up to 2^kThr by design
        if constexpr (OPT || pf_f16(PF)) x.sc[un.blk][un.idx] = fast_exp2(fmaf(x.sc[un.blk][un.idx], x.c, -x.off[un.blk]));
        else x.sc[un.blk][un.idx] = exp2_clamp01(fmaf(x.sc[un.blk][un.idx], x.c, -x.off[un.blk]));
    } else if constexpr (un.kind == 1) {
        x.pf[un.blk][un.idx] = pack_p16x8<PF>(x.sc[un.blk], 8 * un.idx);
        asm volatile("" : "+v"(x.pf[un.blk][un.idx]));
    } else if constexpr (un.kind == 4 && PF == 3) {
        lo_dots_bf16(x.sc[un.blk], 0, 0, x.pf[un.blk][0]);   // both fragments, both halves: sixteen dots
        lo_dots_bf16(x.sc[un.blk], 0, 1, x.pf[un.blk][0]);
        lo_dots_bf16(x.sc[un.blk], 1, 0, x.pf[un.blk][1]);
        lo_dots_bf16(x.sc[un.blk], 1, 1, x.pf[un.blk][1]);
    } else if constexpr (un.kind == 5) {
        lo_packs_bf16(x.sc[un.blk], un.idx / 2, 0, x.pl[un.blk][un.idx / 2]);   // one lo fragment: four packs
        lo_packs_bf16(x.sc[un.blk], un.idx / 2, 1, x.pl[un.blk][un.idx / 2]);
    } else if constexpr (un.kind == 4) {
        lo_half(x.sc[un.blk], un.idx / 2, un.idx % 2, x.pf[un.blk][un.idx / 2], x.pl[un.blk][un.idx / 2]);
    } else if constexpr (un.kind == 2) {
        lanemax_step(un.idx, x.sn[un.blk], x.pm, x.lm[un.blk]);
    } else {
        float t = fmaf(x.lm[0], x.c, -x.off[0]);
#pragma unroll
        for (int b = 1; b < NB; ++b) t = fmaxf(t, fmaf(x.lm[b], x.c, -x.off[b]));
        x.need = t > XSoft<false, PF>::kThr;  // off = m + kBias
    }
}
// PASS: 0 = every unit of the slot in list order; PF = 3 (pb2_dots_last): 1 = all but the dot units, 2 = the dot units.  The dot
// product unit shares the matrix pipe: a v_dot2c issued behind an MFMA waits until that MFMA has left the pipe (four of them beside one
// 32-cycle MFMA: 63.8 cycles against 33.5 for four v_fma_f32, profiles/r04_ubench_dot2.txt), so a slot's dots go behind its other VALU
// work, where the pipe has (nearly) drained.  Legal because nothing in a slot depends on its dots: their packs sit in a later slot
// (xn_schedule_ok), and the dots read a pack of an earlier unit, which the reordering keeps in front of them.
// D = 32 only: the exponentials of a slot, all fmas first, then all v_exp_f32 (profiles/r04_ubench_valu_mix.txt: a v_exp_f32 issued right
// behind the v_fma_f32 whose result it reads costs the pair 16.0 cycles, four fmas followed by their four exponentials 13.6 each, and hipcc,
// left to order a slot's units, puts most exponentials directly behind their fma).  A scheduling barrier between the two groups holds the
// order; the instructions stay hipcc's own, so its hazard padding stays exact (as inline asm the groups cost an s_nop at every boundary).
// d = 32 is the one head dimension whose loop is bound by issue slots (+1.5 %, two-term +1.6 %); at d >= 64 the same change took ~4 % of
// the issue cycles out of the loops and no time -- they run at the board's power cap (DESIGN.md section 4.2, r04_experiments.txt part 7).
constexpr bool xn_group_exps(int D) { return D == 32; }
struct XExpList {
    int n;
    int blk[8], idx[8];
};
// the exp units among units [u0, u0 + cnt) of the step, from the skip-th one on (eight at most)
template <int NB, int PF, bool OPT>
__device__ __host__ constexpr XExpList xn_slot_exps(int u0, int cnt, int skip)
{
    XExpList l{};
    int seen = 0;
    for (int u = u0; u < u0 + cnt; ++u) {
        const XUnit un = xn_units_v<NB, PF, OPT>.u[u];
        if (un.kind != 0) continue;
        if (seen++ < skip || l.n == 8) continue;
        l.blk[l.n] = un.blk;
        l.idx[l.n] = un.idx;
        ++l.n;
    }
    return l;
}
template <int D, int NB, bool OPT, int PF, int U0, int CNT, int SKIP>
__device__ __forceinline__ void xn_exp_group(XCtx<D, NB>& x)
{
    constexpr XExpList L = xn_slot_exps<NB, PF, OPT>(U0, CNT, SKIP);
    if constexpr (L.n > 0) {
        float t[8];
#define FA_E(i) x.sc[L.blk[i]][L.idx[i]]
#define FA_T(i) if constexpr (L.n > i) t[i] = fmaf(FA_E(i), x.c, -x.off[L.blk[i]]);
        FA_T(0) FA_T(1) FA_T(2) FA_T(3) FA_T(4) FA_T(5) FA_T(6) FA_T(7)
        if constexpr (L.n > 1) __builtin_amdgcn_sched_barrier(0);   // every fma of the slot in front of its first exponential
        // (clamp: as in xn_unit)
#define FA_X(i) if constexpr (L.n > i) FA_E(i) = (OPT || pf_f16(PF)) ? fast_exp2(t[i]) : exp2_clamp01(t[i]);
        FA_X(0) FA_X(1) FA_X(2) FA_X(3) FA_X(4) FA_X(5) FA_X(6) FA_X(7)
#undef FA_X
#undef FA_T
#undef FA_E
    }
}
template <int D, int NB, bool OPT, int U, int ABL, int PF, int PASS>
__device__ __forceinline__ void xn_unit_pass(XCtx<D, NB>& x)
{
    constexpr bool is_dot = PF == 3 && xn_units_v<NB, PF, OPT>.u[U].kind == 4;
    constexpr bool grouped_exp = xn_group_exps(D) && !(ABL & 256) && xn_units_v<NB, PF, OPT>.u[U].kind == 0;   // (emitted by xn_exp_group)
    if constexpr (!grouped_exp && (PASS == 0 || (PASS == 1 && !is_dot) || (PASS == 2 && is_dot))) xn_unit<D, NB, OPT, U, ABL, PF>(x);
}
template <int D, int NB, bool OPT, int ABL, int PF, int U0, int... Us>
__device__ __forceinline__ void xn_units(XCtx<D, NB>& x, std::integer_sequence<int, Us...>)
{
    if constexpr (xn_group_exps(D) && !(ABL & 256)) {
        xn_exp_group<D, NB, OPT, PF, U0, (int)sizeof...(Us), 0>(x);
        xn_exp_group<D, NB, OPT, PF, U0, (int)sizeof...(Us), 8>(x);
        static_assert(xn_slot_exps<NB, PF, OPT>(U0, (int)sizeof...(Us), 16).n == 0, "more than sixteen exponentials in one slot");
    }
    if constexpr (PF == 3 && pb2_dots_last(D)) {
        (xn_unit_pass<D, NB, OPT, U0 + Us, ABL, PF, 1>(x), ...);
        (xn_unit_pass<D, NB, OPT, U0 + Us, ABL, PF, 2>(x), ...);
    } else {
        (xn_unit_pass<D, NB, OPT, U0 + Us, ABL, PF, 0>(x), ...);
    }
}

// wait for the asm-issued V^T reads (two ds_read per fragment, LDS returns in order) and hand the fragments to the MFMA operands
// first slot of block A's P.V group that reads V^T fragment v
template <int D, int NB, int PF>
__device__ __host__ constexpr int xn_first_use_of_v(int v)
{
    for (int i = XShape<D, NB, PF>::kFirstPv; i < XShape<D, NB, PF>::kSlots; ++i)
        if (xn_slot<D, NB, PF>(i).kind == 1 && xn_slot<D, NB, PF>(i).idx == v) return i;
    return -1;
}
template <int D, int NB, int I, int ABL, int PF>
__device__ __forceinline__ void xn_wait_v_frags(XCtx<D, NB>& x)
{
    using S = XShape<D, NB, PF>;
    if constexpr (S::kVWaitPerFrag) {
        // slots of block A's P.V group that consume fragment v first: kFirstPv + {0, 1, 3, 4} at D = 64 with one term of P (the row-sum
        // slot sits between), + {0, 1, 6, 7} with two.  Only the V^T reads are in flight here (the K reads start behind block A's group)
        static_assert(S::NV == 4, "per-fragment waits written for D = 64");
        constexpr int v = I == xn_first_use_of_v<D, NB, PF>(0) ? 0 : I == xn_first_use_of_v<D, NB, PF>(1) ? 1 : I == xn_first_use_of_v<D, NB, PF>(2) ? 2
                          : I == xn_first_use_of_v<D, NB, PF>(3) ? 3 : -1;
        static_assert(xn_first_use_of_v<D, NB, PF>(3) < S::kKLoad, "the per-fragment lgkmcnt values assume no K read has been issued yet");
        if constexpr (v >= 0) {
            if constexpr (ABL & 32) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x.vlo[v]), "+v"(x.vhi[v]));
            else if constexpr (ABL & 4) asm volatile("" : "+v"(x.vlo[v]), "+v"(x.vhi[v]));  // timing-only ablation: no wait
            else if constexpr (v == 0) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(x.vlo[0]), "+v"(x.vhi[0]));
            else if constexpr (v == 1) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x.vlo[1]), "+v"(x.vhi[1]));
            else if constexpr (v == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x.vlo[2]), "+v"(x.vhi[2]));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x.vlo[3]), "+v"(x.vhi[3]));
            __builtin_amdgcn_sched_barrier(0);
            x.vf[v] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(x.vlo[v], x.vhi[v], 0, 1, 2, 3, 4, 5, 6, 7));
        }
    } else if constexpr (I == S::kFirstPv) {
        // the reads were started in slots 0 .. NV-1, at least KS slots ago: one wait in front of the first P.V slot orders them all
        // (the K reads of the next step start after it)
        if constexpr (S::NV == 8) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(x.vlo[0]), "+v"(x.vhi[0]), "+v"(x.vlo[1]), "+v"(x.vhi[1]), "+v"(x.vlo[2]), "+v"(x.vhi[2]), "+v"(x.vlo[3]), "+v"(x.vhi[3]));
            asm volatile("" : "+v"(x.vlo[4]), "+v"(x.vhi[4]), "+v"(x.vlo[5]), "+v"(x.vhi[5]), "+v"(x.vlo[6]), "+v"(x.vhi[6]), "+v"(x.vlo[7]), "+v"(x.vhi[7]));
        } else if constexpr (S::NV == 4) {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(x.vlo[0]), "+v"(x.vhi[0]), "+v"(x.vlo[1]), "+v"(x.vhi[1]), "+v"(x.vlo[2]), "+v"(x.vhi[2]), "+v"(x.vlo[3]), "+v"(x.vhi[3]));
        } else {
            static_assert(S::NV == 2, "fragment wait written for D = 32, 64, 128");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x.vlo[0]), "+v"(x.vhi[0]), "+v"(x.vlo[1]), "+v"(x.vhi[1]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < S::NV; ++v) x.vf[v] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(x.vlo[v], x.vhi[v], 0, 1, 2, 3, 4, 5, 6, 7));
    }
}

// the matrix instruction of slot I with everything tied to it: the waits for its V^T fragments, the fragment reads that start in its slot,
// the pad behind a VALU producer, the K reads of the next step
template <int D, int NB, int KB_C, int I, int ABL, bool OPT, int PF>
__device__ __forceinline__ void xn_slot_mfma(XCtx<D, NB>& x)
{
    using S = XShape<D, NB, PF>;
    constexpr XSlot sl = xn_slot<D, NB, PF>(I);
    xn_wait_v_frags<D, NB, I, ABL, PF>(x);
    if constexpr (I < S::NV && !(ABL & 16)) load_v_frag_asm<D, KB_C, I>(x.v_addr, x.vlo[I], x.vhi[I]);  // ABL & 16: no LDS fragment reads
    if constexpr (I < S::NV && (ABL & 16)) asm volatile("" : "=v"(x.vlo[I]), "=v"(x.vhi[I]));
    if constexpr (I < S::NV && (ABL & 32)) load_v_frag_asm<D, KB_C, I>(x.v_addr, x.vlo[I], x.vhi[I]);  // ABL & 32: every fragment read issued twice
    if constexpr (xn_needs_pad<D, NB, PF, OPT>(I)) asm volatile("s_nop 1");
    if constexpr (ABL & 1) {
        // timing-only ablation: no matrix instructions
    } else if constexpr (sl.kind == 0) {
        if constexpr (sl.idx == 0) mfma_s_first(x.sn[sl.blk], x.kf[sl.idx], x.qf[sl.blk][sl.idx]);
        else mfma_s(x.sn[sl.blk], x.kf[sl.idx], x.qf[sl.blk][sl.idx]);
    } else if constexpr (sl.kind == 1) {
        if constexpr (sl.term == 0) mfma_o<PF>(x.o[sl.blk][sl.idx % S::DB], x.vf[sl.idx], x.pf[sl.blk][sl.idx / S::DB]);
        else mfma_o<PF>(x.o[sl.blk][sl.idx % S::DB], x.vf[sl.idx], x.pl[sl.blk][sl.idx / S::DB]);
    } else {
        if constexpr (sl.term == 0) mfma_l<PF>(x.st[sl.blk].lacc, x.ones_a, x.pf[sl.blk][sl.idx]);
        else mfma_l<PF>(x.st[sl.blk].lacc, x.ones_a, x.pl[sl.blk][sl.idx]);
    }
    if constexpr (I >= S::kKLoad && I < S::kKLoad + S::KS) {  // K fragments of the next step (this step's last K.Q^T slot lies behind)
        // asm, like the V^T reads: a compiler-visible LDS load would make hipcc put its own lgkmcnt waits in front of the
        // next step's K.Q^T MFMAs, and those waits -- counted without the asm reads in flight -- drain the V^T reads too
        constexpr int ks = I - S::kKLoad;
        const unsigned a = (unsigned)(size_t)(lds_s16x4_t*)(x.k_nxt + x.k_row_off + x.kb_n2 * 32 * (2 * D) + (((2 * ks) ^ x.k_g) * 16));
        if constexpr (ABL & 16) asm volatile("" : "=v"(x.kf[ks]) : "v"(a));
        else if constexpr (ABL & 64) asm volatile("" : "+v"(x.kf[ks]) : "v"(a));  // timing-only: keep the previous step's (random) fragments
        else asm volatile("ds_read_b128 %0, %1" : "=v"(x.kf[ks]) : "v"(a));
        if constexpr (ABL & 32) asm volatile("ds_read_b128 %0, %1" : "=v"(x.kf[ks]) : "v"(a));
    }
}
template <int D, int NB, int KB_C, int I, int ABL, bool OPT, int PF>
__device__ __forceinline__ void xn_slot_body(XCtx<D, NB>& x)
{
    constexpr XTable tab = xn_table_v<D, NB, PF, OPT>;
    xn_slot_mfma<D, NB, KB_C, I, ABL, OPT, PF>(x);
    if constexpr (!(ABL & 2))   // ABL & 2: no VALU work
        xn_units<D, NB, OPT, ABL, PF, tab.ub[I]>(x, std::make_integer_sequence<int, tab.ub[I + 1] - tab.ub[I]>{});
    if constexpr (!(ABL & 128)) __builtin_amdgcn_sched_barrier(0);  // ABL & 128: slots not pinned (hipcc schedules the step)
}
template <int D, int NB, int KB_C, int ABL, bool OPT, int PF, int... Is>
__device__ __forceinline__ void xn_slots(XCtx<D, NB>& x, std::integer_sequence<int, Is...>)
{
    (xn_slot_body<D, NB, KB_C, Is, ABL, OPT, PF>(x), ...);
}

template <int D, int NB, int KB_C, int ABL = 0, bool OPT = false, int PF = 0>
__device__ __forceinline__ bool xn_step(const char* v_lds, const char* k_nxt, int kb_n2, int k_row_off, int k_g, int v_lane_off,
                                        const bf16x8& ones_a, const bf16x8 (&qf)[NB][XShape<D, NB>::KS], f32x16 (&sc)[NB], f32x16 (&sn)[NB],
                                        f32x16 (&o)[NB][XShape<D, NB>::DB], BlockState (&st)[NB], float c, const float (&off)[NB],
                                        bf16x8 (&kf)[XShape<D, NB>::KS], float (&lm)[NB])
{
    XCtx<D, NB> x{ones_a, qf, sc, sn, o, st, off, kf, lm, c, k_nxt, kb_n2, k_row_off, k_g, (unsigned)(size_t)(lds_s16x4_t*)(v_lds + v_lane_off)};
    x.need = false;
    static_assert(xn_schedule_ok<D, NB, PF>(OPT), "static schedule of the step: a P fragment is read before it is complete, or the lane maxima read scores in flight");
    xn_slots<D, NB, KB_C, ABL, OPT, PF>(x, std::make_integer_sequence<int, XShape<D, NB, PF>::kSlots>{});
    // the K reads are at least a P.V group old: this wait is free, and it keeps every asm-issued load inside the basic block that
    // issued it (hipcc may move or spill a register across a branch without knowing a load is in flight)
    if constexpr (XShape<D, NB>::KS == 8)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]), "+v"(kf[4]), "+v"(kf[5]), "+v"(kf[6]), "+v"(kf[7]));
    else if constexpr (XShape<D, NB>::KS == 4)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]), "+v"(kf[2]), "+v"(kf[3]));
    else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[0]), "+v"(kf[1]));
    return x.need;
}

// One (NWAVES * 32 * NB)-row tile.  OPT: optimistic mix; returns false when some row of the workgroup left the safe range (its
// results were stored and are overwritten by the redo).
template <int D, int NB, int NWAVES, bool CAUSAL, bool OUT_F32, int G, int ABL, bool OPT, int PF = 0>
__device__ __forceinline__ int xn_tile(const FwdParams& p, char* smem)   // 1 = done; 0 = redo; 2 = redo unless V is exactly zero (xn_kernel_body)
{
    constexpr int KS = XShape<D, NB>::KS, DB = XShape<D, NB>::DB;
    constexpr float kBiasC = XSoft<OPT, PF>::kBias;
    const unsigned long long prof_entry = (ABL & 1024) ? stamp() : 0ull;
    const unsigned long long prof_rt_in = (ABL & 2048) ? __builtin_amdgcn_s_memrealtime() : 0ull;   // timeline: 100 MHz ticks at entry
    using C = Bf16Cfg<D, NWAVES>;
    constexpr int BM = NWAVES * 32 * NB;
    constexpr int KR = 2 * G, VR = 2 * G;
    static_assert(G == 1 || G == 2, "ring index arithmetic written for G = 1, 2");
    constexpr int T = C::kTileBytes;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;

    char* const k_ring = smem;  // K ring, then V ring
    char* const v_ring = k_ring + KR * T;

    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles;
    int qt = w % p.q_tiles;
    if (CAUSAL) qt = causal_tile(p, qt);
    const int n = p.n;
    const int q0 = qt * BM + wave * 32 * NB;  // first row of block 0; block b starts at q0 + 32 b

    const int b = slab / p.heads, h = slab % p.heads;
    const __bf16* qg = (const __bf16*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
    const __bf16* kg = (const __bf16*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const __bf16* vg = (const __bf16*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    const int64_t o_slab_off = b * p.o_batch_stride + h * p.o_head_stride;

    // Keys of this workgroup: all n, or its share of a key-split launch (FwdParams::n_kv): the workgroup whose "head" index is h reads the
    // keys [h * n_kv, min((h + 1) * n_kv, n_kv_total)) -- kv_head_stride carries the offset -- and works in LOCAL key indices (local key i
    // is key kbeg + i of the slab).  Causal shares: n_kv is a multiple of the tile height, so a share starts at or below the tile's first
    // row (every row then sees the share's first key and has an exponent reference) or past its last row (an empty share: lse = -inf is
    // stored, the combine gives it weight 0 and never reads its O).
    int nk = n, kbeg = 0;
    if (p.n_kv > 0) {
        kbeg = h * p.n_kv;
        nk = min(p.n_kv, p.n_kv_total - kbeg);
    }
    int kv_end = nk;
    if (CAUSAL) kv_end = min(nk, qt * BM + BM - kbeg);
    const bool idle = CAUSAL && kv_end <= 0;         // causal key share entirely above this tile's rows
    if (idle) kv_end = 0;
    const int nst = (kv_end + kKvBlk - 1) / kKvBlk;  // 64-key stages
    const int nsub = (kv_end + 31) / 32;             // 32-key sub-tiles
    const int q0r = q0 - kbeg;                       // first row of block 0 in local key coordinates (causal: local key <= local row)
    // The optimistic mix drops every term more than 126 - bias binades below the row's reference (bf16 underflow: exact zeros).  Zeros are
    // what the power budget likes (these loops run at the board's cap, and a sparser P buys clock on an unchanged
    // instruction stream), so the bias is as high as the accuracy of the P format allows -- the dropped mass is bounded by nk 2^-T:
    //   bf16 P (PF = 0)    T = 10 + ceil(log2 nk)   dropped mass <= 2^-10 of the reference term, half of what rounding P to 8 bits may cost
    //                      (nk = 8192: bias 103, T = 23; round 4: +1.2 % on every bf16-P shape against T = 26, errors on random data unchanged
    //                      to the last digit -- profiles/r04_experiments.txt, part 8; T = 22: +1.5 %, T = 24: +1.0 %, T = 26: the form before);
    //   two bf16 terms     T = 26 up to 8192 keys, one more per doubling beyond: <= 2^-13 (the range test at the end of the tile is unchanged).
    const int lg_nk = 32 - __builtin_clz(max(nk, 2) - 1);   // ceil(log2 nk)
    const float kBias = (OPT && PF == 3) ? kBiasC - (float)max(0, lg_nk - 13) : (OPT && PF == 0) ? kBiasC + (float)min(9, max(0, 26 - (10 + lg_nk))) : kBiasC;
    // (the short-row increment stops at 9 -- T = 17 for nk <= 128 --: 2^-(bias + 16) must stay a NORMAL fp32 number, v_exp_f32 flushes
    // subnormal results to zero and a zero threshold could never send a tiny-V tile to the redo: ADVICE r04)
    const float tiny_acc = (OPT && PF == 0) ? __builtin_amdgcn_exp2f(-(kBias + 16.0f)) : kOptTinyAcc;   // (kOptTinyAcc = 2^-(100 + 16); >= 2^-125 here)

    auto k_slot = [&](int j) { return k_ring + (j & (KR - 1)) * T; };
    auto v_slot = [&](int j) { return v_ring + (j & (VR - 1)) * T; };

    TileDma<D, NWAVES> dma;
    dma.init(kg, vg, max(nk, 1), p.kv_row_stride, wave, lane);
    if (!idle) dma.issue_k(0u, k_slot(0), wave);
    // every tile the first barrier group needs is requested before anything is waited for: one memory round trip, not two
#pragma unroll
    for (int g = 1; g <= G; ++g)
        if (g < nst) dma.issue_k((unsigned)g * dma.stage_step, k_slot(g), wave);
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (g < nst) dma.issue_v((unsigned)g * dma.stage_step, v_slot(g), wave);

    bf16x8 qf[NB][KS];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const __bf16* qr = qg + (int64_t)min(q0 + 32 * blk + lq, n - 1) * p.q_row_stride + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[blk][ks] = *(const bf16x8*)(qr + ks * 16);
    }
    const bf16x8 ones_a = rowsum_a_operand<PF>(lane);

    f32x16 o[NB][DB], s0[NB], s1[NB];
    BlockState st[NB];
    float off[NB], lm[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        st[blk].m = -INFINITY;
        lm[blk] = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) st[blk].lacc[r] = 0.0f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[blk][db][r] = 0.0f;
    }

    const int k_row_off = lq * C::kRowBytes;
    const int k_g = hi ^ k_swizzle<D>(lq);
    const int li = lane & 15;
    const int v_lane_off = (hi * (D / 16) + ((lane >> 4) & 1)) * 128 + (li >> 2) * 32 + (li & 3) * 8;
    const float c = p.scale_log2e;

    // sub-tile t needs a mask for the block whose first row is qb?
    auto needs_mask = [&](int t, int qb) { return (t * 32 + 32 > nk) || (CAUSAL && (t * 32 + 31 > qb)); };

    // Top of stage j, j a multiple of G: K(j+1 .. j+G), V(j .. j+G-1) visible; K(j+G+1 .. j+2G), V(j+G .. j+2G-1) enqueued into the
    // ring slots nobody reads any more (K tiles are only read into kf one step ahead of their use, and every LDS read of
    // a wave has returned before it arrives at the barrier).
    auto sync_top = [&](int j) {
        if (!(ABL & 8)) {  // ABL & 8: timing-only ablation without the wait + barrier
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if constexpr (ABL & 512) return;   // timing-only ablation: no LDS-DMA in the loop (the tiles of the prologue are reused)
#pragma unroll
        for (int g = 1; g <= G; ++g)
            if (__builtin_expect(j + G + g < nst, 1)) dma.issue_k((unsigned)(j + G + g) * dma.stage_step, k_slot(j + G + g), wave);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (__builtin_expect(j + G + g < nst, 1)) dma.issue_v((unsigned)(j + G + g) * dma.stage_step, v_slot(j + G + g), wave);
    };
    bf16x8 kf[KS];
    auto load_kf = [&](int t) {
        const char* k_lds = k_slot(t >> 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = load_k_frag<D>(k_lds, k_row_off, k_g, t & 1, ks);
    };
    // scores of sub-tile t for all blocks from the fragments in kf, phase-structured (prologue and tail)
    float mx_sample[NB];   // optimistic mix: row maxima over the sub-tiles the prologue samples (see the prologue)
    bool wide = false;     // ... and whether the first sub-tile's scores spread over enough binades for that to pay
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) mx_sample[blk] = -INFINITY;
    auto qk_regs = [&](int t, f32x16 (&s)[NB], bool first = false, bool sample = false) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                if (ks == 0) mfma_s_first(s[blk], kf[ks], qf[blk][ks]);
                else mfma_s(s[blk], kf[ks], qf[blk][ks]);
            }
        drain_scores<NB>(s);  // cold path: let the scores retire before the VALU reads them
        float mx[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            if (needs_mask(t, q0r + 32 * blk)) mask16(s[blk], t * 32, q0r + 32 * blk + lq, nk, hi, CAUSAL);
            mx[blk] = rowmax16(s[blk]);
        }
        if (sample) {   // only the maxima are kept
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) mx_sample[blk] = fmaxf(mx_sample[blk], mx[blk]);
            return;
        }
        if (first) {  // nothing accumulated yet: set the references, leave the (zero) accumulators alone
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                mx_sample[blk] = mx[blk];
                const float mc = mx[blk] * c;
                st[blk].m = fmaf(-fabsf(mc), 0x1p-23f, mc);
                off[blk] = st[blk].m + kBias;
                if constexpr (OPT && FA_OPT_SAMPLE) wide = wide || (mx[blk] - rowmin16(s[blk])) * c > kSampleSpread;
            }
        } else if (!OPT) {
            xn_rescale<NB, DB, PF>(mx, c, st, o, off);
        }
    };
    // exp, pack, P.V and row sums of sub-tile t for all blocks, phase-structured (tail)
    auto finish_sub = [&](int t, f32x16 (&s)[NB]) {
        const char* v_lds = v_slot(t >> 1);
        bf16x8 vf[2 * DB];
#pragma unroll
        for (int v = 0; v < 2 * DB; ++v) vf[v] = load_v_frag<D>(v_lds, v_lane_off, t & 1, v);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            bf16x8 pf[2];
            exp_range<PF>(s[blk], pf, c, off[blk], 0, 8, !OPT && !pf_f16(PF));
            exp_range<PF>(s[blk], pf, c, off[blk], 8, 16, !OPT && !pf_f16(PF));
            bf16x8 pl[2];
            if constexpr (PF == 3) {
                asm volatile("s_nop 0" : "+v"(pf[0]), "+v"(pf[1]));   // as below
                lo_frag_bf16(s[blk], 0, pf[0], pl[0]);
                lo_frag_bf16(s[blk], 1, pf[1], pl[1]);
                asm volatile("s_nop 1" : "+v"(pl[0]), "+v"(pl[1]));
            }
            if constexpr (PF == 2) {
                asm volatile("s_nop 0" : "+v"(pf[0]), "+v"(pf[1]));   // the lo halves are asm: a transcendental's result needs a wait state
#pragma unroll
                for (int u = 0; u < 4; ++u) lo_half(s[blk], u / 2, u % 2, pf[u / 2], pl[u / 2]);
                asm volatile("s_nop 1" : "+v"(pl[0]), "+v"(pl[1]));
            }
            // a VALU result needs two wait states before an MFMA may read it; hipcc counts them for its own MFMAs, not
            // for an asm one (the pipelined loop packs P at least one whole slot ahead of its first use)
            asm volatile("s_nop 1" : "+v"(pf[0]), "+v"(pf[1]));
#pragma unroll
            for (int v = 0; v < 2 * DB; ++v) mfma_o<PF>(o[blk][v % DB], vf[v], pf[v / DB]);
            mfma_l<PF>(st[blk].lacc, ones_a, pf[0]);
            asm volatile("s_nop 7");  // dependent row-sum MFMAs back to back: the hazard is ours
            mfma_l<PF>(st[blk].lacc, ones_a, pf[1]);
            if constexpr (PF >= 2) {
#pragma unroll
                for (int v = 0; v < 2 * DB; ++v) mfma_o<PF>(o[blk][v % DB], vf[v], pl[v / DB]);
                mfma_l<PF>(st[blk].lacc, ones_a, pl[0]);
                asm volatile("s_nop 7");
                mfma_l<PF>(st[blk].lacc, ones_a, pl[1]);
            }
        }
    };

    // ---------------- prologue: K(0) landed -> scores of sub-tile 0, fragments of sub-tile 1 ----------------
    wait_lds_dma();
    __syncthreads();
    unsigned long long prof_landed = 0;
    if constexpr ((ABL & 1024) != 0) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(qf[0][0]), "+v"(qf[NB - 1][KS - 1]));   // Q fragments have arrived as well
        prof_landed = stamp();
    }
    if (!idle) {
        load_kf(0);
        qk_regs(0, s0, true);
        // Optimistic mix: the exponent reference of a row is fixed for the whole tile, and every P more than 26 binades below it underflows
        // bf16 to an exact zero.  Zero operands cost the matrix core less power, and the chip runs to its power budget (DESIGN.md 4.6:
        // c4 at scale 2, 80 % zeros, runs 5.5 % faster than at scale 1, 40 % zeros, through the same instruction stream).  ANY score of
        // the row is a valid reference (a lower bound of the row maximum), so the maximum over all sub-tiles the prologue has in LDS
        // anyway (K stages 0 .. G: 64 (G + 1) keys) is taken instead of the first sub-tile's: 2.7 sigma instead of 2.0 sigma expected on
        // random data.  Costs 2 G + 1 phase-structured K.Q^T passes, so only long tiles take it (measured on one box, sampled / not:
        // c4 0.2319 / 0.2363 ms, c5's shard 1.806 / 1.845, d = 128 0.4136 / 0.4203, but 16 x 2048 0.0364 / 0.0347), and only waves whose
        // first sub-tile spreads over enough binades for anything to underflow (at 1/sqrt(d) scaling nothing does: 0.2400 / 0.2380).
        // Round 6, last day: with the reference RE-CENTRED on the row sum eight stages into the tile (xn_recentre) the sample only shapes the first
        // 512 of 8192 keys, and its five extra K.Q^T passes cost ~1 % of a c4 tile: without it c4 -1.4 % (ablation libraries; product libraries -0.5 %),
        // c5's shard -1.6 %, d = 128 -1.0 %, d = 32 -1.4 %, 32 x 4096 -2.3 %, errors identical; re-centring earlier (2 or 4 stages in) buys nothing
        // on top (profiles/r06_exp15_sample_vs_recentre.txt).  So a tile that is going to re-centre does not sample -- which today is every long
        // tile of the product's launches; the sample remains the reference of tiles that do not (the ablation library's forms), and `wide`
        // (the first sub-tile's spread) still gates the re-centring.
        bool sampled = false;
        if constexpr (OPT && FA_OPT_SAMPLE) {
            // (rows of kSampleMinStages stages have their 32 fast stages, causal or not: every tile that would sample re-centres)
            // (the two-term NB = 2 kernels keep their sample: at d = 64 they sit at exactly 256 registers, and without the sampling loop hipcc's
            // allocation of the fp32-output instantiation lands on 260 -- one workgroup per CU instead of two; they gain 0.5 % at most)
            constexpr bool kRecentres = ABL == 0 && FA_OPT_RECENTRE != 0 && !(PF == 3 && NB == 2);
            if ((FA_OPT_SAMPLE_WHEN_RECENTRED != 0 || !kRecentres) && nst >= kSampleMinStages && __any(wide)) {
                const int ts = min(min(nsub, 2 * min(G + 1, nst)), FA_OPT_SAMPLE_SUBTILES + 1);   // sub-tiles of the K stages the prologue requested
#pragma unroll 1
                for (int t = ts - 1; t >= 1; --t) {              // ends with sub-tile 1: its fragments stay in kf for the first step
                    load_kf(t);
                    qk_regs(t, s1, false, true);
                }
                sampled = ts > 1;
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) {
                    const float mc = mx_sample[blk] * c;
                    st[blk].m = fmaf(-fabsf(mc), 0x1p-23f, mc);
                    off[blk] = st[blk].m + kBias;
                }
            }
        }
        if (!sampled) load_kf(1);
    }

    // ---------------- fast loop: groups of G whole stages whose sub-tiles 2j .. 2j+2 are in range and mask-free ----------------
    // (closed form of: jf = 0; while ((2 jf + 3) 32 <= kv_end && !needs_mask(2 jf + 2, q0) && !needs_mask(0, q0)) ++jf; -- the loop was
    // O(N / 64) scalar iterations per wave, ~2.5k cycles at N = 8192)
    int jf = kv_end >= 96 ? (kv_end / 32 - 3) / 2 + 1 : 0;                    // (2 jf + 3) * 32 <= kv_end; kv_end <= n covers the ragged tail
    if (CAUSAL) jf = q0r >= 95 ? min(jf, (q0r - 95) / 64 + 1) : 0;           // 64 j + 95 <= q0: sub-tile 2 j + 2 lies below the first row's diagonal
    if (nk < 32) jf = 0;
    // The last stage may run in the fast loop too when its own two sub-tiles are whole and mask-free: its second step then
    // computes scores of a sub-tile that does not exist (from whatever the ring slot holds) and nobody consumes them --
    // the rescale test of that step is ignored.  Without this the final 128 keys of every slab took the slow tail path.
    if (jf == nst - 1 && (2 * jf + 2) * 32 <= kv_end && !needs_mask(2 * jf + 1, q0r) && !needs_mask(0, q0r)) jf = nst;
    jf -= jf % G;
    unsigned long long prof_t0 = 0, prof_r0 = 0;
    if constexpr ((ABL & 1024) != 0) {   // ablation library only: cycle stamps around the fast loop (shader clock and the 100 MHz real-time counter)
        prof_t0 = stamp();
        prof_r0 = __builtin_amdgcn_s_memrealtime();
    }
    // (re-centring pays on long tiles whose scores spread over enough binades for anything to underflow -- the prologue's `wide`, which also gates
    // the sampled reference: at 1 / sqrt(d) scaling nothing does and the tile keeps its reference, at no cost; 8 is a multiple of G)
    int next_rc = (jf >= 32 && OPT && FA_OPT_SAMPLE && __any(wide)) ? FA_OPT_RC_FIRST : 0x7fffffff;   // (== recentres && any(wide) where the product is concerned)
    // Two workgroups share a CU where the registers allow (NB = 2, d <= 64): two waves per SIMD, and the arbiter serves the OLDER one first --
    // it runs at nearly the speed of a lone wave, finishes early, and the younger one spends the rest of the launch alone at the lower
    // efficiency of one wave per SIMD (stamped kernel, 16 x 8192 d = 32: 760 against 1290 cycles per step, half of the waves each; tiles end
    // after 205k and 342k cycles).  The two take turns instead, by the CLOCK -- a turn counted in own steps lets the favoured wave run ahead just
    // the same --, the wave slot's parity saying whose the even periods are; about four periods per tile (periods of 1k - 4k cycles: half the
    // effect; of 16k and more: tiles end after 290k - 322k cycles, the launch 3.5 % shorter at the same power cap).  Causal launches keep the
    // arbiter's order: their pairs are a heavy and a light tile, and the older one is the heavy one (all periods measured 0.5 - 2 % slower).
    constexpr bool TURNS = FA_XN_TAKE_TURNS != 0 && NB == 2 && D <= 64 && !CAUSAL && (ABL & ~(1024 | 2048)) == 0;
    unsigned turn = 0, turn_bit = 0;
    bool turns = false;
    if constexpr (TURNS) {
        turns = p.take_turns != 0 && nst >= D;   // rows of 2048 (d = 64: 4096) keys and more: below, nothing to gain (128 x 1024: d = 32 -2.9 ... +1.0 %, d = 64 +1.7 ... +2.9 %)
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        turn = hwid & 1u;                                                   // wave slot of the SIMD, bit 0
        turn_bit = min(17, max(12, 31 - __builtin_clz(max(nst, 1) * 600)));  // a stage is two steps of ~1200 cycles with both resident
    }
    for (int j = 0; j < jf; j += G) {
        unsigned long long now = 0;
        if constexpr (TURNS) if (turns) now = __builtin_readcyclecounter();   // (an s_memtime in front of the wait is ~1 % of a lone wave's group: only where it pays)
        sync_top(j);   // (its s_waitcnt lgkmcnt(0) covers the clock read)
        if constexpr (TURNS) {
            if (turns) {
                if ((((unsigned)now >> turn_bit) ^ turn) & 1u) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const char* v_lds = v_slot(j + g);
            const char* k_nxt = k_slot(j + g + 1);
            bool need = xn_step<D, NB, 0, ABL, OPT, PF>(v_lds, k_nxt, 0, k_row_off, k_g, v_lane_off, ones_a, qf, s0, s1, o, st, c, off, kf, lm);
            if (!OPT && __builtin_expect(__any(need), 0)) {
                float mx[NB];
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) mx[blk] = xhalf_max(lm[blk]);
                xn_rescale<NB, DB, PF>(mx, c, st, o, off);
            }
            need = xn_step<D, NB, 1, ABL, OPT, PF>(v_lds, k_nxt, 1, k_row_off, k_g, v_lane_off, ones_a, qf, s1, s0, o, st, c, off, kf, lm);
            if (!OPT && __builtin_expect(__any(need) && 2 * (j + g) + 2 < nsub, 0)) {
                float mx[NB];
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) mx[blk] = xhalf_max(lm[blk]);
                xn_rescale<NB, DB, PF>(mx, c, st, o, off);
            }
        }
        if constexpr (OPT && ABL == 0 && FA_OPT_RECENTRE) {   // after 8, 24, 72, ... stages of a long tile (wave-uniform)
            if (__builtin_expect(j + G == next_rc, 0)) {
                xn_recentre<NB, DB>(st, o, off, (int)kBias);
                next_rc *= 3;
            }
        }
    }

    unsigned long long prof_t1 = 0, prof_r1 = 0;
    if constexpr ((ABL & 1024) != 0) {
        prof_t1 = stamp();
        prof_r1 = __builtin_amdgcn_s_memrealtime();
    }

    // ---------------- tail: remaining stages, phase-structured, masks applied where needed ----------------
    // Invariant at the top of stage j: scores of sub-tile 2j in s0 with the rescale decision taken, kf = fragments of sub-tile 2j+1.
    const int nsub_w = CAUSAL ? min(nsub, (q0r + 32 * (NB - 1) + 31) / 32 + 1) : nsub;
    for (int j = jf; j < nst; ++j) {
        if (j % G == 0) sync_top(j);
        const int t0 = 2 * j, t1 = 2 * j + 1;
        if (t0 < nsub_w) {
            finish_sub(t0, s0);
            if (t1 < nsub_w) {
                qk_regs(t1, s1);
                load_kf(t1 + 1);
                finish_sub(t1, s1);
                if (t1 + 1 < nsub_w) {
                    qk_regs(t1 + 1, s0);
                    load_kf(t1 + 2);
                }
            }
        }
    }

    bool bad = false;
    bool hard = false;   // ... for another reason than accumulators that are EXACTLY zero

    unsigned long long prof_tail = 0;
    if constexpr ((ABL & 1024) != 0) prof_tail = stamp();
    // ---------------- store; verify (optimistic mix) ----------------
    // The optimistic tile stores its result BEFORE the workgroup votes on it: a failed tile is simply overwritten by the redo,
    // and nothing of the first attempt is live across the vote (with the store behind the vote hipcc carried the
    // accumulators of the common path through copies and 12 MB of scratch per launch).
    drain_accumulators<NB, DB>(o, st);  // tied to the accumulators: a bare drain has no data dependence and may be scheduled past
    if (OPT) {
        // every P was exponentiated against the first sub-tile's maximum: the tile stands iff no term left the safe range,
        // which the row sums prove (a term > 2^100, +inf or NaN makes its row sum fail this test)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) bad = bad || !(st[blk].lacc[0] < kOptLimit);
        hard = bad;
    }
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        // (distinct text per mix: identical store code of the two inlined tiles gets tail-merged by hipcc, which then shuffles
        // the accumulators of the common path through copies and 12 MB of scratch per launch)
        if constexpr (OPT) asm volatile("; store, optimistic mix");
        else asm volatile("; store, rescaled mix");
        const float lt = st[blk].lacc[0];
        const float inv = 1.0f / lt;
        const int qi = q0 + 32 * blk + lq;
        if constexpr (OPT) {
            // The optimistic mix keeps P near 2^-kBias, so the accumulators hold ~2^-100 |O| l: products p v of the terms that matter
            // reach fp32's subnormal range when |v| is below ~2^-26 -- and vanish altogether below ~2^-50.  A lane whose accumulators
            // are ALL tiny or zero sends the tile to the rescaled redo (p <= 1 there): V of such magnitudes (and an all-zero V) is
            // computed correctly, twice as slowly -- down to |v| ~ 2^-62: the redo keeps the row maximum within 2^-64 of 1.
            float amax = 0.0f;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; r += 2) amax = fmaxf(fmaxf(amax, fabsf(o[blk][db][r])), fabsf(o[blk][db][r + 1]));
            const bool tiny = amax < tiny_acc && qi < n && !idle;
            bad = bad || tiny;
            hard = hard || (tiny && amax != 0.0f);
        }
        if (qi < n && idle) {   // empty causal key share: only its log-sum-exp (-inf) is stored; the combine never reads its O
            if (p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = -INFINITY;
        } else if (qi < n) {
            if constexpr (OPT) asm volatile("; rows, optimistic mix");
            else asm volatile("; rows, rescaled mix");
            const int64_t o_off = o_slab_off + (int64_t)qi * p.o_row_stride + 4 * hi;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (OUT_F32) {
                        f32x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = o[blk][db][4 * g + e] * inv;
                        *(f32x4*)((float*)p.o + o_off + db * 32 + 8 * g) = pk;
                    } else if ((g & 1) == 0) {
                        // 16-byte stores: the two lanes of a row (hi = 0 / 1) hold alternate 4-column groups; one
                        // v_permlane32_swap per dword hands lane hi = 0 both halves of column group g and lane hi = 1 both
                        // halves of group g + 1 (the epilogue is store-issue bound: half as many, twice as wide)
                        bf16x4 pe, po;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            pe[e] = (__bf16)(o[blk][db][4 * g + e] * inv);
                            po[e] = (__bf16)(o[blk][db][4 * (g + 1) + e] * inv);
                        }
                        const u32x2 ue = __builtin_bit_cast(u32x2, pe), uo = __builtin_bit_cast(u32x2, po);
                        const auto r0 = __builtin_amdgcn_permlane32_swap(ue[0], uo[0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(ue[1], uo[1], false, false);
                        // lanes 0..31: r[0] = own group g, r[1] = partner's group g;  lanes 32..63: r[0] = partner's group g+1, r[1] = own
                        u32x4 w;
                        w[0] = r0[0];
                        w[1] = r1[0];
                        w[2] = r0[1];
                        w[3] = r1[1];
                        // hi = 0: columns 8g .. 8g+7;  hi = 1: columns 8(g+1) .. 8(g+1)+7  (o_off already carries + 4 hi)
                        *(u32x4*)((__bf16*)p.o + o_off - 4 * hi + db * 32 + 8 * (g + hi)) = w;
                    }
                }
            if (!(ABL & 1024) && p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = (st[blk].m + kBias + __builtin_amdgcn_logf(lt)) * kLn2;
        }
    }
    if constexpr ((ABL & 1024) != 0) {
        const unsigned long long t_issued = stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the output stores have left
        const unsigned long long t2 = stamp();
        if (lane == 0 && p.lse != nullptr) {
            float* dst = p.lse + ((int64_t)blockIdx.x * NWAVES + wave) * 8;
            dst[0] = (float)(prof_t1 - prof_t0);      // shader cycles in the fast loop
            dst[1] = (float)(prof_r1 - prof_r0);      // 10 ns ticks in the fast loop
            dst[2] = (float)(2 * jf);                 // steps executed there
            dst[3] = (float)(t2 - prof_entry);        // shader cycles from the first instruction of the tile to its last store
            dst[4] = (float)(prof_landed - prof_entry);   // entry -> Q, K(0..2), V(0..1) landed (issue + HBM round trip, all CUs at once)
            dst[5] = (float)(prof_t0 - prof_landed);      // first scores: K fragments, K.Q^T of sub-tile 0, row maxima, references
            dst[6] = (float)(prof_tail - prof_t1);        // tail stages outside the fast loop
            dst[7] = (float)(t_issued - prof_tail);       // drain, O / l, packing, store issue (t2 - t_issued: stores landing)
            if constexpr ((ABL & 2048) != 0) {   // timeline of the launch (fa_driver_ablation --mode timeline): where and when this workgroup ran
                unsigned hwid, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                unsigned* u = (unsigned*)dst;
                u[0] = (unsigned)prof_rt_in, u[1] = (unsigned)__builtin_amdgcn_s_memrealtime(), u[2] = hwid, u[3] = xcc;
                u[4] = (unsigned)qt, u[5] = (unsigned)slab, u[6] = blockIdx.x, u[7] = (unsigned)nst;
            }
        }
    }
    if (OPT) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // stores and DMA of this attempt done before a redo starts
        if (__syncthreads_or(bad ? 1 : 0)) return __syncthreads_or(hard ? 1 : 0) ? 0 : 2;   // workgroup-wide: the redo shares tiles and barriers
    }
    return 1;
}

// ---- kernels -----------------------------------------------------------------------------------------------------------
// OPTIMISTIC: try the fixed-reference mix first, redo the tile with the rescaling mix if its verification fails.  ABL != 0: the
// timing-only ablations of DESIGN.md section 4 (results are garbage; instantiated in the ablation library only).
template <int D, int NB, int NWAVES, bool CAUSAL, bool OUT_F32, int G, int ABL, bool OPTIMISTIC, int PF = 0>
__device__ __forceinline__ void xn_kernel_body(const FwdParams& p, char* smem)
{
    if (OPTIMISTIC && ABL == 0) {
        const int r = xn_tile<D, NB, NWAVES, CAUSAL, OUT_F32, G, 0, true, PF>(p, smem);
        if (r == 1) return;
        if (r == 2 && bf16_v_is_zero<D, NWAVES>(p)) return;
        count_cliff(p, 0);
    }
    if (ABL != 0 && OPTIMISTIC) {
        (void)xn_tile<D, NB, NWAVES, CAUSAL, OUT_F32, G, ABL, true, PF>(p, smem);
        return;
    }
    (void)xn_tile<D, NB, NWAVES, CAUSAL, OUT_F32, G, ABL, false, PF>(p, smem);
}

template <int NWAVES, bool CAUSAL, bool OUT_F32, int G, int ABL = 0, bool OPTIMISTIC = true>
__global__ __launch_bounds__(NWAVES* kWave, 1) void fa_fwd_bf16_x4_kernel(FwdParams p)
{
    __shared__ __attribute__((aligned(1024))) char smem[4 * G * Bf16Cfg<64, NWAVES>::kTileBytes];
    xn_kernel_body<64, 4, NWAVES, CAUSAL, OUT_F32, G, ABL, OPTIMISTIC>(p, smem);
}
template <int D, int NWAVES, bool CAUSAL, bool OUT_F32, int G, int ABL = 0, bool OPTIMISTIC = true>
__global__ __launch_bounds__(NWAVES* kWave, 1) void fa_fwd_bf16_x2_kernel(FwdParams p)
{
    __shared__ __attribute__((aligned(1024))) char smem[4 * G * Bf16Cfg<D, NWAVES>::kTileBytes];
    xn_kernel_body<D, 2, NWAVES, CAUSAL, OUT_F32, G, ABL, OPTIMISTIC>(p, smem);
}

// The accurate modes: P and V in fp16, lazily rescaled mix only (XSoft<false, PF>).  PF = 1: one fp16 term of P, 11 significant bits
// instead of bf16's 8 ("p16": ~1e-3 of the fp32 reference at scale 1, not guaranteed inside it for large launches).  PF = 2: hi + lo
// fp16 terms ("p16x2": ~22 bits, the kernel FA_KERNEL_AUTO gives a caller who asks for the fp32 accumulator; NB = 2 only, see
// launch_bf16_p16x2 -- the NB = 4 form was built, measured 6 % behind and dropped).  p.v points at the fp16
// copy of V made by launch_cvt_v_f16 (fa_cvt.hip), whose overflow flag these kernels honour (FwdParams::flag_mode = 1).
template <int NWAVES, bool CAUSAL, bool OUT_F32, int G>
__global__ __launch_bounds__(NWAVES* kWave, 1) void fa_fwd_bf16_x4_p16_kernel(FwdParams p)
{
    __shared__ __attribute__((aligned(1024))) char smem[4 * G * Bf16Cfg<64, NWAVES>::kTileBytes];
    if (flag_says_skip(p)) return;
    (void)xn_tile<64, 4, NWAVES, CAUSAL, OUT_F32, G, 0, false, 1>(p, smem);
}
// Causal launches rely on one workgroup per CU (heavy tiles first, light ones behind them on the same CU): when the allocation
// happens to leave room for a second wave per SIMD the hardware co-schedules two heavy tiles and the launch gets 20-30 % slower
// (measured twice: forced with __launch_bounds__(256, 2) on the bf16-P kernel, and by accident when this kernel dropped to 124 + 104
// registers).  amdgpu_waves_per_eu does not change what the hardware co-schedules; an LDS allocation of more than half the CU's
// 160 KB does: see xn_launch_order().  For large causal grids the second resident workgroup is a gain (128 x 8192: 1.22 ms against
// 1.46), hence the bound of two for d <= 64 -- it pins what the allocation gave by accident.
template <int D, int NWAVES, bool CAUSAL, bool OUT_F32, int G>
__global__ __launch_bounds__(NWAVES* kWave, D <= 64 ? 2 : 1) void fa_fwd_bf16_x2_p16_kernel(FwdParams p)
{
    __shared__ __attribute__((aligned(1024))) char smem[4 * G * Bf16Cfg<D, NWAVES>::kTileBytes];
    if (flag_says_skip(p)) return;
    (void)xn_tile<D, 2, NWAVES, CAUSAL, OUT_F32, G, 0, false, 1>(p, smem);
}
template <int D, int NWAVES, bool CAUSAL, bool OUT_F32, int G>
__global__ __launch_bounds__(NWAVES* kWave, D <= 64 ? 2 : 1) void fa_fwd_bf16_x2_p16x2_kernel(FwdParams p)
{
    __shared__ __attribute__((aligned(1024))) char smem[4 * G * Bf16Cfg<D, NWAVES>::kTileBytes];
    if (flag_says_skip(p)) return;
    (void)xn_tile<D, 2, NWAVES, CAUSAL, OUT_F32, G, 0, false, 2>(p, smem);
}

// P as bf16 hi + bf16 lo ("pb2", PF = 3; round 4): the accurate path in ONE launch.  bf16 has fp32's exponent range, so V is used as it
// is (no fp16 copy, no scratch, no overflow flag, no fallback launch), the optimistic mix applies (no maxima, no test, no branch in the
// loop, and the exact zeros far below the sampled reference that the power budget likes) with the lazily rescaled redo behind it, and
// every layout the bf16-P kernels take is legal.  P is good to ~2^-17 (hi: 8 bits to nearest, lo: the next 8 of the exact residual).
template <int NWAVES, bool CAUSAL, bool OUT_F32, int G, int ABL = 0, bool OPTIMISTIC = true>
__global__ __launch_bounds__(NWAVES* kWave, 1) void fa_fwd_bf16_x4_pb2_kernel(FwdParams p)
{
    __shared__ __attribute__((aligned(1024))) char smem[4 * G * Bf16Cfg<64, NWAVES>::kTileBytes];
    xn_kernel_body<64, 4, NWAVES, CAUSAL, OUT_F32, G, ABL, OPTIMISTIC, 3>(p, smem);
}
template <int D, int NWAVES, bool CAUSAL, bool OUT_F32, int G, int ABL = 0, bool OPTIMISTIC = true>
__global__ __launch_bounds__(NWAVES* kWave, pb2_wgs_per_cu(D)) void fa_fwd_bf16_x2_pb2_kernel(FwdParams p)
{
    __shared__ __attribute__((aligned(1024))) char smem[4 * G * Bf16Cfg<D, NWAVES>::kTileBytes];
    xn_kernel_body<D, 2, NWAVES, CAUSAL, OUT_F32, G, ABL, OPTIMISTIC, 3>(p, smem);
}

// ---- launch helpers ----------------------------------------------------------------------------------------------------
// grid: one workgroup per (slab, q tile) of NWAVES * 32 * NB rows; the K / V tiles are addressed with 32-bit byte offsets
template <int NB>
static bool xn_grid(const FwdParams& p0, FwdParams& p, dim3& grid, dim3& block)
{
    constexpr int NWAVES = 4, BM = NWAVES * 32 * NB;
    p = p0;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return false;
    grid = dim3((unsigned)total);
    block = dim3(NWAVES * kWave);
    return true;
}

template <int G, bool OPTIMISTIC, bool CAUSAL>
static hipError_t launch_x4(const FwdParams& p0, int out_f32, hipStream_t stream)
{
    FwdParams p;
    dim3 grid, block;
    if (!xn_grid<4>(p0, p, grid, block)) return hipErrorInvalidValue;
    if (out_f32)
        hipLaunchKernelGGL((fa_fwd_bf16_x4_kernel<4, CAUSAL, true, G, 0, OPTIMISTIC>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_bf16_x4_kernel<4, CAUSAL, false, G, 0, OPTIMISTIC>), grid, block, 0, stream, p);
    return hipGetLastError();
}

// mode 1: barrier every stage; 3: rescaling mix only (no optimistic attempt); otherwise the product tiling
template <bool CAUSAL>
static hipError_t launch_x4_modes(const FwdParams& p, int out_f32, int mode, hipStream_t stream)
{
    if (mode == 1) return launch_x4<1, true, CAUSAL>(p, out_f32, stream);
    if (mode == 3) return launch_x4<2, false, CAUSAL>(p, out_f32, stream);
    return launch_x4<2, true, CAUSAL>(p, out_f32, stream);
}

template <bool CAUSAL>
static hipError_t launch_x4_p16(const FwdParams& p0, int out_f32, hipStream_t stream)
{
    FwdParams p;
    dim3 grid, block;
    if (!xn_grid<4>(p0, p, grid, block)) return hipErrorInvalidValue;
    if (out_f32)
        hipLaunchKernelGGL((fa_fwd_bf16_x4_p16_kernel<4, CAUSAL, true, 2>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_bf16_x4_p16_kernel<4, CAUSAL, false, 2>), grid, block, 0, stream, p);
    return hipGetLastError();
}

// Causal launches of the NB = 2 kernels and the second workgroup of a CU.  The d = 32 kernels (181 registers) and the d = 64 kernels
// (228 - 240 since the reference moves are per block; 264 before, and forcing that version to 256 with __launch_bounds__ made it
// spill into AGPRs inside the loop for no gain) fit twice on a CU and the hardware co-schedules them.  With many tiles per CU, or
// short ones, that hides latency and wins.  With at most two long tiles per CU the whole grid is resident at once and the launch
// lasts as long as the CU with the heaviest PAIR: in slab order, heavy tiles first, that is two heavy tiles (c4-causal fp16 P:
// 0.242 ms).  Two answers, measured on causal 16 x 8192 (ms; fp16 P d = 64 / fp16 P d = 32 / bf16 P d = 32):
//   * one workgroup per CU -- unused dynamic LDS pushes the allocation past half of the CU's 160 KB (amdgpu_waves_per_eu does
//     not change what the hardware co-schedules) -- and the light tiles follow the heavy ones:          0.197 / 0.154 / 0.120
//   * two per CU, odd rounds of an XCD's workgroups walking their slab from the light end (alt_order): 0.181 / 0.140 / 0.109
//     -- for grids of nearly two full rounds; emptier ones take the first answer (xn_launch_order).
// More than two rounds (32 x 8192: 0.343 plain, 0.370 alternating) and short rows (128 x 1024: 0.062 with two per CU, 0.076 with
// one) keep the plain order with two per CU.
template <int D, int G>
static unsigned xn_launch_order(FwdParams& p, const dim3& grid, int causal, bool co_resident)
{
    constexpr int ring = 4 * G * Bf16Cfg<D, 4>::kTileBytes;
    const long wgs = (long)grid.x * grid.y * grid.z;
    p.alt_order = 0;
    // Non-causal, every workgroup resident from the start with a partner of equal work: the pair takes turns at the issue priority (xn_tile).
    // Launches of several rounds keep the arbiter's oldest-first -- staggered pairs overlap one tile's prologue and epilogue with the other's
    // loop, and pairs that end together leave the last partial round alone on its CUs (40 x 8192 d = 32: +12.6 % with turns, 128 x 8192 +2.3 %).
    p.take_turns = (!causal && co_resident && ring < 84 * 1024 && wgs > 256 && wgs <= 2 * 256) ? 1 : 0;
    if (!causal || !co_resident || ring >= 84 * 1024 || wgs > 2 * 256) return 0;
    // Nearly two full rounds: pair the tiles, whatever the row length (short rows too: 128 x 1024 d = 64 0.040 -> 0.034 ms, 64 x 2048
    // 0.065 -> 0.054, d = 32 64 x 2048 0.050 -> 0.039).  A half-filled second round leaves too many heavy tiles without a partner:
    // long rows then take the padded launch (ms paired / padded: 12 x 8192 d = 32 0.127 / 0.114, 5 x 16384 0.220 / 0.196, fp16 P
    // 12 x 8192 0.202 / 0.193; against 20 x 6144 d = 32 0.117 / 0.133, fp16 P 7 x 16384 0.371 / 0.430, 20 x 5000 0.143 / 0.170,
    // 9 x 12288 d = 64 0.249 / 0.301), short rows keep both workgroups and the plain order (40 x 2048: 0.060 against 0.061 padded).
    if (wgs > 384) {
        p.alt_order = 1;         // both workgroups of a CU resident, tiles paired heavy + light (causal_tile)
        return 0;
    }
    if (p.n < 4096) return 0;
    return 84 * 1024 - ring;     // one workgroup per CU, heavy tiles first
}

template <int D, int G, bool OPTIMISTIC = true>
static hipError_t launch_x2(const FwdParams& p0, int causal, int out_f32, hipStream_t stream)
{
    FwdParams p;
    dim3 grid, block;
    if (!xn_grid<2>(p0, p, grid, block)) return hipErrorInvalidValue;
    const unsigned solo = xn_launch_order<D, G>(p, grid, causal, D <= 64);
    auto go = [&](unsigned dyn_lds) {
        if (causal) {
            if (out_f32)
                hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<D, 4, true, true, G, 0, OPTIMISTIC>), grid, block, dyn_lds, stream, p);
            else
                hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<D, 4, true, false, G, 0, OPTIMISTIC>), grid, block, dyn_lds, stream, p);
        } else {
            if (out_f32)
                hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<D, 4, false, true, G, 0, OPTIMISTIC>), grid, block, 0, stream, p);
            else
                hipLaunchKernelGGL((fa_fwd_bf16_x2_kernel<D, 4, false, false, G, 0, OPTIMISTIC>), grid, block, 0, stream, p);
        }
        return hipGetLastError();
    };
    hipError_t e = go(solo);
    if (e != hipSuccess && solo != 0) e = go(0);   // the padding is an optimisation: a runtime that refuses it still gets the launch
    return e;
}

template <int D, int PF = 1>
static hipError_t launch_x2_p16(const FwdParams& p0, int causal, int out_f32, hipStream_t stream)
{
    FwdParams p;
    dim3 grid, block;
    if (!xn_grid<2>(p0, p, grid, block)) return hipErrorInvalidValue;
    const unsigned solo = xn_launch_order<D, 2>(p, grid, causal, D <= 64);
    auto go = [&](unsigned dyn_lds) {
        if constexpr (PF == 2) {
            if (causal) {
                if (out_f32)
                    hipLaunchKernelGGL((fa_fwd_bf16_x2_p16x2_kernel<D, 4, true, true, 2>), grid, block, dyn_lds, stream, p);
                else
                    hipLaunchKernelGGL((fa_fwd_bf16_x2_p16x2_kernel<D, 4, true, false, 2>), grid, block, dyn_lds, stream, p);
            } else {
                if (out_f32)
                    hipLaunchKernelGGL((fa_fwd_bf16_x2_p16x2_kernel<D, 4, false, true, 2>), grid, block, 0, stream, p);
                else
                    hipLaunchKernelGGL((fa_fwd_bf16_x2_p16x2_kernel<D, 4, false, false, 2>), grid, block, 0, stream, p);
            }
        } else if (causal) {
            if (out_f32)
                hipLaunchKernelGGL((fa_fwd_bf16_x2_p16_kernel<D, 4, true, true, 2>), grid, block, dyn_lds, stream, p);
            else
                hipLaunchKernelGGL((fa_fwd_bf16_x2_p16_kernel<D, 4, true, false, 2>), grid, block, dyn_lds, stream, p);
        } else {
            if (out_f32)
                hipLaunchKernelGGL((fa_fwd_bf16_x2_p16_kernel<D, 4, false, true, 2>), grid, block, 0, stream, p);
            else
                hipLaunchKernelGGL((fa_fwd_bf16_x2_p16_kernel<D, 4, false, false, 2>), grid, block, 0, stream, p);
        }
        return hipGetLastError();
    };
    hipError_t e = go(solo);
    if (e != hipSuccess && solo != 0) e = go(0);   // see launch_x2
    return e;
}

// (one output type per translation unit: the two-term schedules are the longest compiles of the library)
template <bool CAUSAL, bool OUT_F32, bool OPTIMISTIC = true>
static hipError_t launch_x4_pb2(const FwdParams& p0, hipStream_t stream)
{
    FwdParams p;
    dim3 grid, block;
    if (!xn_grid<4>(p0, p, grid, block)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((fa_fwd_bf16_x4_pb2_kernel<4, CAUSAL, OUT_F32, 2, 0, OPTIMISTIC>), grid, block, 0, stream, p);
    return hipGetLastError();
}
template <int D, bool OUT_F32, bool OPTIMISTIC = true>
static hipError_t launch_x2_pb2(const FwdParams& p0, int causal, hipStream_t stream)
{
    FwdParams p;
    dim3 grid, block;
    if (!xn_grid<2>(p0, p, grid, block)) return hipErrorInvalidValue;
    const unsigned solo = xn_launch_order<D, 2>(p, grid, causal, D <= 64);
    auto go = [&](unsigned dyn_lds) {
        if (causal) hipLaunchKernelGGL((fa_fwd_bf16_x2_pb2_kernel<D, 4, true, OUT_F32, 2, 0, OPTIMISTIC>), grid, block, dyn_lds, stream, p);
        else hipLaunchKernelGGL((fa_fwd_bf16_x2_pb2_kernel<D, 4, false, OUT_F32, 2, 0, OPTIMISTIC>), grid, block, 0, stream, p);
        return hipGetLastError();
    };
    hipError_t e = go(solo);
    if (e != hipSuccess && solo != 0) e = go(0);   // see launch_x2
    return e;
}

template <int ABL>
static hipError_t launch_x4_ablation(const FwdParams& p0, hipStream_t stream)
{
    FwdParams p;
    dim3 grid, block;
    if (!xn_grid<4>(p0, p, grid, block)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((fa_fwd_bf16_x4_kernel<4, false, false, 2, ABL>), grid, block, 0, stream, p);
    return hipGetLastError();
}

// the 32-bit slab addressing of the LDS-DMA descriptors (TileDma)
static inline bool xn_addressable(const FwdParams& p, int d)
{
    return ((int64_t)(p.n - 1) * p.kv_row_stride + d) * 2 < (int64_t)0xffffffffLL;
}

// timing-only ablations (results are garbage), defined in fa_fwd_bf16_x4_ablation.hip (ablation library only)
hipError_t launch_bf16_x4_ablation(const FwdParams& p, int mode, hipStream_t stream);
// the causal instantiations, defined in fa_fwd_bf16_x4_causal.hip
hipError_t launch_bf16_x4_causal(const FwdParams& p, int out_f32, int mode, hipStream_t stream);

}  // namespace fa
