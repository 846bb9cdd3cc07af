// fa_f32_t3_kernel.h -- fp32 tensors on the bf16 matrix pipe, three products per contraction, in the one-wave-per-SIMD /
// explicit-register-file / static-slot construction of fa_bf16_xn_kernel.h.  Long non-causal rows at head dim 64 (config c3).
//
// Same arithmetic as fa_split_kernel.h (x = x_hi + x_lo, two bf16 terms; a.b ~= a_hi.b_hi + a_lo.b_hi + a_hi.b_lo, fp32 accumulate),
// different machinery.  The split kernel converts every K / V tile in every workgroup (global -> registers -> VALU -> four LDS
// images) and leaves its schedule to hipcc (matrix pipe 60 % busy at c3).  Here
//   * K and V are split ONCE per launch by a pre-pass (fa_cvt.hip: four dense bf16 arrays K_hi, K_lo, V_hi, V_lo in stream-ordered
//     scratch; the same pass bounds max |k| and max |q|_2 for the logit-width guard), so the tiles arrive by LDS-DMA like in the
//     bf16 kernels -- no VALU, no VGPR round trip;
//   * every MFMA is asm with its register files chosen (scores -> VGPRs, O and the Q' fragments -> AGPRs) and sits in a static slot
//     with its share of the VALU work: per 32-key step and wave 48 MFMAs (24 K.Q'^T, 24 P.V) against 32 exp + 32 adds + 16
//     hi/lo splits of P -- the intent was a loop bound by the matrix pipe; measured, it is bound by instruction issue (see below);
//   * the softmax is the reference-free optimistic one of the split kernel's fast pass: p = exp2(s) with s already in the exp2
//     domain (Q' = Q * scale * log2 e), fp32 row sums on the VALU.  A row sum outside (2^-100, 2^100) or a non-finite output RAISES
//     THE LAUNCH CHAIN'S FLAG instead of being redone here: the exact fp32 kernel queued behind recomputes the launch (fa_launch.cpp).
// Scope (everything else stays with fa_split_kernel.h): head dim 64, non-causal, N a multiple of 64, plain (BH, N, d) layout.
//
//   step t (32 keys, 48 MFMA slots):  K.Q'^T of sub-tile t+1: 4 k-steps x 3 terms x blocks A, B (24)  |  P.V of A: 12  |  P.V of B: 12
//   V^T fragments (hi and lo) read in slots 0..7, waited for in front of slot 24; K_lo fragments of THIS step's scores read in slots 0..3
//   (waited for in front of slot 4), K_hi fragments of step t+1 in slots 24..27.
#pragma once
#include "fa_bf16_xn_kernel.h"
#include "fa_split_kernel.h"   // split2 / split8: the two-term bf16 split

namespace fa {

constexpr float kT3Limit = 0x1p100f;

// ---- static schedule ---------------------------------------------------------------------------------------------------------
struct T3Slot {
    int kind;  // 0 = K.Q'^T, 1 = P.V
    int blk, idx, term;   // K.Q'^T: idx = k-step, term 0: K_hi.Q_lo, 1: K_hi.Q_hi, 2: K_lo.Q_hi;  P.V: idx = V^T fragment v = tt * 2 + db, term 0: V_lo.P_hi, 1: V_hi.P_lo, 2: V_hi.P_hi
};
__device__ __host__ constexpr T3Slot t3_slot(int i)
{
    if (i < 24) return {0, i % 2, i / 6, (i % 6) / 2};
    const int blk = (i - 24) / 12, j = (i - 24) % 12;
    const int pair = j / 6, r = j % 6;   // two V^T fragments (db = 0, 1) alternate: a dependent MFMA never follows its producer directly
    return {1, blk, pair * 2 + r % 2, r / 2};
}
constexpr int kT3Slots = 48;
// VALU work as single instructions ("micro-ops"), dealt over the MFMA slots by issue cost.  Per pair q of P values of a block (scores
// 2q, 2q+1 -> p0, p1 -> hi / lo bf16 pairs): E0 E1 (v_exp_f32), A0 A1 (row-sum adds), H (v_cvt_pk_bf16_f32: the hi pair), U0 U1 (hi back
// to fp32: shift / mask), D0 D1 (p - hi), L (v_cvt_pk_bf16_f32: the lo pair).  Software-pipelined over the pairs so that no instruction
// but L follows its producer directly:
//     E0(q) E1(q) U0(q-1) U1(q-1) A0(q) A1(q) H(q) D0(q-1) D1(q-1) L(q-1)
// Each is one asm volatile statement: the order below IS the issue order.
//
// What this construction showed (profiles/ubench/ubench_valu_mix.hip, profiles/r02_ubench_valu_mix.txt): with one wave per SIMD an
// instruction of any kind issues every ~5.3 cycles (v_exp_f32 9.1, v_cvt_pk_bf16_f32 8.1, even s_nop 4.7), an MFMA costs ~11 cycles of
// issue on top of its 32 cycles of pipe, and packed fp32 adds do not overlap an MFMA at all (4 v_pk_add_f32 beside one MFMA: 62 cycles).
// Per step that is 16 pairs x 67 cycles of split/exp work + 48 x 11 cycles of MFMA issue + ~240 cycles of LDS reads and addresses + the
// s_nop hipcc pads between dependent asm statements (~40 x 4.7) ~= 2000-2100 cycles against 1536 cycles of matrix pipe: THE LOOP IS
// BOUND BY INSTRUCTION ISSUE, NOT BY THE PIPE, and lands where the compiler-scheduled split kernel (several waves per SIMD) already
// is.  Variants measured on the way, all correct (2.4e-4 at c3), relative to the split kernel on the same box: units of 12-24 cycles
// as builtins 1.04x its time; this form 0.98x; score blocks pinned to fixed registers with packed adds in two-instruction chunks
// 1.07x; the same in four-instruction chunks 1.07x.
enum T3Op { kE0, kE1, kA0, kA1, kH, kU0, kU1, kD0, kD1, kL };
struct T3Unit {
    int op, blk, q, cost;
};
constexpr int kT3Units = 2 * 8 * 10;
struct T3UnitList {
    T3Unit u[kT3Units];
};
__device__ __host__ constexpr T3UnitList t3_make_units()
{
    T3UnitList l{};
    int n = 0;
    for (int b = 0; b < 2; ++b) {
        for (int q = 0; q <= 8; ++q) {
            if (q < 8) {
                l.u[n++] = {kE0, b, q, 9};
                l.u[n++] = {kE1, b, q, 9};
            }
            if (q > 0) {
                l.u[n++] = {kU0, b, q - 1, 5};
                l.u[n++] = {kU1, b, q - 1, 5};
            }
            if (q < 8) {
                l.u[n++] = {kA0, b, q, 5};
                l.u[n++] = {kA1, b, q, 5};
                l.u[n++] = {kH, b, q, 8};
            }
            if (q > 0) {
                l.u[n++] = {kD0, b, q - 1, 5};
                l.u[n++] = {kD1, b, q - 1, 5};
                l.u[n++] = {kL, b, q - 1, 8};
            }
        }
    }
    return l;
}
constexpr int kT3Wend = 40;   // block B's second P fragment is first read in slot 42, its first in slot 36 (checked below)
struct T3Table {
    int ub[kT3Slots + 1];
};
__device__ __host__ constexpr T3Table t3_make_table()
{
    const T3UnitList l = t3_make_units();
    T3Table t{};
    int total = 0;
    for (int u = 0; u < kT3Units; ++u) total += l.u[u].cost;
    int n = 0, cum_next = l.u[0].cost;
    for (int i = 0; i <= kT3Slots; ++i) {
        const int wb = i < kT3Wend ? i : kT3Wend;
        const int target = total * wb / kT3Wend + 2;
        while (n < kT3Units && cum_next <= target) {
            ++n;
            if (n < kT3Units) cum_next += l.u[n].cost;
        }
        t.ub[i] = n;
    }
    t.ub[kT3Slots] = kT3Units;
    return t;
}
// the micro-op that completes P fragment f (8 values = pairs 4f .. 4f+3) of block b: the L of its last pair
__device__ __host__ constexpr int t3_frag_done_unit(int b, int f)
{
    const T3UnitList l = t3_make_units();
    int last = -1;
    for (int u = 0; u < kT3Units; ++u)
        if (l.u[u].op == kL && l.u[u].blk == b && l.u[u].q / 4 == f) last = u;
    return last;
}
// static check of the schedule: every P fragment is complete at least one whole slot before the first MFMA that reads it (a VALU result
// needs two wait states before an MFMA may read it, and hipcc pads nothing in front of an asm MFMA)
__device__ __host__ constexpr bool t3_schedule_ok()
{
    const T3Table t = t3_make_table();
    for (int i = 24; i < kT3Slots; ++i) {
        const T3Slot s = t3_slot(i);
        if (t3_frag_done_unit(s.blk, s.idx / 2) >= t.ub[i - 1]) return false;
    }
    return true;
}
static_assert(t3_schedule_ok(), "a P fragment is completed too late for its first P.V slot");

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
struct T3Ctx {
    const bf16x8 (&qh)[2][4];
    const bf16x8 (&ql)[2][4];
    f32x16 (&sc)[2];
    f32x16 (&sn)[2];
    f32x16 (&o)[2][2];
    f32x2_t (&lsum)[2];   // the two running row sums of a block (even / odd scores), as one packed pair
    bf16x8 (&kfh)[4];
    bf16x8 (&kfl)[4];
    const char* kh_nxt;   // K_hi tile the (hi) fragments of the step after next come from
    const char* kl_cur;   // K_lo tile of the sub-tile whose scores THIS step produces (its fragments are read in slots 0..3, used from slot 4 on)
    int kb_n2, kb_cur, k_row_off, k_g;
    unsigned vh_addr, vl_addr;
    s16x4 vlo[8], vhi[8];   // fragments 0..3: V_hi, 4..7: V_lo
    bf16x8 vf[8];
    u32x4_t ph[2][2], pl[2][2];   // P_hi / P_lo fragments, one packed bf16 pair per element
    // vector types, not arrays: element access at a constant index is then a register by construction (arrays of a struct that is
    // passed by reference through the fold expressions were left in scratch memory)
    f32x16 t[2];          // hi back in fp32, then p - hi (between U and L): elements 2q, 2q+1
};

template <int U>
__device__ __forceinline__ void t3_unit(T3Ctx& x)
{
    constexpr T3Unit un = t3_make_units().u[U];
    constexpr int b = un.blk, q = un.q, f = q / 4, e = q % 4;
    if constexpr (un.op == kE0) asm volatile("v_exp_f32 %0, %0" : "+v"(x.sc[b][2 * q]));
    else if constexpr (un.op == kE1) asm volatile("v_exp_f32 %0, %0" : "+v"(x.sc[b][2 * q + 1]));
    else if constexpr (un.op == kA0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.lsum[b][0]) : "v"(x.sc[b][2 * q]));
    else if constexpr (un.op == kA1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x.lsum[b][1]) : "v"(x.sc[b][2 * q + 1]));
    else if constexpr (un.op == kH) {
        unsigned h;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x.sc[b][2 * q]), "v"(x.sc[b][2 * q + 1]));
        x.ph[b][f][e] = h;
    } else if constexpr (un.op == kU0) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(x.t[b][2 * q]) : "v"(x.ph[b][f][e]));
    else if constexpr (un.op == kU1) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(x.t[b][2 * q + 1]) : "v"(x.ph[b][f][e]));
    else if constexpr (un.op == kD0) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x.t[b][2 * q]) : "v"(x.sc[b][2 * q]));
    else if constexpr (un.op == kD1) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(x.t[b][2 * q + 1]) : "v"(x.sc[b][2 * q + 1]));
    else {
        unsigned l;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l) : "v"(x.t[b][2 * q]), "v"(x.t[b][2 * q + 1]));
        x.pl[b][f][e] = l;
    }
}
template <int U0, int... Us>
__device__ __forceinline__ void t3_units(T3Ctx& x, std::integer_sequence<int, Us...>)
{
    (t3_unit<U0 + Us>(x), ...);
}

template <int KB_C, int I, int ABL>
__device__ __forceinline__ void t3_slot_body(T3Ctx& x)
{
    constexpr int D = 64;
    constexpr T3Slot sl = t3_slot(I);
    constexpr T3Table tab = t3_make_table();
    if constexpr (I == 24) {   // all sixteen V^T reads were issued in slots 0..7: one wait orders them (the K reads start behind it)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(x.vlo[0]), "+v"(x.vhi[0]), "+v"(x.vlo[1]), "+v"(x.vhi[1]), "+v"(x.vlo[2]), "+v"(x.vhi[2]), "+v"(x.vlo[3]), "+v"(x.vhi[3]));
        asm volatile("" : "+v"(x.vlo[4]), "+v"(x.vhi[4]), "+v"(x.vlo[5]), "+v"(x.vhi[5]), "+v"(x.vlo[6]), "+v"(x.vhi[6]), "+v"(x.vlo[7]), "+v"(x.vhi[7]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 8; ++v) x.vf[v] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(x.vlo[v], x.vhi[v], 0, 1, 2, 3, 4, 5, 6, 7));
    }
    if constexpr (ABL & 4) { if constexpr (I < 8) asm volatile("" : "=v"(x.vlo[I]), "=v"(x.vhi[I])); }
    else if constexpr (I < 4) {
        load_v_frag_asm<D, KB_C, I>(x.vh_addr, x.vlo[I], x.vhi[I]);
        const unsigned a = (unsigned)(size_t)(lds_s16x4_t*)(x.kl_cur + x.k_row_off + x.kb_cur * 32 * (2 * D) + (((2 * I) ^ x.k_g) * 16));
        asm volatile("ds_read_b128 %0, %1" : "=v"(x.kfl[I]) : "v"(a));
    } else if constexpr (I < 8) {
        load_v_frag_asm<D, KB_C, I - 4>(x.vl_addr, x.vlo[I], x.vhi[I]);
    }
    // a VALU result needs two wait states before an MFMA reads it; the schedule check above keeps a whole slot between a split and its
    // first reader, so no padding is needed here
    if constexpr (ABL & 1) {
    } else if constexpr (sl.kind == 0) {
        const bf16x8& a = sl.term == 2 ? x.kfl[sl.idx] : x.kfh[sl.idx];
        const bf16x8& b = sl.term == 0 ? x.ql[sl.blk][sl.idx] : x.qh[sl.blk][sl.idx];
        if constexpr (I == 4 && !(ABL & 4))   // the K_lo fragments were read in slots 0..3 (LDS returns in order; this slot's two V^T reads are behind them)
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x.kfl[0]), "+v"(x.kfl[1]), "+v"(x.kfl[2]), "+v"(x.kfl[3]));
        if constexpr (sl.idx == 0 && sl.term == 0) mfma_s_first(x.sn[sl.blk], a, b);
        else mfma_s(x.sn[sl.blk], a, b);
    } else {
        constexpr int v = sl.idx, f = v / 2, db = v % 2;
        const bf16x8& a = sl.term == 0 ? x.vf[4 + v] : x.vf[v];
        const bf16x8 b = __builtin_bit_cast(bf16x8, sl.term == 1 ? x.pl[sl.blk][f] : x.ph[sl.blk][f]);
        mfma_o<false>(x.o[sl.blk][db], a, b);
    }
    if constexpr (I >= 24 && I < 28) {   // K_hi fragments of the step after next (K_lo: read by that step itself, slots 0..3)
        constexpr int ks = I - 24;
        const unsigned a = (unsigned)(size_t)(lds_s16x4_t*)(x.kh_nxt + x.k_row_off + x.kb_n2 * 32 * (2 * D) + (((2 * ks) ^ x.k_g) * 16));
        if constexpr (ABL & 4) asm volatile("" : "+v"(x.kfh[ks]) : "v"(a));
        else asm volatile("ds_read_b128 %0, %1" : "=v"(x.kfh[ks]) : "v"(a));
    }
    if constexpr (!(ABL & 2)) t3_units<tab.ub[I]>(x, std::make_integer_sequence<int, tab.ub[I + 1] - tab.ub[I]>{});
    __builtin_amdgcn_sched_barrier(0);
}
template <int KB_C, int ABL, int... Is>
__device__ __forceinline__ void t3_slots(T3Ctx& x, std::integer_sequence<int, Is...>)
{
    (t3_slot_body<KB_C, Is, ABL>(x), ...);
}

// lo_off_k / lo_off_v: byte distance from the K_hi (V_hi) ring to the K_lo (V_lo) ring
template <int KB_C, int ABL = 0>
__device__ __forceinline__ void t3_step(const char* vh_lds, int lo_off_v, const char* kh_nxt, const char* kl_cur, int kb_cur, int kb_n2, int k_row_off, int k_g, int v_lane_off,
                                        const bf16x8 (&qh)[2][4], const bf16x8 (&ql)[2][4], f32x16 (&sc)[2], f32x16 (&sn)[2], f32x16 (&o)[2][2],
                                        f32x2_t (&lsum)[2], bf16x8 (&kfh)[4], bf16x8 (&kfl)[4])
{
    T3Ctx x{qh, ql, sc, sn, o, lsum, kfh, kfl, kh_nxt, kl_cur, kb_n2, kb_cur, k_row_off, k_g,
            (unsigned)(size_t)(lds_s16x4_t*)(vh_lds + v_lane_off), (unsigned)(size_t)(lds_s16x4_t*)(vh_lds + lo_off_v + v_lane_off)};
    t3_slots<KB_C, ABL>(x, std::make_integer_sequence<int, kT3Slots>{});
    // the K reads are sixteen slots old: this wait is free, and it keeps every asm-issued load inside the basic block that issued it
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kfh[0]), "+v"(kfh[1]), "+v"(kfh[2]), "+v"(kfh[3]));
}

// p.k / p.v: K_hi / V_hi (dense (bh, n, 64) bf16); p.k_lo / p.v_lo: the low terms; p.stats: {max |k| word, max |q'|^2 word} of the pre-pass
template <int G, int ABL = 0>
__global__ __launch_bounds__(256, 1) void fa_fwd_f32_t3_kernel(FwdParams p)
{
    constexpr int D = 64, NB = 2, KS = 4, NWAVES = 4;
    using C = Bf16Cfg<D, NWAVES>;
    constexpr int T = C::kTileBytes, R = 2 * G, BM = NWAVES * 32 * NB;
    static_assert(G == 2, "ring arithmetic written for a barrier every two stages");
    // K_hi | K_lo | V_hi | V_lo rings.  The K_lo fragments are read by the step that uses them (sixteen VGPRs less across the P.V phase), so a
    // K_lo tile lives one step longer than its K_hi tile -- into the stage at whose top the DMA of the tile 2G stages ahead is enqueued:
    // the K_lo ring has 2R slots.  5 R T = 160 KiB: the whole LDS of the CU.
    __shared__ __attribute__((aligned(1024))) char smem[5 * R * T];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;

    // logit-width guard, decided from the pre-pass maxima (tagged with this call's serial number in their upper halves): too wide for
    // 16-bit operand terms -> raise the flag and leave the launch to the exact kernel queued behind
    if (p.flag_mode == 3) {
        const unsigned long long kw = p.stats[0], qw = p.stats[1];
        const float kmax = (unsigned)(kw >> 32) == p.flag_serial ? __uint_as_float((unsigned)kw) : INFINITY;
        const float qn2 = (unsigned)(qw >> 32) == p.flag_serial ? __uint_as_float((unsigned)qw) : INFINITY;
        if (!(sqrtf(qn2) * kmax <= 100.0f * kLog2e)) {   // (kGuardLimit of fa_split_kernel.h as it was in round 3; false for NaN as well)
            if (threadIdx.x == 0) __hip_atomic_store(p.flag, p.flag_serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }

    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles, qt = w % p.q_tiles;
    const int n = p.n;
    const int q0 = qt * BM + wave * 32 * NB;
    const float* qg = (const float*)p.q + (int64_t)slab * p.q_batch_stride;
    const int64_t kv_off = (int64_t)slab * n * D;
    const __bf16* khg = (const __bf16*)p.k + kv_off;
    const __bf16* klg = (const __bf16*)p.k_lo + kv_off;
    const __bf16* vhg = (const __bf16*)p.v + kv_off;
    const __bf16* vlg = (const __bf16*)p.v_lo + kv_off;
    const int64_t o_slab_off = (int64_t)slab * p.o_batch_stride;
    const int nst = n / kKvBlk;   // n is a multiple of 64 (checked by the launcher)

    char* const kh_ring = smem;
    char* const kl_ring = smem + R * T;
    constexpr int LO_V = R * T;   // V_lo ring right behind V_hi
    char* const vh_ring = smem + 3 * R * T;
    auto kh_slot = [&](int j) { return kh_ring + (j & (R - 1)) * T; };
    auto kl_slot = [&](int j) { return kl_ring + (j & (2 * R - 1)) * T; };
    auto vh_slot = [&](int j) { return vh_ring + (j & (R - 1)) * T; };

    TileDma<D, NWAVES> dh, dl;
    dh.init(khg, vhg, n, D, wave, lane);
    dl.init(klg, vlg, n, D, wave, lane);
    auto issue_k = [&](int j) {
        dh.issue_k((unsigned)j * dh.stage_step, kh_slot(j), wave);
        dl.issue_k((unsigned)j * dl.stage_step, kl_slot(j), wave);
    };
    auto issue_v = [&](int j) {
        dh.issue_v((unsigned)j * dh.stage_step, vh_slot(j), wave);
        dl.issue_v((unsigned)j * dl.stage_step, vh_slot(j) + LO_V, wave);
    };
    issue_k(0);
#pragma unroll
    for (int g = 1; g <= G; ++g)
        if (g < nst) issue_k(g);
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (g < nst) issue_v(g);

    // Q' = Q * scale * log2(e), split into hi + lo: lane (lq, hi) holds Q'[row][16 ks + 8 hi .. + 7]
    bf16x8 qh[NB][KS], ql[NB][KS];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const float* qr = qg + (int64_t)min(q0 + 32 * blk + lq, n - 1) * p.q_row_stride + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f32x4 a = *(const f32x4*)(qr + ks * 16) * p.scale_log2e, c = *(const f32x4*)(qr + ks * 16 + 4) * p.scale_log2e;
            split8(a, c, qh[blk][ks], ql[blk][ks]);
        }
    }

    // The Q' fragments are B operands in AGPRs for the whole kernel.  hipcc would copy them there (v_accvgpr_write) right in front of
    // the first asm MFMA that names them -- a VALU write -> MFMA read hazard it cannot see inside an asm statement (the prologue's
    // first scores came out as garbage).  Pin them into AGPRs here, with the wait states behind.
    asm volatile("s_nop 4" : "+a"(qh[0][0]), "+a"(qh[0][1]), "+a"(qh[0][2]), "+a"(qh[0][3]), "+a"(qh[1][0]), "+a"(qh[1][1]), "+a"(qh[1][2]), "+a"(qh[1][3]));
    asm volatile("s_nop 4" : "+a"(ql[0][0]), "+a"(ql[0][1]), "+a"(ql[0][2]), "+a"(ql[0][3]), "+a"(ql[1][0]), "+a"(ql[1][1]), "+a"(ql[1][2]), "+a"(ql[1][3]));

    f32x16 o[NB][2], s0[NB], s1[NB];
    f32x2_t lsum[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        lsum[blk] = f32x2_t{0.0f, 0.0f};
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[blk][db][r] = 0.0f;
    }

    const int k_row_off = lq * C::kRowBytes;
    const int k_g = hi ^ k_swizzle<D>(lq);
    const int li = lane & 15;
    const int v_lane_off = (hi * (D / 16) + ((lane >> 4) & 1)) * 128 + (li >> 2) * 32 + (li & 3) * 8;

    auto sync_top = [&](int j) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (ABL & 8) return;
#pragma unroll
        for (int g = 1; g <= G; ++g)
            if (__builtin_expect(j + G + g < nst, 1)) issue_k(j + G + g);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (__builtin_expect(j + G + g < nst, 1)) issue_v(j + G + g);
    };
    bf16x8 kfh[KS], kfl[KS];
    auto load_kf = [&](int t, bool with_lo) {
        const char* k_lds = kh_slot(t >> 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            kfh[ks] = load_k_frag<D>(k_lds, k_row_off, k_g, t & 1, ks);
            if (with_lo) kfl[ks] = load_k_frag<D>(kl_slot(t >> 1), k_row_off, k_g, t & 1, ks);
        }
    };

    // ---------------- prologue: K(0) landed -> scores of sub-tile 0, fragments of sub-tile 1 ----------------
    wait_lds_dma();
    __syncthreads();
    load_kf(0, true);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                const bf16x8& a = term == 2 ? kfl[ks] : kfh[ks];
                const bf16x8& b = term == 0 ? ql[blk][ks] : qh[blk][ks];
                if (ks == 0 && term == 0) mfma_s_first(s0[blk], a, b);
                else mfma_s(s0[blk], a, b);
            }
    drain_scores<NB>(s0);
    load_kf(1, false);   // K_hi of sub-tile 1; the first step reads its K_lo itself

    // ---------------- main loop: groups of G stages, no masks (N is a multiple of 64, non-causal) ----------------
    // Whole groups first, without a branch around the steps: a conditional step puts a second phi between the accumulators' asm
    // definitions and the loop-header phi, the header phi then stays a VGPR phi and the compiler copies all 64 accumulators out of
    // and back into AGPRs every iteration (128 v_accvgpr moves per four steps).
    auto stage = [&](int jg) {
        const char* vh_lds = vh_slot(jg);
        const char* kh_nxt = kh_slot(jg + 1);
        // step 2 jg scores sub-tile 2 jg + 1 (this stage, second half), step 2 jg + 1 scores the first half of the next stage
        t3_step<0, ABL>(vh_lds, LO_V, kh_nxt, kl_slot(jg), 1, 0, k_row_off, k_g, v_lane_off, qh, ql, s0, s1, o, lsum, kfh, kfl);
        t3_step<1, ABL>(vh_lds, LO_V, kh_nxt, kl_slot(jg + 1), 0, 1, k_row_off, k_g, v_lane_off, qh, ql, s1, s0, o, lsum, kfh, kfl);
    };
    int j = 0;
    for (; j + G <= nst; j += G) {
        sync_top(j);
#pragma unroll
        for (int g = 0; g < G; ++g) stage(j + g);
    }
    if (j < nst) {   // an odd stage count: one more stage (G = 2)
        sync_top(j);
        stage(j);
    }

    // ---------------- epilogue: O / l, store; a row outside the provable range raises the chain's flag ----------------
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(o[0][0]), "+a"(o[0][1]), "+a"(o[1][0]), "+a"(o[1][1]));
    bool ok = true;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const float lt = xhalf_sum(lsum[blk][0] + lsum[blk][1]);
        const float inv = 1.0f / lt;
        const int qi = q0 + 32 * blk + lq;
        float mag = 0.0f;
        if (qi < n) {
            const int64_t o_off = o_slab_off + (int64_t)qi * p.o_row_stride + 4 * hi;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        pk[e] = o[blk][db][4 * g + e] * inv;
                        mag += fabsf(pk[e]);
                    }
                    *(f32x4*)((float*)p.o + o_off + db * 32 + 8 * g) = pk;
                }
            if (p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = __builtin_amdgcn_logf(lt) * kLn2;
            ok = ok && (lt > 1.0f / kT3Limit) && (lt < kT3Limit) && (mag < INFINITY);   // false for NaN as well
        }
    }
    if (!ok && p.flag != nullptr) __hip_atomic_store(p.flag, p.flag_serial, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace fa
