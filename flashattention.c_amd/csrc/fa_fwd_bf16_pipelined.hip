// fa_fwd_bf16_pipelined.hip -- the bf16 kernel that ships for D = 64 (and non-causal D = 32): two 32-row blocks per wave in
// lockstep over 32-key sub-tiles, software-pipelined inside one instruction stream ("pp3").  Design notes in the block
// comment below and in DESIGN.md section 3.
#include "fa_bf16_step.h"
#include "fa_kernels.h"

namespace fa {

// =====================================================================================================================
// Third generation ("pp3"): the two 32-row blocks of a wave walk the keys in LOCKSTEP over 32-key sub-tiles, so every K
// and V^T fragment fetched from LDS feeds two MFMAs (ablation on MI355X: halving the fragment reads of pp2 is worth 15 %).
// MFMA/VALU overlap now comes from software pipelining across sub-tiles inside the single instruction stream:
//
//   step t:   Q  phase   K.Q^T of sub-tile t+1 for A and B (2*KS MFMAs)     ||  exp + pack of block A, sub-tile t
//             P1 phase   P.V + row sums of block A, sub-tile t             ||  exp + pack of block B, sub-tile t
//             P2 phase   P.V + row sums of block B (V^T fragments reused)  ||  row maxima of sub-tile t+1, rescale decision
//
// about 6 VALU instructions per MFMA in every phase.  Scores live in two register buffers per block (s0/s1, swapped every
// step; a 64-key stage = two explicitly unrolled steps).  K ring 3 stages, V ring 2 stages, one barrier per 64 keys.
// =====================================================================================================================
template <int D>
struct Plan3 {
    static constexpr int KS = D / 16, DB = D / 32;
    static constexpr int NV = 2 * DB;  // V^T fragments per 32-key sub-tile: (16-key step tt, 32-col block db)
};

struct Lazy2 {  // exponent offsets in use for the two blocks
    float offa, offb;
};

// decision for both blocks at once (one rare wave-uniform branch per step)
template <int D>
__device__ __forceinline__ void lazy_rescale2(float mxa, float mxb, float c, BlockState& sta, BlockState& stb, f32x16 (&oa)[D / 32],
                                              f32x16 (&ob)[D / 32], Lazy2& lz)
{
    float mca = mxa * c, mcb = mxb * c;
    mca = fmaf(-fabsf(mca), 0x1p-23f, mca);
    mcb = fmaf(-fabsf(mcb), 0x1p-23f, mcb);
    if (__builtin_expect(__any((mca - sta.m > kLazyThr) || (mcb - stb.m > kLazyThr)), 0)) {
        asm volatile("; lazy rescale (both blocks)" ::: "memory");
        mfma_drain();  // the accumulators rescaled below may have an MFMA in flight (hazard not padded across the branch)
        const float na = fmaxf(sta.m, mca), nb = fmaxf(stb.m, mcb);
        const float aa = fast_exp2(sta.m - na), ab = fast_exp2(stb.m - nb);
        sta.m = na;
        stb.m = nb;
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                oa[db][r] *= aa;
                ob[db][r] *= ab;
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            sta.lacc[r] *= aa;
            stb.lacc[r] *= ab;
        }
    }
    lz.offa = sta.m + kLazyThr;
    lz.offb = stb.m + kLazyThr;
}
// One pipelined step.  CUR/NXT score buffers are passed by reference (the caller swaps them every step).
//   kf           : K fragments of sub-tile t+1 (scores computed in this step), fetched during the previous step
//   v_lds/KB_C   : LDS address / 32-key block of the V sub-tile t (accumulated in this step)
//   k_lds2/kb_n2 : K sub-tile t+2 -- its fragments are fetched during the P1 phase of this step (kf carries them over), so
//                  that every LDS read is issued a whole phase before its first use (under load the LDS latency is several
//                  MFMA slots; a read issued two slots ahead stalls the in-order wave and drains the matrix pipe)
//   MASKED       : the scores of sub-tile t+1 (first key key0_next) get the causal / sequence-end mask before their maxima
//                  are taken -- the same pipelined step serves the diagonal and ragged tiles, no slow path
//   OPT          : optimistic mix (fa_bf16_common.h): exp without clamp, no maxima, no rescale test
template <int D, int KB_C, bool PROF = false, bool MASKED = false, bool CAUSAL = false, bool OPT = false>
__device__ __forceinline__ void pp3_step(const char* v_lds, const char* k_lds2, int kb_n2, int k_row_off, int k_g, int v_lane_off, const bf16x8& ones_a, const bf16x8 (&qfa)[D / 16],
                                         const bf16x8 (&qfb)[D / 16], f32x16& sa_cur, f32x16& sb_cur, f32x16& sa_nxt, f32x16& sb_nxt,
                                         f32x16 (&oa)[D / 32], f32x16 (&ob)[D / 32], bf16x8 (&pfa)[2], bf16x8 (&pfb)[2], BlockState& sta,
                                         BlockState& stb, float c, Lazy2& lz, bf16x8 (&kf)[D / 16], bool honor_test, unsigned long long* tm = nullptr,
                                         int key0_next = 0, int qia = 0, int qib = 0, int n = 0, int hi = 0)
{
    using P = Plan3<D>;
    constexpr int KS = P::KS, DB = P::DB, NV = P::NV;
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (PROF) t0 = stamp();
    s16x4 vlo[NV], vhi[NV];
    const unsigned v_addr = (unsigned)(size_t)(lds_s16x4_t*)(v_lds + v_lane_off);
    // ---------------- Q phase: K.Q^T of sub-tile t+1 (A and B share each K fragment)  ||  exp + pack of A.
    // Every MFMA gets its own slot (two matrix instructions back to back park the in-order wave on the matrix pipe);
    // the V^T fragments of the P phases are fetched here, a whole phase ahead of their first use.
    {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // V^T fragment reads of this step: all issued in the first NV slots, >= KS slots before the P1 phase
            if (2 * ks + 0 == 0) load_v_frag_asm<D, KB_C, 0>(v_addr, vlo[0], vhi[0]);
            if (NV > 2 && 2 * ks + 0 == 2) load_v_frag_asm<D, KB_C, 2 % NV>(v_addr, vlo[2 % NV], vhi[2 % NV]);
            if (NV > 4 && 2 * ks + 0 == 4) load_v_frag_asm<D, KB_C, 4 % NV>(v_addr, vlo[4 % NV], vhi[4 % NV]);
            if (NV > 6 && 2 * ks + 0 == 6) load_v_frag_asm<D, KB_C, 6 % NV>(v_addr, vlo[6 % NV], vhi[6 % NV]);
            if (ks == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                sa_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qfa[ks], z, 0, 0, 0);
            } else {
                sa_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qfa[ks], sa_nxt, 0, 0, 0);
            }
            exp_range(sa_cur, pfa, c, lz.offa, 16 * (2 * ks) / (2 * KS), 16 * (2 * ks + 1) / (2 * KS), !OPT);
            __builtin_amdgcn_sched_barrier(0);
            if (NV > 1 && 2 * ks + 1 == 1) load_v_frag_asm<D, KB_C, 1 % NV>(v_addr, vlo[1 % NV], vhi[1 % NV]);
            if (NV > 3 && 2 * ks + 1 == 3) load_v_frag_asm<D, KB_C, 3 % NV>(v_addr, vlo[3 % NV], vhi[3 % NV]);
            if (NV > 5 && 2 * ks + 1 == 5) load_v_frag_asm<D, KB_C, 5 % NV>(v_addr, vlo[5 % NV], vhi[5 % NV]);
            if (NV > 7 && 2 * ks + 1 == 7) load_v_frag_asm<D, KB_C, 7 % NV>(v_addr, vlo[7 % NV], vhi[7 % NV]);
            if (ks == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                sb_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qfb[ks], z, 0, 0, 0);
            } else {
                sb_nxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qfb[ks], sb_nxt, 0, 0, 0);
            }
            exp_range(sa_cur, pfa, c, lz.offa, 16 * (2 * ks + 1) / (2 * KS), 16 * (2 * ks + 2) / (2 * KS), !OPT);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (PROF) t1 = stamp();
    // the asm-issued V^T reads were started a whole phase ago; this wait is the one that orders them before the MFMAs
    // (the "+v" operands stop the compiler from touching the destination registers earlier)
    static_assert(NV <= 8, "V^T fragment staging written for NV <= 8");
#pragma unroll
    for (int v = 0; v < NV; ++v) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vlo[v]), "+v"(vhi[v]));
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 vf[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) vf[v] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(vlo[v], vhi[v], 0, 1, 2, 3, 4, 5, 6, 7));
    // ---------------- P1 phase: P.V + row sums of A  ||  exp + pack of B (its first E1 elements: the k-step-1 operand of
    // block B is only needed by the third MFMA of the P2 phase, and P2 has VALU slots to spare)
    constexpr int E1 = (D == 64) ? 12 : 16;
#pragma unroll
    for (int v = 0; v < NV + 2; ++v) {
        if (v < NV) {
            oa[v % DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[v], pfa[v / DB], oa[v % DB], 0, 0, 0);
        } else {
            sta.lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_a, pfa[v - NV], sta.lacc, 0, 0, 0);
        }
        if (v < KS) kf[v] = load_k_frag<D>(k_lds2, k_row_off, k_g, kb_n2, v);   // scores of the NEXT step (kf is free: Q phase done)
        exp_range(sb_cur, pfb, c, lz.offb, E1 * v / (NV + 2), E1 * (v + 1) / (NV + 2), !OPT);
        if (MASKED && v == NV + 1) {  // K.Q^T of sub-tile t+1 finished a phase ago
            mask16(sa_nxt, key0_next, qia, n, hi, CAUSAL);
            mask16(sb_nxt, key0_next, qib, n, hi, CAUSAL);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (PROF) t2 = stamp();
    // ---------------- P2 phase: P.V + row sums of B  ||  lane-local maxima of sub-tile t+1 and the rescale test.
    // The test only needs each lane's own partial maximum: a row outgrows its reference iff one of its two lanes does,
    // so the cross-half exchange happens inside the rare rescale branch, not here.
    float lma = 0.0f, lmb = 0.0f, pm[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool need = false;
#pragma unroll
    for (int v = 0; v < NV + 2; ++v) {
        if (v < NV) {
            ob[v % DB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[v], pfb[v / DB], ob[v % DB], 0, 0, 0);
        } else {
            stb.lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_a, pfb[v - NV], stb.lacc, 0, 0, 0);
        }
        if (E1 < 16 && v < 2) exp_range(sb_cur, pfb, c, lz.offb, E1 + (16 - E1) * v / 2, E1 + (16 - E1) * (v + 1) / 2, !OPT);
#pragma unroll
        for (int u = 0; u < 6; ++u)
            if (!OPT && (E1 < 16 ? 2 + u / 2 : u * NV / 6) == v) {
                if (u < 3) lanemax_step(u, sa_nxt, pm, lma);
                else lanemax_step(u - 3, sb_nxt, pm, lmb);
            }
        // the test is evaluated one MFMA slot before the branch that consumes it (VALU compare -> scalar branch latency)
        if (!OPT && v == NV) need = fmaxf(fmaf(lma, c, -lz.offa), fmaf(lmb, c, -lz.offb)) > 0.0f;  // off = m + kLazyThr
        __builtin_amdgcn_sched_barrier(0);
    }
    if (PROF) t3 = stamp();
    if (!OPT && __builtin_expect(__any(need) && honor_test, 0)) {  // honor_test is false when sub-tile t+1 does not exist
        mfma_drain();  // the last P.V / row-sum MFMAs of block B may still be in flight
        lazy_rescale2<D>(xhalf_max(lma), xhalf_max(lmb), c, sta, stb, oa, ob, lz);
    }
    if (PROF) {
        const unsigned long long t4 = stamp();
        tm[0] += t1 - t0;
        tm[1] += t2 - t1;
        tm[2] += t3 - t2;
        tm[3] += t4 - t3;
    }
}

// G = stages (64 keys each) between two workgroup barriers.  A barrier costs the skew between the four waves, not a data
// wait (measured: ~600 cycles per barrier, DMA wait ~0), so it is paid once per G stages; the price is LDS: rings of 2G tiles
// for K and for V (G = 2: 64 KiB per workgroup, two workgroups per CU; G = 1: 32 KiB, used by the 2-wave workgroups).
// One tile of NWAVES * 64 query rows.  OPT: optimistic mix (fa_bf16_common.h); returns false, with nothing stored, when some
// row of the workgroup left its safe range.
template <int D, int NWAVES, bool CAUSAL, bool OUT_F32, bool PROF, int G, bool OPT>
__device__ __forceinline__ int pp3_tile(const FwdParams& p, char* smem)   // 1 = done; 0 = redo; 2 = redo unless V is exactly zero (kernel body)
{
    using C = Bf16Cfg<D, NWAVES>;
    constexpr float kBias = OPT ? kOptBias : kLazyThr;
    constexpr int KS = D / 16, DB = D / 32;
    constexpr int BM = NWAVES * 64;
    constexpr int KR = 2 * G, VR = 2 * G;  // ring depths in tiles (powers of two)
    static_assert(G == 1 || G == 2, "ring index arithmetic written for G = 1, 2");
    constexpr int T = C::kTileBytes;

    const unsigned long long t_entry = PROF ? stamp() : 0;
    char* const k_ring = smem;  // K ring, then V ring
    char* const v_ring = smem + KR * T;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;

    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles;
    int qt = w % p.q_tiles;
    if (CAUSAL) qt = causal_tile(p, qt);
    const int n = p.n;
    const int q0a = qt * BM + wave * 64, q0b = q0a + 32;

    const int b = slab / p.heads, h = slab % p.heads;
    const __bf16* qg = (const __bf16*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
    const __bf16* kg = (const __bf16*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const __bf16* vg = (const __bf16*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    const int64_t o_slab_off = b * p.o_batch_stride + h * p.o_head_stride;

    int kv_end = n;
    if (CAUSAL) kv_end = min(n, qt * BM + BM);
    const int nst = (kv_end + kKvBlk - 1) / kKvBlk;  // 64-key stages
    const int nsub = (kv_end + 31) / 32;             // 32-key sub-tiles

    auto k_slot = [&](int j) { return k_ring + (j & (KR - 1)) * T; };
    auto v_slot = [&](int j) { return v_ring + (j & (VR - 1)) * T; };

    TileDma<D, NWAVES> dma;
    dma.init(kg, vg, n, p.kv_row_stride, wave, lane);
    dma.issue_k(0u, k_slot(0), wave);
    // every tile the first barrier group needs is requested before anything is waited for: one memory round trip, not two
#pragma unroll
    for (int g = 1; g <= G; ++g)
        if (g < nst) dma.issue_k((unsigned)g * dma.stage_step, k_slot(g), wave);
#pragma unroll
    for (int g = 0; g < G; ++g)
        if (g < nst) dma.issue_v((unsigned)g * dma.stage_step, v_slot(g), wave);

    bf16x8 qfa[KS], qfb[KS];
    {
        const __bf16* qra = qg + (int64_t)min(q0a + lq, n - 1) * p.q_row_stride + hi * 8;
        const __bf16* qrb = qg + (int64_t)min(q0b + lq, n - 1) * p.q_row_stride + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qfa[ks] = *(const bf16x8*)(qra + ks * 16);
            qfb[ks] = *(const bf16x8*)(qrb + ks * 16);
        }
    }
    const bf16x8 ones_a = rowsum_a_operand(lane);

    f32x16 oa[DB], ob[DB], sa0, sb0, sa1, sb1;
    bf16x8 pfa[2], pfb[2];
    BlockState sta, stb;
    Lazy2 lz;
    sta.m = stb.m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) sta.lacc[r] = stb.lacc[r] = 0.0f;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oa[db][r] = ob[db][r] = 0.0f;

    const int k_row_off = lq * C::kRowBytes;
    const int k_g = hi ^ k_swizzle<D>(lq);
    const int li = lane & 15;
    const int v_lane_off = (hi * (D / 16) + ((lane >> 4) & 1)) * 128 + (li >> 2) * 32 + (li & 3) * 8;
    const float c = p.scale_log2e;

    // sub-tile t needs a mask for the block whose first row is q0?
    auto needs_mask = [&](int t, int q0) { return (t * 32 + 32 > n) || (CAUSAL && (t * 32 + 31 > q0)); };

    unsigned long long tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // Top of stage j, j a multiple of G.  Afterwards K(j+1 .. j+G) and V(j .. j+G-1) are visible to every wave and
    // K(j+G+1 .. j+2G), V(j+G .. j+2G-1) are on their way into the ring slots of K(j-G+1 .. j), V(j-G .. j-1), which
    // nobody reads any more: K tiles are only ever read into the fragment registers kf one step ahead of their use, and
    // every LDS read of a wave has returned before it arrives at the barrier.
    auto sync_top = [&](int j) {
        const unsigned long long s0 = PROF ? stamp() : 0;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // own DMA pieces landed, own LDS reads returned
        const unsigned long long s1 = PROF ? stamp() : 0;
        __syncthreads();
        const unsigned long long s2 = PROF ? stamp() : 0;
#pragma unroll
        for (int g = 1; g <= G; ++g)
            if (__builtin_expect(j + G + g < nst, 1)) dma.issue_k((unsigned)(j + G + g) * dma.stage_step, k_slot(j + G + g), wave);
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (__builtin_expect(j + G + g < nst, 1)) dma.issue_v((unsigned)(j + G + g) * dma.stage_step, v_slot(j + G + g), wave);
        if (PROF) {
            tm[4] += s1 - s0;
            tm[5] += s2 - s1;
            tm[6] += stamp() - s2;
        }
    };
    bf16x8 kf[KS];  // K fragments of the sub-tile whose scores are computed next
    auto load_kf = [&](int t) {
        const char* k_lds = k_slot(t >> 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = load_k_frag<D>(k_lds, k_row_off, k_g, t & 1, ks);
    };
    // scores of sub-tile t for both blocks from the fragments in kf, phase-structured (prologue and tail)
    auto qk_regs = [&](int t, f32x16& sa, f32x16& sb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sa[r] = sb[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qfa[ks], sa, 0, 0, 0);
            sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qfb[ks], sb, 0, 0, 0);
        }
        if (needs_mask(t, q0a)) mask16(sa, t * 32, q0a + lq, n, hi, CAUSAL);
        if (needs_mask(t, q0b)) mask16(sb, t * 32, q0b + lq, n, hi, CAUSAL);
        if (OPT) {  // (only called for sub-tile 0 in this mix) fix the references for the whole row
            const float mca = rowmax16(sa) * c, mcb = rowmax16(sb) * c;
            sta.m = fmaf(-fabsf(mca), 0x1p-23f, mca);
            stb.m = fmaf(-fabsf(mcb), 0x1p-23f, mcb);
            lz.offa = sta.m + kBias;
            lz.offb = stb.m + kBias;
        } else {
            lazy_rescale2<D>(rowmax16(sa), rowmax16(sb), c, sta, stb, oa, ob, lz);
        }
    };
    // ---------------- prologue: K(0) landed -> scores of sub-tile 0, fragments of sub-tile 1 ----------------
    wait_lds_dma();
    __syncthreads();
    load_kf(0);
    qk_regs(0, sa0, sb0);
    load_kf(1);

    // ---------------- fast loop: groups of G whole stages whose sub-tiles 2j .. 2j+2 are in range and mask-free ----------------
    // closed form of the obvious scan over j (which was O(N / 64) scalar iterations per wave): (2 jf + 3) * 32 <= kv_end, and for a
    // causal tile sub-tile 2 j + 2 below the diagonal of the first row of both blocks (q0a < q0b)
    int jf = kv_end >= 96 ? (kv_end / 32 - 3) / 2 + 1 : 0;
    if (CAUSAL) jf = min(q0a, q0b) >= 95 ? min(jf, (min(q0a, q0b) - 95) / 64 + 1) : 0;
    if (n < 32) jf = 0;
    // The last stage may run in the fast loop too when its own two sub-tiles are whole and mask-free: its second step then
    // computes scores of a sub-tile that does not exist (from whatever the ring slot holds) and nobody consumes them --
    // the rescale test of that step is ignored.  Without this the final 128 keys of every slab took the slow tail path.
    if (jf == nst - 1 && (2 * jf + 2) * 32 <= kv_end && !needs_mask(2 * jf + 1, q0a) && !needs_mask(2 * jf + 1, q0b) && !needs_mask(0, q0a))
        jf = nst;
    jf -= jf % G;
    const unsigned long long t_begin = PROF ? stamp() : 0;
    for (int j = 0; j < jf; j += G) {
        sync_top(j);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const char* v_lds = v_slot(j + g);
            const char* k_nxt = k_slot(j + g + 1);
            // step 2(j+g): scores(2(j+g)+1) from kf; P.V(2(j+g)) from V block 0; fetch for scores(2(j+g)+2): K(j+g+1) block 0
            pp3_step<D, 0, PROF, false, false, OPT>(v_lds, k_nxt, 0, k_row_off, k_g, v_lane_off, ones_a, qfa, qfb, sa0, sb0, sa1, sb1, oa, ob, pfa, pfb, sta,
                                 stb, c, lz, kf, true, tm);
            // step 2(j+g)+1: scores(2(j+g)+2) from kf; P.V(2(j+g)+1) from V block 1; fetch for scores(2(j+g)+3): K(j+g+1) block 1
            pp3_step<D, 1, PROF, false, false, OPT>(v_lds, k_nxt, 1, k_row_off, k_g, v_lane_off, ones_a, qfa, qfb, sa1, sb1, sa0, sb0, oa, ob, pfa, pfb, sta,
                                 stb, c, lz, kf, 2 * (j + g) + 2 < nsub, tm);
        }
    }
    if (PROF) {
        tm[7] = stamp() - t_begin;
        if (lane == 0 && p.lse != nullptr) {
            float* dst = p.lse + ((int64_t)blockIdx.x * NWAVES + wave) * 16;
            for (int i = 0; i < 8; ++i) dst[i] = (float)tm[i];
            dst[8] = (float)jf;
        }
    }

    // ---------------- tail: the remaining stages through the same pipelined step with masks applied ----------------
    // Sub-tiles past the end of the keys (their ring slots hold stale or zero data) are masked out entirely, so the step
    // after the last real sub-tile is harmless.  Every wave keeps taking part in the barriers / DMA, but only computes
    // the stages its own rows can see (causal: up to the diagonal of its last row).
    const int nsub_w = CAUSAL ? min(nsub, (q0b + 31) / 32 + 1) : nsub;
    for (int j = jf; j < nst; ++j) {
        if (j % G == 0) sync_top(j);
        if (2 * j < nsub_w) {
            const char* v_lds = v_slot(j);
            const char* k_nxt = k_slot(j + 1);
            pp3_step<D, 0, false, true, CAUSAL, OPT>(v_lds, k_nxt, 0, k_row_off, k_g, v_lane_off, ones_a, qfa, qfb, sa0, sb0, sa1, sb1, oa, ob, pfa,
                                                pfb, sta, stb, c, lz, kf, true, nullptr, (2 * j + 1) * 32, q0a + lq, q0b + lq, n, hi);
            pp3_step<D, 1, false, true, CAUSAL, OPT>(v_lds, k_nxt, 1, k_row_off, k_g, v_lane_off, ones_a, qfa, qfb, sa1, sb1, sa0, sb0, oa, ob, pfa,
                                                pfb, sta, stb, c, lz, kf, true, nullptr, (2 * j + 2) * 32, q0a + lq, q0b + lq, n, hi);
        }
    }

    // ---------------- store; verify (optimistic mix) ----------------
    // The optimistic tile stores its result BEFORE the workgroup votes on it: a failed tile is simply overwritten by the redo,
    // and nothing of the first attempt is live across the vote.
    mfma_drain();  // the loop exit is a branch: the last P.V / row-sum MFMAs may still be in flight
    // the tile stands iff no term left the safe range, which the row sums prove (fa_bf16_common.h)
    bool bad = OPT && (!(sta.lacc[0] < kOptLimit) || !(stb.lacc[0] < kOptLimit));
    bool hard = bad;   // ... for another reason than accumulators that are EXACTLY zero
    auto store_block = [&](const f32x16 (&o)[DB], const BlockState& st, int q0) {
        if constexpr (OPT) asm volatile("; store, optimistic mix");  // distinct text per mix: keeps hipcc from tail-merging the
        else asm volatile("; store, rescaled mix");                   // store code of the two inlined tiles (copies + scratch)
        const float lt = st.lacc[0];
        const float inv = 1.0f / lt;
        const int qi = q0 + lq;
        if constexpr (OPT) {
            // the tiny-accumulator vote of the one-wave-per-SIMD kernels (xn_tile), which this kernel lacked until the end of round 5: P sits
            // near 2^-100, so the products of the terms that matter go subnormal when |v| is below ~2^-26 and vanish below ~2^-50 -- such rows
            // came out with a meaningless relative error (profiles/r05_exp/exp10_tiny_v_by_kernel.py).  All tiny or zero: the rescaled redo.
            float amax = 0.0f;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; r += 2) amax = fmaxf(fmaxf(amax, fabsf(o[db][r])), fabsf(o[db][r + 1]));
            const bool tiny = amax < kOptTinyAcc && qi < n;
            bad = bad || tiny;
            hard = hard || (tiny && amax != 0.0f);
        }
        if (qi < n) {
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
                        for (int e = 0; e < 4; ++e) pk[e] = o[db][4 * g + e] * inv;
                        *(f32x4*)((float*)p.o + o_off + db * 32 + 8 * g) = pk;
                    } else if ((g & 1) == 0) {
                        // 16-byte stores: the two lanes of a row (hi = 0 / 1) hold alternate 4-column groups; one
                        // v_permlane32_swap per dword hands lane hi = 0 both halves of column group g and lane hi = 1 both
                        // halves of group g + 1 (the epilogue is store-issue bound: half as many, twice as wide)
                        bf16x4 pe, po;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            pe[e] = (__bf16)(o[db][4 * g + e] * inv);
                            po[e] = (__bf16)(o[db][4 * (g + 1) + e] * inv);
                        }
                        const u32x2 ue = __builtin_bit_cast(u32x2, pe), uo = __builtin_bit_cast(u32x2, po);
                        const auto r0 = __builtin_amdgcn_permlane32_swap(ue[0], uo[0], false, false);
                        const auto r1 = __builtin_amdgcn_permlane32_swap(ue[1], uo[1], false, false);
                        u32x4 w;
                        w[0] = r0[0];
                        w[1] = r1[0];
                        w[2] = r0[1];
                        w[3] = r1[1];
                        *(u32x4*)((__bf16*)p.o + o_off - 4 * hi + db * 32 + 8 * (g + hi)) = w;
                    }
                }
            if (!PROF && p.lse != nullptr && hi == 0)
                p.lse[(int64_t)slab * n + qi] = (st.m + kBias + __builtin_amdgcn_logf(lt)) * kLn2;
        }
    };
    store_block(oa, sta, q0a);
    store_block(ob, stb, q0b);
    if (OPT) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // stores and DMA of this attempt done before a redo starts
        if (__syncthreads_or(bad ? 1 : 0)) return __syncthreads_or(hard ? 1 : 0) ? 0 : 2;   // workgroup-wide: the redo shares tiles and barriers
    }
    if (PROF && lane == 0 && p.lse != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        float* dst = p.lse + ((int64_t)blockIdx.x * NWAVES + wave) * 16;
        dst[9] = (float)(stamp() - t_entry);  // whole kernel
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        dst[10] = (float)(((xcc & 0xf) << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 8) & 0xf) << 4) | ((hwid >> 4) & 3));
        dst[11] = (float)(t_entry & 0xffffff);
    }
    return 1;
}

// OPTIMISTIC: try the fixed-reference mix first, redo the tile with the lazily rescaled mix if its verification fails
template <int D, int NWAVES, bool CAUSAL, bool OUT_F32, bool PROF = false, int G = 2, bool OPTIMISTIC = true>
__global__ __launch_bounds__(NWAVES* kWave, 2) void fa_fwd_bf16_pp3_kernel(FwdParams p)
{
    using C = Bf16Cfg<D, NWAVES>;
    __shared__ __attribute__((aligned(1024))) char smem[4 * G * C::kTileBytes];
    if (OPTIMISTIC && !PROF) {
        const int r = pp3_tile<D, NWAVES, CAUSAL, OUT_F32, false, G, true>(p, smem);
        if (r == 1) return;
        if (r == 2 && bf16_v_is_zero<D, NWAVES>(p)) return;   // an all-zero V: the stored zeros are the result
        count_cliff(p, 0);
    }
    (void)pp3_tile<D, NWAVES, CAUSAL, OUT_F32, PROF, G, false>(p, smem);
}

// the buffer-form LDS-DMA addresses a slab with 32-bit byte offsets
static bool pp3_addressable(const FwdParams& p, int d)
{
    return ((int64_t)(p.n - 1) * p.kv_row_stride + d) * 2 < (int64_t)0xffffffffLL;
}

#if FA_ABLATION
static hipError_t launch_pp3_prof(const FwdParams& p0, hipStream_t stream)
{
    FwdParams p = p0;
    p.q_tiles = (p.n + 255) / 256;
    dim3 grid((unsigned)(p.bh * p.q_tiles)), block(256);
    hipLaunchKernelGGL((fa_fwd_bf16_pp3_kernel<64, 4, false, false, true, 2>), grid, block, 0, stream, p);
    return hipGetLastError();
}
#endif

template <int D, int NWAVES, int G, bool OPTIMISTIC = true>
static hipError_t launch_pp3(const FwdParams& p0, int causal, int out_f32, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = NWAVES * 64;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)total), block(NWAVES * kWave);
    // Two workgroups per CU by construction: on causal grids of whole rounds (a multiple of 256 workgroups) the tiles of a slab are
    // dealt alternately from the heavy and the light end (causal_tile).  Measured, causal d = 64, ms plain -> alternating: 128 x 2048
    // 0.117 -> 0.110, 256 x 1024 0.073 -> 0.064, 512 x 512 0.051 -> 0.047, 64 x 2048 0.065 -> 0.054; with a partial last round it
    // loses (48 x 3000, 576 workgroups: 0.085 -> 0.095) and stays off.  (16 x 8192: 0.204 -> 0.154, which only reaches what the
    // one-wave-per-SIMD kernel does on such grids -- the dispatch keeps sending long rows there.)
    p.alt_order = (causal && NWAVES == 4 && total % 256 == 0) ? 1 : 0;
    if (causal) {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_pp3_kernel<D, NWAVES, true, true, false, G, OPTIMISTIC>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_pp3_kernel<D, NWAVES, true, false, false, G, OPTIMISTIC>), grid, block, 0, stream, p);
    } else {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_pp3_kernel<D, NWAVES, false, true, false, G, OPTIMISTIC>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_pp3_kernel<D, NWAVES, false, false, false, G, OPTIMISTIC>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

bool bf16_pipelined_supported(const FwdParams& p, int d) { return (d == 64 || d == 32) && pp3_addressable(p, d); }

// mode: 0 = product configuration (optimistic mix first; 4-wave workgroups: barrier every 2 stages, 2-wave workgroups: every
//       stage), 1 = in-kernel phase timers (D = 64, non-causal; written to lse), 3 = lazily rescaled mix only
hipError_t launch_bf16_pipelined(const FwdParams& p, int d, int nwaves, int causal, int out_f32, int mode, hipStream_t stream)
{
    if (!bf16_pipelined_supported(p, d)) return hipErrorInvalidValue;
#if FA_ABLATION
    if (mode == 1) return d == 64 ? launch_pp3_prof(p, stream) : hipErrorInvalidValue;
#else
    if (mode == 1) return hipErrorInvalidValue;
#endif
#if FA_ABLATION
    // previous-generation and comparison tilings: two-wave workgroups, the lazily rescaled mix alone, head dim 32 (the NB = 2 kernel took it over)
    if (d == 64 && nwaves == 2) return mode == 3 ? launch_pp3<64, 2, 1, false>(p, causal, out_f32, stream) : launch_pp3<64, 2, 1>(p, causal, out_f32, stream);
    if (d == 64 && mode == 3) return launch_pp3<64, 4, 2, false>(p, causal, out_f32, stream);
    if (d == 32 && nwaves == 2) return launch_pp3<32, 2, 1>(p, causal, out_f32, stream);
    if (d == 32) return mode == 3 ? launch_pp3<32, 4, 2, false>(p, causal, out_f32, stream) : launch_pp3<32, 4, 2>(p, causal, out_f32, stream);
#endif
    // the product tiling: head dim 64, four-wave workgroups, optimistic mix first (the only one the dispatch reaches)
    if (d != 64 || nwaves != 4 || mode != 0) return hipErrorInvalidValue;
    return launch_pp3<64, 4, 2>(p, causal, out_f32, stream);
}

}  // namespace fa
