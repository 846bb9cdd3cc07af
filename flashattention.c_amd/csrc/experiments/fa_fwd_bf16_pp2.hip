// fa_fwd_bf16_pp2.hip -- second-generation "ping-pong" kernel and its ablation variants (kept as the measured baseline
// of DESIGN.md section 3: timing-only ablations of the main loop, not a product path).  See fa_fwd_bf16_pipelined.hip for
// the kernel that ships.
#include "fa_bf16_common.h"
#include "fa_kernels.h"

namespace fa {

// =====================================================================================================================
// Ping-pong kernel, second generation ("pp2"): the same two-block half-tile-skewed structure, with the softmax trimmed
// to what the VALU cannot avoid (PMC + profiles/r01_ubench_issue.txt: the loop is bound by per-wave VALU issue, ~5.7
// cycles per instruction, not by the matrix pipe):
//   * row sums come from the matrix core: one v_mfma_f32_16x16x32_bf16 per P fragment against a constant 0/1 A operand
//     built so that EVERY lane receives the full sum of its own query row (both half-waves) -- 4 small MFMAs replace 32
//     adds per block-tile, and numerator and denominator now see the same bf16-rounded P;
//   * the O accumulator is rescaled lazily: the running max used in the exponent only moves when some row of the wave
//     outgrows it by more than 2^kLazyThr (wave-uniform, rare branch); the exponent carries a -kLazyThr bias so p <= 1
//     still holds and the v_exp clamp keeps protecting against overflow for any input magnitude;
//   * per MFMA slot the wave issues ~5 VALU + 1-2 LDS instructions: the issue time of a slot matches the 32 cycles its
//     MFMA occupies the pipe.
// =====================================================================================================================
template <int D>
struct Plan2 {
    static constexpr int KS = D / 16, DB = D / 32;
    static constexpr int NPV = 4 * DB, NSUM = 4, NQK = 2 * KS, NSLOT = NPV + NSUM + NQK;
    static constexpr int N_MAX = NSLOT / 10 > 0 ? NSLOT / 10 : 1;  // slots for the max phase (before the rescale decision)
    static constexpr int N_FIN = NSLOT / 5;                       // slots for packing P
    static constexpr int N_EXP = NSLOT - N_MAX - N_FIN;
    static constexpr int max_slot(int e) { return e * N_MAX / 32; }
    static constexpr int exp_slot(int e) { return N_MAX + e * N_EXP / 32; }
    static constexpr int fin_slot(int f) { return N_MAX + N_EXP + f * N_FIN / 4; }
};

// per-lane constant A operand of the row-sum MFMA: A[i][k] = ((i >> 2) & 1) == ((k >> 3) & 1)
template <int D>
__device__ __forceinline__ float lazy_rescale(float mx_raw, float c, BlockState& st, f32x16 (&o)[D / 32])
{
    float mc = mx_raw * c;
    mc = fmaf(-fabsf(mc), 0x1p-23f, mc);  // nudge down: c*s_max - mc >= 0 exactly (see fa_fwd_bf16_kernel)
    if (__builtin_expect(__any(mc - st.m > kLazyThr), 0)) {
        asm volatile("; lazy rescale" ::: "memory");  // keep this a real (non-speculated) branch
        mfma_drain();  // the accumulators rescaled below may have an MFMA in flight (hazard not padded across the branch)
        const float m_new = fmaxf(st.m, mc);
        const float alpha = fast_exp2(st.m - m_new);  // 0 on the first tile (m = -inf)
        st.m = m_new;
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) st.lacc[r] *= alpha;
    }
    return st.m + kLazyThr;
}

__device__ __forceinline__ float block_rowmax(const f32x16 (&s)[2])
{
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    return xhalf_max(mx);
}

// phase-structured softmax (prologue, masked tail tiles, epilogue)
template <int D>
__device__ __forceinline__ void softmax_block2(f32x16 (&s)[2], BlockState& st, f32x16 (&o)[D / 32], bf16x8 (&pf)[4], float c)
{
    const float off = lazy_rescale<D>(block_rowmax(s), c, st, o);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = exp2_clamp01(fmaf(s[kb][r], c, -off));
#pragma unroll
    for (int f = 0; f < 4; ++f) pf[f] = pack_bf16x8(s[f >> 1], 8 * (f & 1));
}

template <int D>
__device__ __forceinline__ void load_frag2(int i, const char* k_lds, const char* v_lds, int k_row_off, int k_g, int v_lane_off,
                                           bf16x8 (&fr)[Plan2<D>::NSLOT])
{
    using P = Plan2<D>;
    constexpr int RB = 2 * D;
    if (i < P::NPV) {
        const int db = i % P::DB, kt = i / P::DB, kb = kt >> 1, t = kt & 1;
        const int off0 = ((kb * 8 + 4 * t + 0) * (D / 16) + 2 * db) * 128;
        const int off1 = ((kb * 8 + 4 * t + 2) * (D / 16) + 2 * db) * 128;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_lds + v_lane_off + off0));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_lds + v_lane_off + off1));
        fr[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
    } else if (i >= P::NPV + P::NSUM && i < P::NSLOT) {
        const int q = i - P::NPV - P::NSUM, ks = q >> 1, kb = q & 1;
        fr[i] = *(const bf16x8*)(k_lds + k_row_off + kb * 32 * RB + (((2 * ks) ^ k_g) * 16));
    }
}

template <int D>
__device__ __forceinline__ void mfma_slot2(int i, const bf16x8 (&fr)[Plan2<D>::NSLOT], const bf16x8& ones_a, const bf16x8 (&pf)[4],
                                           const bf16x8 (&qf)[D / 16], f32x16 (&o)[D / 32], BlockState& st, f32x16 (&s)[2])
{
    using P = Plan2<D>;
    if (i < P::NPV) {
        const int db = i % P::DB, kt = i / P::DB;
        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], pf[kt], o[db], 0, 0, 0);
    } else if (i < P::NPV + P::NSUM) {
        st.lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_a, pf[i - P::NPV], st.lacc, 0, 0, 0);
    } else {
        const int q = i - P::NPV - P::NSUM, ks = q >> 1, kb = q & 1;
        if (ks == 0) {
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.0f;
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], qf[ks], z, 0, 0, 0);
        } else {
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i], qf[ks], s[kb], 0, 0, 0);
        }
    }
}

// One half iteration of the mask-free main loop: the MFMA stream of block X (P.V, row sums, then K.Q^T of the next
// tile) with the softmax of block Y sliced into its slots.
template <int D, int AHEAD, int ABL>
__device__ __forceinline__ void half_iteration2(const char* k_lds, const char* v_lds, int k_row_off, int k_g, int v_lane_off,
                                                const bf16x8& ones_a, const bf16x8 (&pfx)[4], const bf16x8 (&qfx)[D / 16],
                                                f32x16 (&ox)[D / 32], BlockState& stx, f32x16 (&sx)[2], f32x16 (&sy)[2],
                                                BlockState& sty, f32x16 (&oy)[D / 32], bf16x8 (&pfy)[4], float c)
{
    using P = Plan2<D>;
    bf16x8 fr[P::NSLOT];
    float pm[4];
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
        if (!(ABL & 8) || !(i & 1)) load_frag2<D>(i, k_lds, v_lds, k_row_off, k_g, v_lane_off, fr);
    // ---- slots [0, N_MAX): MFMAs of X beside the partial row maxima of Y
#pragma unroll
    for (int i = 0; i < P::N_MAX; ++i) {
        if (i + AHEAD < P::NSLOT && (!(ABL & 8) || !((i + AHEAD) & 1))) load_frag2<D>(i + AHEAD, k_lds, v_lds, k_row_off, k_g, v_lane_off, fr);
        if ((ABL & 8) && (i & 1)) fr[i] = fr[i - 1];
        if (!(ABL & 2)) mfma_slot2<D>(i, fr, ones_a, pfx, qfx, ox, stx, sx);
        else asm volatile("" ::"v"(fr[i]));
#pragma unroll
        for (int e = 0; e < 32; e += 2)
            if (!(ABL & 1) && !(ABL & 16) && P::max_slot(e) == i) {
                const float a = sy[e >> 4][e & 15], b2 = sy[e >> 4][(e & 15) + 1];
                const int k = (e >> 1) & 3;
                pm[k] = (e < 8) ? max3_raw(a, a, b2) : max3_raw(pm[k], a, b2);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    float off = 0.0f;
    if (!(ABL & 1) && !(ABL & 16)) {
        const float mx = xhalf_max(fmaxf(max3_raw(pm[0], pm[1], pm[2]), pm[3]));
        off = lazy_rescale<D>(mx, c, sty, oy);  // rare wave-uniform branch inside
    }
    if (ABL & 16) off = sty.m + kLazyThr;
    // ---- slots [N_MAX, NSLOT): exp, then pack
#pragma unroll
    for (int i = P::N_MAX; i < P::NSLOT; ++i) {
        if (i + AHEAD < P::NSLOT && (!(ABL & 8) || !((i + AHEAD) & 1))) load_frag2<D>(i + AHEAD, k_lds, v_lds, k_row_off, k_g, v_lane_off, fr);
        if ((ABL & 8) && (i & 1)) fr[i] = fr[i - 1];
        if (!(ABL & 2)) mfma_slot2<D>(i, fr, ones_a, pfx, qfx, ox, stx, sx);
        else asm volatile("" ::"v"(fr[i]));
#pragma unroll
        for (int e = 0; e < 32; ++e)
            if (!(ABL & 1) && P::exp_slot(e) == i) sy[e >> 4][e & 15] = exp2_clamp01(fmaf(sy[e >> 4][e & 15], c, -off));
#pragma unroll
        for (int f = 0; f < 4; ++f)
            if (!(ABL & 1) && P::fin_slot(f) == i) {
                pfy[f] = pack_bf16x8(sy[f >> 1], 8 * (f & 1));
                asm volatile("" : "+v"(pfy[f]));  // pin the pack to this slot (its consumers live in the next basic block)
            }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int D, int NWAVES, bool CAUSAL, bool OUT_F32, int AHEAD, int ABL = 0>
__global__ __launch_bounds__(NWAVES* kWave, 2) void fa_fwd_bf16_pp2_kernel(FwdParams p)
{
    using C = Bf16Cfg<D, NWAVES>;
    constexpr int KS = D / 16, DB = D / 32;
    constexpr int BM = NWAVES * 64;

    __shared__ __attribute__((aligned(1024))) char smem[4 * C::kTileBytes];  // K ring [2], then V ring [2]
    char* const k_ring = smem;
    char* const v_ring = smem + 2 * C::kTileBytes;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;

    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles;
    int qt = w % p.q_tiles;
    if (CAUSAL) qt = p.q_tiles - 1 - qt;
    const int n = p.n;
    const int q0a = qt * BM + wave * 64, q0b = q0a + 32;

    const int b = slab / p.heads, h = slab % p.heads;
    const __bf16* qg = (const __bf16*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
    const __bf16* kg = (const __bf16*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const __bf16* vg = (const __bf16*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    const int64_t o_slab_off = b * p.o_batch_stride + h * p.o_head_stride;

    int kv_end = n;
    if (CAUSAL) kv_end = min(n, qt * BM + BM);
    const int nt = (kv_end + kKvBlk - 1) / kKvBlk;

    issue_k_tile<D, NWAVES>(kg, 0, n, p.kv_row_stride, k_ring, wave, lane);

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

    f32x16 oa[DB], ob[DB], sa[2], sb[2];
    bf16x8 pfa[4], pfb[4];
    BlockState sta, stb;
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

    auto needs_mask = [&](int tile, int q0) { return (tile * kKvBlk + kKvBlk > n) || (CAUSAL && (tile * kKvBlk + kKvBlk - 1 > q0)); };
    auto stage_tiles = [&](int j) {
        wait_lds_dma();   // K(j), V(j-1): own pieces landed
        __syncthreads();  // everyone's pieces landed; everyone is done with K(j-1), V(j-2)
        if (j + 1 < nt) issue_k_tile<D, NWAVES>(kg, (j + 1) * kKvBlk, n, p.kv_row_stride, k_ring + ((j + 1) & 1) * C::kTileBytes, wave, lane);
        issue_v_tile<D, NWAVES>(vg, j * kKvBlk, n, p.kv_row_stride, v_ring + (j & 1) * C::kTileBytes, wave, lane);
    };

    // ---------------- prologue: tile 0 (phase-structured) ----------------
    wait_lds_dma();
    __syncthreads();
    if (nt > 1) issue_k_tile<D, NWAVES>(kg, kKvBlk, n, p.kv_row_stride, k_ring + C::kTileBytes, wave, lane);
    issue_v_tile<D, NWAVES>(vg, 0, n, p.kv_row_stride, v_ring, wave, lane);
    qk_block<D>(k_ring, k_row_off, k_g, qfa, sa);
    if (needs_mask(0, q0a)) mask_block<CAUSAL>(sa, 0, q0a + lq, n, hi);
    qk_block<D>(k_ring, k_row_off, k_g, qfb, sb);
    softmax_block2<D>(sa, sta, oa, pfa, c);

    // Masks are only needed on the LAST tiles of a workgroup (ragged tail, causal diagonal): a mask-free loop whose halves
    // are straight-line slot-pinned code, then a plain loop for the few masked tiles (see fa_fwd_bf16_pp_kernel).
    int j_split = nt;
    for (int j = 1; j < nt; ++j)
        if (needs_mask(j - 1, q0b) || needs_mask(j, q0a)) {
            j_split = j;
            break;
        }

    for (int j = 1; j < j_split; ++j) {
        if (!(ABL & 4)) stage_tiles(j);
        const char* k_lds = k_ring + (j & 1) * C::kTileBytes;
        const char* v_lds = v_ring + ((j - 1) & 1) * C::kTileBytes;
        half_iteration2<D, AHEAD, ABL>(k_lds, v_lds, k_row_off, k_g, v_lane_off, ones_a, pfa, qfa, oa, sta, sa, sb, stb, ob, pfb, c);
        half_iteration2<D, AHEAD, ABL>(k_lds, v_lds, k_row_off, k_g, v_lane_off, ones_a, pfb, qfb, ob, stb, sb, sa, sta, oa, pfa, c);
    }

    for (int j = j_split; j < nt; ++j) {
        stage_tiles(j);
        const char* k_lds = k_ring + (j & 1) * C::kTileBytes;
        const char* v_lds = v_ring + ((j - 1) & 1) * C::kTileBytes;
        if (needs_mask(j - 1, q0b)) mask_block<CAUSAL>(sb, (j - 1) * kKvBlk, q0b + lq, n, hi);
        pv_block<D>(v_lds, v_lane_off, pfa, oa);
        sum_block(ones_a, pfa, sta);
        qk_block<D>(k_lds, k_row_off, k_g, qfa, sa);
        softmax_block2<D>(sb, stb, ob, pfb, c);
        if (needs_mask(j, q0a)) mask_block<CAUSAL>(sa, j * kKvBlk, q0a + lq, n, hi);
        pv_block<D>(v_lds, v_lane_off, pfb, ob);
        sum_block(ones_a, pfb, stb);
        qk_block<D>(k_lds, k_row_off, k_g, qfb, sb);
        softmax_block2<D>(sa, sta, oa, pfa, c);
    }

    // ---------------- epilogue: P.V of the last tile ----------------
    wait_lds_dma();
    __syncthreads();
    {
        const char* v_lds = v_ring + ((nt - 1) & 1) * C::kTileBytes;
        if (needs_mask(nt - 1, q0b)) mask_block<CAUSAL>(sb, (nt - 1) * kKvBlk, q0b + lq, n, hi);
        pv_block<D>(v_lds, v_lane_off, pfa, oa);
        sum_block(ones_a, pfa, sta);
        softmax_block2<D>(sb, stb, ob, pfb, c);
        pv_block<D>(v_lds, v_lane_off, pfb, ob);
        sum_block(ones_a, pfb, stb);
    }

    // ---------------- store ----------------
    mfma_drain();  // the loop exit is a branch: the last P.V / row-sum MFMAs may still be in flight
    auto store_block = [&](const f32x16 (&o)[DB], const BlockState& st, int q0) {
        const float lt = st.lacc[0];
        const float inv = 1.0f / lt;
        const int qi = q0 + lq;
        if (qi < n) {
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
                    } else {
                        bf16x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(o[db][4 * g + e] * inv);
                        *(bf16x4*)((__bf16*)p.o + o_off + db * 32 + 8 * g) = pk;
                    }
                }
            if (p.lse != nullptr && hi == 0)
                p.lse[(int64_t)slab * n + qi] = (st.m + kLazyThr + __builtin_amdgcn_logf(lt)) * kLn2;
        }
    };
    store_block(oa, sta, q0a);
    store_block(ob, stb, q0b);
}

template <int D, int NWAVES, int AHEAD>
static hipError_t launch_pp2(const FwdParams& p0, int causal, int out_f32, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = NWAVES * 64;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)total), block(NWAVES * kWave);
    if (causal) {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_pp2_kernel<D, NWAVES, true, true, AHEAD>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_pp2_kernel<D, NWAVES, true, false, AHEAD>), grid, block, 0, stream, p);
    } else {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_pp2_kernel<D, NWAVES, false, true, AHEAD>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_pp2_kernel<D, NWAVES, false, false, AHEAD>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

template <int ABL>
static hipError_t launch_pp2_ablation(const FwdParams& p0, hipStream_t stream)
{
    FwdParams p = p0;
    p.q_tiles = (p.n + 255) / 256;
    dim3 grid((unsigned)(p.bh * p.q_tiles)), block(256);
    hipLaunchKernelGGL((fa_fwd_bf16_pp2_kernel<64, 4, false, false, 2, ABL>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_bf16_pp2(const FwdParams& p, int causal, int out_f32, int variant, hipStream_t stream)
{
    switch (variant) {
        case 6: return launch_pp2<64, 4, 4>(p, causal, out_f32, stream);
        // ablations of the main loop (results are garbage; timing only): 1 = no softmax VALU, 2 = no MFMA,
        // 4 = no barrier / DMA, 8 = half the LDS fragment reads, 16 = no max phase; and combinations
        case 11: return launch_pp2_ablation<1>(p, stream);
        case 12: return launch_pp2_ablation<2>(p, stream);
        case 13: return launch_pp2_ablation<3>(p, stream);
        case 14: return launch_pp2_ablation<4>(p, stream);
        case 15: return launch_pp2_ablation<5>(p, stream);
        case 16: return launch_pp2_ablation<6>(p, stream);
        case 17: return launch_pp2_ablation<8>(p, stream);
        case 18: return launch_pp2_ablation<16>(p, stream);
        case 19: return launch_pp2_ablation<24>(p, stream);
        case 20: return launch_pp2_ablation<12>(p, stream);
        case 21: return launch_pp2_ablation<28>(p, stream);
        default: return launch_pp2<64, 4, 2>(p, causal, out_f32, stream);
    }
}

}  // namespace fa
