// fa_fwd_bf16.hip -- fused flash-attention forward, bf16 in / fp32 accumulate / bf16 out, for gfx950.
//
// Replaces the hot loop of flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:214-354 and
// :434,480-484) with a design derived for CDNA4, not a translation of the CUDA tiling:
//
//   workgroup   NWAVES wavefronts (64 lanes each); wave w owns QB blocks of 32 query rows; all waves share the
//               K/V tiles of one (batch*head) slab, KVBLK = 64 keys per tile.
//   HBM -> LDS  K and V tiles go straight to LDS with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave
//               instruction, no VGPR round trip), double buffered: tile j+1 is in flight while tile j is consumed;
//               one barrier per tile.  The DMA destination is lane-linear, so both LDS images are produced by
//               permuting the per-lane SOURCE address:
//                 K image  row-major [key][D] with the 16-byte slots of each row XOR-swizzled so that the
//                          ds_read_b128 of an MFMA A fragment (16 different keys, same column slot) is
//                          bank-conflict free;
//                 V image  [key/4][col/16][4][16] sub-tiles -- the gather shape of ds_read_b64_tr_b16, which
//                          hands each lane 4 consecutive KEYS of one column, i.e. V^T fragments, for free.
//   S^T = K Q^T v_mfma_f32_32x32x16_bf16 with K as the A operand ("swapped" product): lane (q = lane&31, hi = lane>>5)
//               ends up holding 16 scores of ONE query row per 32-key block, so the row max / row sum of the online
//               softmax are in-lane reductions plus a single exchange with lane^32 (the reference does a serial
//               32-wide scan per thread through shared memory, flashattention.cu:265-274).
//   softmax     exp2 domain: p = exp2(s * scale*log2e - m); running max m and partial row sum l stay in registers;
//               the two half-wave partial sums are only combined in the epilogue.
//   O^T += V^T P^T   second MFMA chain.  The 16 scores a lane holds are exactly the B-operand k-slots of that
//               lane if the contraction index is permuted as key = 4*hi + (r&3) + 8*(r>>2); because a contraction
//               may be summed in any order, the SAME permutation is applied to the V^T fragment addresses and P
//               never leaves its registers (no LDS round trip, no cross-lane shuffles).
//   epilogue    O / l, bf16 pack, 8-byte stores; optional row log-sum-exp (the reference's unused O_l,
//               flashattention.cu:609).
//
// Algorithmic cost per (32 query rows x 64 keys) at D = 64: 16 MFMA (32 cycles each on one SIMD),
// 8 ds_read_b128 + 16 ds_read_b64_tr_b16, ~170 VALU/transcendental instructions.
//
// This file holds the phase-structured kernels (one 32/64-row block per wave, QK^T -> softmax -> PV per 64-key tile):
// fa_fwd_bf16_kernel (D = 128, causal D = 32) and its leaner four-waves-per-SIMD sibling fa_fwd_bf16_w4_kernel, plus the
// dispatcher.  The kernel used for D = 64 lives in fa_fwd_bf16_pipelined.hip.
#include "fa_bf16_common.h"
#include "fa_kernels.h"

namespace fa {

template <int D, int NWAVES, int QB, bool CAUSAL, bool OUT_F32, int MINWAVES>
__global__ __launch_bounds__(NWAVES* kWave, MINWAVES) void fa_fwd_bf16_kernel(FwdParams p)
{
    using C = Bf16Cfg<D, NWAVES>;
    constexpr int KS = D / 16;        // k-steps of S^T = K Q^T
    constexpr int DB = D / 32;        // 32-wide blocks of the head dim in O^T
    constexpr int KB = kKvBlk / 32;   // 32-key blocks per tile
    constexpr int BM = NWAVES * QB * 32;

    __shared__ __attribute__((aligned(1024))) char smem[2 * C::kStageBytes];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;

    // ---- work item: (slab, q tile); XCD-contiguous so one XCD's L2 serves all q tiles of a slab
    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles;
    int qt = w % p.q_tiles;
    if (CAUSAL) qt = p.q_tiles - 1 - qt;  // longest (latest) q tiles first
    const int n = p.n;
    const int q0 = qt * BM + wave * (QB * 32);

    const int b = slab / p.heads, h = slab % p.heads;
    const __bf16* qg = (const __bf16*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
    const __bf16* kg = (const __bf16*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const __bf16* vg = (const __bf16*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    const int64_t o_slab_off = b * p.o_batch_stride + h * p.o_head_stride;

    // ---- number of K/V tiles this workgroup walks
    int kv_end = n;
    if (CAUSAL) kv_end = min(n, qt * BM + BM);
    const int nt = (kv_end + kKvBlk - 1) / kKvBlk;

    issue_kv_tile<D, NWAVES>(kg, vg, 0, n, p.kv_row_stride, smem, wave, lane);

    // ---- Q fragments (B operand of S^T = K Q^T): lane (lq, hi) holds Q[q][16*ks + 8*hi .. +7]
    bf16x8 qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = min(q0 + qb * 32 + lq, n - 1);
        const __bf16* qr = qg + (int64_t)qrow * p.q_row_stride + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *(const bf16x8*)(qr + ks * 16);
    }

    f32x16 o[QB][DB];
    float m[QB], l[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m[qb] = -INFINITY;
        l[qb] = 0.0f;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.0f;
    }

    // ---- per-lane LDS read offsets
    // K fragment (A operand): key = 32*kb + lq, logical slot = 2*ks + hi  ->  physical slot = (2*ks) ^ (hi ^ swz)
    const int k_row_off = lq * C::kRowBytes;
    const int k_g = hi ^ k_swizzle<D>(lq);
    // V^T fragment: 16-lane group g = lane>>4 reads the [4 keys][16 cols] sub-tile (key group hi, col half g&1)
    const int li = lane & 15;
    const int v_lane_off = (hi * (D / 16) + ((lane >> 4) & 1)) * 128 + (li >> 2) * 32 + (li & 3) * 8;

    const float c = p.scale_log2e;

    for (int j = 0; j < nt; ++j) {
        // Tile j must have landed: hipcc only orders an LDS-DMA against LDS reads it cannot disambiguate, and a wave that
        // skips a tile (causal) issues no such read -- so every wave drains its own DMA queue explicitly, THEN the
        // barrier publishes all waves' pieces and proves everyone is done with the stage tile j+1 will overwrite.
        wait_lds_dma();
        __syncthreads();
        if (j + 1 < nt)
            issue_kv_tile<D, NWAVES>(kg, vg, (j + 1) * kKvBlk, n, p.kv_row_stride,
                                     smem + ((j + 1) & 1) * C::kStageBytes, wave, lane);
        const int kv0 = j * kKvBlk;
        if (CAUSAL && kv0 > q0 + QB * 32 - 1) continue;  // tile entirely above this wave's diagonal

        const char* ks_lds = smem + (j & 1) * C::kStageBytes;
        const char* vs_lds = ks_lds + C::kTileBytes;

        // ================= S^T = K Q^T =================
        f32x16 s[QB][KB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][kb][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int slot_off = ((2 * ks) ^ k_g) * 16;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const bf16x8 kf = *(const bf16x8*)(ks_lds + k_row_off + kb * 32 * C::kRowBytes + slot_off);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb)
                    s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][ks], s[qb][kb], 0, 0, 0);
            }
        }

        // ================= online softmax (registers only) =================
        const bool need_mask = (kv0 + kKvBlk > n) || (CAUSAL && (kv0 + kKvBlk - 1 > q0));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (need_mask) {
                const int qi = q0 + qb * 32 + lq;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + kb * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
                        const bool dead = (key >= n) || (CAUSAL && key > qi);
                        if (dead) s[qb][kb][r] = -INFINITY;
                    }
            }
            float mx = s[qb][0][0];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
            mx = xhalf_max(mx);
            // running max kept in the scaled (exp2) domain, nudged DOWN by <= 2 ulp so that c*s_max - m >= 0 exactly;
            // every p of the row carries the common factor 2^-m (cancels in O / l), and the hardware clamp on v_exp_f32
            // caps the row maximum (and anything within those 2 ulp of it) at exactly 1 -- no overflow for any input
            // magnitude (the iota known-answer workload of test.cu reaches |s| ~ 1e15) and no branch.
            float mc = mx * c;
            mc = fmaf(-fabsf(mc), 0x1p-23f, mc);
            const float m_new = fmaxf(m[qb], mc);
            const float alpha = fast_exp2(m[qb] - m_new);  // exp2(-inf) = 0 on the first tile
            m[qb] = m_new;
            float rs = 0.0f;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = exp2_clamp01(fmaf(s[qb][kb][r], c, -m_new));
                    s[qb][kb][r] = pv;
                    rs += pv;
                }
            l[qb] = fmaf(l[qb], alpha, rs);
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][db][r] *= alpha;
        }

        // ================= O^T += V^T P^T =================
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bf16x8 pf[QB];
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) pf[qb] = pack_bf16x8(s[qb][kb], 8 * t);
#pragma unroll
                for (int db = 0; db < DB; ++db) {
                    const int off0 = ((kb * 8 + 4 * t + 0) * (D / 16) + 2 * db) * 128;
                    const int off1 = ((kb * 8 + 4 * t + 2) * (D / 16) + 2 * db) * 128;
                    const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vs_lds + v_lane_off + off0));
                    const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(vs_lds + v_lane_off + off1));
                    const bf16x8 vf = __builtin_bit_cast(
                        bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qb], o[qb][db], 0, 0, 0);
                }
            }
    }

    // ================= epilogue: O / l, pack, store =================
    mfma_drain();  // the loop exit is a branch: the last P.V MFMAs may still be in flight
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float lt = xhalf_sum(l[qb]);
        const float inv = 1.0f / lt;
        const int qi = q0 + qb * 32 + lq;
        if (qi < n) {
            const int64_t o_off = o_slab_off + (int64_t)qi * p.o_row_stride + 4 * hi;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (OUT_F32) {
                        f32x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = o[qb][db][4 * g + e] * inv;
                        *(f32x4*)((float*)p.o + o_off + db * 32 + 8 * g) = pk;
                    } else {
                        bf16x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(o[qb][db][4 * g + e] * inv);
                        *(bf16x4*)((__bf16*)p.o + o_off + db * 32 + 8 * g) = pk;
                    }
                }
            if (p.lse != nullptr && hi == 0)
                p.lse[(int64_t)slab * n + qi] = (m[qb] + __builtin_amdgcn_logf(lt)) * kLn2;
        }
    }
}

// =====================================================================================================================
// "w4": one 32-row block per wave, four waves per SIMD (<= 128 VGPRs), phase-structured like fa_fwd_bf16_kernel but on the
// same VALU diet as pp2/pp3 (matrix-core row sums, lazily rescaled accumulator, lane-local rescale test, v_max3 without
// canonicalisation).  Overlap of MFMA and VALU is left to the four co-resident waves.
// =====================================================================================================================
template <int D>
__device__ __forceinline__ void softmax_block3(f32x16 (&s)[2], BlockState& st, f32x16 (&o)[D / 32], bf16x8 (&pf)[4], float c)
{
    // lane-local maximum of the 32 scores (no cross-half exchange unless the rare rescale fires)
    float p0 = max3_safe(s[0][0], s[0][1], s[0][2]), p1 = max3_safe(s[0][3], s[0][4], s[0][5]);
    float p2 = max3_safe(s[0][6], s[0][7], s[0][8]), p3 = max3_safe(s[0][9], s[0][10], s[0][11]);
    p0 = max3_safe(p0, s[0][12], s[0][13]);
    p1 = max3_safe(p1, s[0][14], s[0][15]);
    p2 = max3_safe(p2, s[1][0], s[1][1]);
    p3 = max3_safe(p3, s[1][2], s[1][3]);
    p0 = max3_safe(p0, s[1][4], s[1][5]);
    p1 = max3_safe(p1, s[1][6], s[1][7]);
    p2 = max3_safe(p2, s[1][8], s[1][9]);
    p3 = max3_safe(p3, s[1][10], s[1][11]);
    p0 = max3_safe(p0, s[1][12], s[1][13]);
    p1 = max3_safe(p1, s[1][14], s[1][15]);
    const float lm = fmaxf(max3_safe(p0, p1, p2), p3);
    if (__builtin_expect(__any(fmaf(lm, c, -st.m) > kLazyThr), 0)) {
        asm volatile("; lazy rescale" ::: "memory");
        mfma_drain();  // the accumulators rescaled below may have an MFMA in flight (hazard not padded across the branch)
        float mc = xhalf_max(lm) * c;
        mc = fmaf(-fabsf(mc), 0x1p-23f, mc);
        const float m_new = fmaxf(st.m, mc);
        const float alpha = fast_exp2(st.m - m_new);
        st.m = m_new;
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) st.lacc[r] *= alpha;
    }
    const float off = st.m + kLazyThr;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = exp2_clamp01(fmaf(s[kb][r], c, -off));
        pf[2 * kb] = pack_bf16x8(s[kb], 0);
        pf[2 * kb + 1] = pack_bf16x8(s[kb], 8);
    }
}

template <int D, int NWAVES, bool CAUSAL, bool OUT_F32, int MINWAVES>
__global__ __launch_bounds__(NWAVES* kWave, MINWAVES) void fa_fwd_bf16_w4_kernel(FwdParams p)
{
    using C = Bf16Cfg<D, NWAVES>;
    constexpr int KS = D / 16, DB = D / 32;
    constexpr int BM = NWAVES * 32;

    __shared__ __attribute__((aligned(1024))) char smem[2 * C::kStageBytes];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;

    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles;
    int qt = w % p.q_tiles;
    if (CAUSAL) qt = p.q_tiles - 1 - qt;
    const int n = p.n;
    const int q0 = qt * BM + wave * 32;

    const int b = slab / p.heads, h = slab % p.heads;
    const __bf16* qg = (const __bf16*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
    const __bf16* kg = (const __bf16*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const __bf16* vg = (const __bf16*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    const int64_t o_slab_off = b * p.o_batch_stride + h * p.o_head_stride;

    int kv_end = n;
    if (CAUSAL) kv_end = min(n, qt * BM + BM);
    const int nt = (kv_end + kKvBlk - 1) / kKvBlk;

    issue_kv_tile<D, NWAVES>(kg, vg, 0, n, p.kv_row_stride, smem, wave, lane);

    bf16x8 qf[KS];
    {
        const __bf16* qr = qg + (int64_t)min(q0 + lq, n - 1) * p.q_row_stride + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qr + ks * 16);
    }
    const bf16x8 ones_a = rowsum_a_operand(lane);

    f32x16 o[DB];
    BlockState st;
    st.m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) st.lacc[r] = 0.0f;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;

    const int k_row_off = lq * C::kRowBytes;
    const int k_g = hi ^ k_swizzle<D>(lq);
    const int li = lane & 15;
    const int v_lane_off = (hi * (D / 16) + ((lane >> 4) & 1)) * 128 + (li >> 2) * 32 + (li & 3) * 8;
    const float c = p.scale_log2e;

    for (int j = 0; j < nt; ++j) {
        wait_lds_dma();
        __syncthreads();
        if (j + 1 < nt)
            issue_kv_tile<D, NWAVES>(kg, vg, (j + 1) * kKvBlk, n, p.kv_row_stride, smem + ((j + 1) & 1) * C::kStageBytes, wave, lane);
        const int kv0 = j * kKvBlk;
        if (CAUSAL && kv0 > q0 + 31) continue;  // tile entirely above this wave's diagonal
        const char* k_lds = smem + (j & 1) * C::kStageBytes;
        const char* v_lds = k_lds + C::kTileBytes;

        f32x16 s[2];
        bf16x8 pf[4];
        qk_block<D>(k_lds, k_row_off, k_g, qf, s);
        if ((kv0 + kKvBlk > n) || (CAUSAL && (kv0 + kKvBlk - 1 > q0))) mask_block<CAUSAL>(s, kv0, q0 + lq, n, hi);
        softmax_block3<D>(s, st, o, pf, c);
        pv_block<D>(v_lds, v_lane_off, pf, o);
        sum_block(ones_a, pf, st);
    }

    mfma_drain();  // the loop exit is a branch: the last P.V / row-sum MFMAs may still be in flight
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
        if (p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = (st.m + kLazyThr + __builtin_amdgcn_logf(lt)) * kLn2;
    }
}

template <int D, int NWAVES, int MINWAVES>
static hipError_t launch_w4(const FwdParams& p0, int causal, int out_f32, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = NWAVES * 32;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)total), block(NWAVES * kWave);
    if (causal) {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_w4_kernel<D, NWAVES, true, true, MINWAVES>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_w4_kernel<D, NWAVES, true, false, MINWAVES>), grid, block, 0, stream, p);
    } else {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_w4_kernel<D, NWAVES, false, true, MINWAVES>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_w4_kernel<D, NWAVES, false, false, MINWAVES>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

// MINWAVES_C: occupancy hint of the causal instantiations (D = 64 at 4 waves per SIMD spilled 40 bytes per lane in the mask code)
template <int D, int NWAVES, int QB, int MINWAVES, int MINWAVES_C = MINWAVES>
static hipError_t launch_cfg(const FwdParams& p0, int causal, int out_f32, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = NWAVES * QB * 32;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)total), block(NWAVES * kWave);
    if (causal) {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_kernel<D, NWAVES, QB, true, true, MINWAVES_C>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_kernel<D, NWAVES, QB, true, false, MINWAVES_C>), grid, block, 0, stream, p);
    } else {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_kernel<D, NWAVES, QB, false, true, MINWAVES>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_kernel<D, NWAVES, QB, false, false, MINWAVES>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

// ---- product dispatch (variant 0) --------------------------------------------------------------------------------------
// Measured on MI355X, steady clocks, TFLOP/s at bh = 16, n = 8192 (profiles/): d = 64 non-causal  x4 1044 / pipelined 1022;
// d = 64 causal  x2 884 / pipelined 2-wave 738 / 4-wave 642 / x4 594;  d = 128  x2 1254 / w4 1012 / phase-structured 950 (causal:
// x2 1109 / phase-structured 827);  d = 32  x2 780 / pipelined 767.
enum Bf16Choice { kChoosePhase, kChoosePipelined4, kChoosePipelined2, kChooseX4, kChooseW4, kChooseX2D128, kChooseX2D64, kChooseX2D32 };

static Bf16Choice choose_bf16(int64_t bh, int64_t n, int d, int causal, bool addressable)
{
    // Rows of two stages: the phase-structured kernel (128-row workgroups, nothing to fill or drain).  ms at BH x N = 131072 rows,
    // phase-structured / one-wave-per-SIMD (profiles/r03_short_rows.txt): N = 128 d=64 0.017 / 0.021 (causal 0.017 / 0.024), d=32 0.008 /
    // 0.013 (0.009 / 0.016), d=128 0.033 / 0.044 (0.032 / 0.052); d=128 causal N = 256 0.037 / 0.044, N = 512 0.050 / 0.055; at N = 256
    // the other head dims tie, from N = 512 on the pipelined kernels lead.
    if (n <= 128 || (d == 128 && causal && n <= 512)) return kChoosePhase;
    if (d == 128) {
        // one wave per SIMD, 256-row workgroups, one workgroup per CU: needs enough workgroups to occupy the CUs
        if (addressable && bh * ((n + 255) / 256) >= 128) return kChooseX2D128;
        return causal ? kChoosePhase : kChooseW4;
    }
    if (!addressable) return kChoosePhase;
    const int64_t items256 = bh * ((n + 255) / 256), items512 = bh * ((n + 511) / 512);
    // Grids of at most 128 tiles of 256 rows leave half the CUs idle under the one-wave-per-SIMD kernels: the phase-structured kernel's
    // 128-row workgroups are twice as many.  ms phase-structured / NB = 2 (profiles/r03_short_rows.txt, second part), d = 64, BH x N:
    // 8 x 1024 0.0166 / 0.0188 (causal 0.0182 / 0.0236), 32 x 1024 0.0194 / 0.0207 (0.0198 / 0.0246), 8 x 2048 0.0302 / 0.0320
    // (0.0326 / 0.0367), 4 x 3072 0.0423 / 0.0442, 16 x 512 causal 0.0111 / 0.0171; 16 x 2048 (128 tiles) 0.0342 / 0.0345, 32 x 1536 (192)
    // 0.0389 / 0.0306; d = 32: 8 x 1024 0.0123 / 0.0146 (0.0143 / 0.0193), 16 x 1536 0.0187 / 0.0207.  Rows of 4096 keys and more on such
    // grids are key-split launches of the NB = 2 kernel (fa_plan.cpp, fa_launch.cpp).
    if (n < 4096 && items256 <= 128) return kChoosePhase;
    // d = 32 (TFLOP/s, x2 / pipelined / phase-structured): 16 x 8192 non-causal 780 / 767 / -, causal 428 / 415 / 416; 128 x 8192
    // causal 709 / - / 560
    if (d == 32) return kChooseX2D32;
    // d == 64
    if (causal) {
        // With every workgroup resident at once (two-wave kernel) a causal launch lasts as long as its heaviest pair of tiles.  The
        // one-wave-per-SIMD kernel with 256-row tiles runs one or two workgroups per CU in an order chosen for the grid (heavy tiles
        // first, or heavy + light pairs: xn_launch_order in fa_bf16_xn_kernel.h).  Measured (ms, two-wave / x2), BH x N:
        // 16 x 8192: 0.206 / 0.148, 12 x 8192: 0.196 / 0.145, 8 x 16384: 0.307 / 0.279, 16 x 4096: 0.070 / 0.066, 32 x 4096:
        // 0.113 / 0.084, 32 x 8192: 0.303 / 0.290, 64 x 4096: 0.185 / 0.184, 64 x 8192: 0.568 / 0.563, 128 x 8192: 1.078 / 1.069;
        // short rows, 128 x 2048: 0.115 / 0.115, 128 x 1024: 0.040 / 0.040 -- those stay where they were.
        if (items256 <= 512 || n >= 4096) return kChooseX2D64;
        return kChoosePipelined4;
    }
    // Non-causal.  Rounds of workgroups, in units of 131072 query rows of work:
    //   x4   512-row workgroups, one per CU, ~10 % faster per row at full occupancy; a partly filled round costs a whole one;
    //   pp3  256-row workgroups, two per CU; a last round with at most one workgroup per CU runs in ~0.71 of a round (a lone
    //        wave gets 1 / 1.41 of a SIMD pair's throughput);
    //   x2   256-row workgroups, one per CU: 2-4 % ahead of pp3 when there is at most one round of them.
    // Measured (TFLOP/s, pp3 / x2 / x4), BH x N: 8 x 8192: 963 / 962 / 730, 12 x 8192: 919 / 842 / 996, 16 x 8192: 1060 / 1056 /
    // 1166, 24 x 8192: 1075 / 1034 / 975, 16 x 4096: 849 / 876 / 662, 32 x 2048: 767 / 790 / 587, 64 x 1024: 649 / 654 / 505.
    const double pad512 = (double)(((n + 511) / 512) * 512) / (double)n, pad256 = (double)(((n + 255) / 256) * 256) / (double)n;
    const int64_t rem = items256 % 512;
    const double cost_pp3 = ((double)(items256 / 512) + (rem == 0 ? 0.0 : rem <= 256 ? 0.71 : 1.0)) * pad256;
    const double cost_x4 = (double)((items512 + 255) / 256) / 1.10 * pad512;
    if (n >= 2048 && cost_x4 <= cost_pp3) return kChooseX4;
    if (items256 <= 256) return kChooseX2D64;
    // Round 3: the NB = 2 kernel takes its exponent reference from 192 sampled keys on rows of 4096 keys and more (more zero operands, more
    // clock: DESIGN.md 4.6), the two-wave kernel does not -- on long rows it is now 2-3 % ahead of pp3 at every partly filled round
    // (ms pp3 / x2, BH x N: 24 x 8192 0.407 / 0.399, 40 x 8192 0.635 / 0.619, 16 x 12288 0.579 / 0.561, 48 x 12288 0.649 / 0.607, 12 x 16384
    // 0.762 / 0.736, 20 x 16384 1.228 / 1.199; at 4096 keys they alternated: 12 x 4096 0.078 / 0.075, 24 x 4096 0.139 / 0.134, 32 x 4096 0.164 / 0.168)
    // Round 6: the NB = 2 kernel re-centres that reference on the row sum and no longer pays for sampling it; from 4096 keys on it now leads at
    // every partly filled round of the sweep (ms pp3 / x2: 40 x 4096 0.208 / 0.201, 48 x 4096 0.201 / 0.192, 24 x 6144 0.258 / 0.247, 32 x 6144
    // 0.283 / 0.270, 48 x 6144 0.430 / 0.413; at 3072 keys they still alternate: 48 x 3072 0.137 / 0.140, 64 x 3072 0.160 / 0.153)
    if (n >= 4096) return kChooseX2D64;
    return kChoosePipelined4;
}

const char* bf16_kernel_name(int64_t bh, int64_t n, int d, int causal)
{
    switch (choose_bf16(bh, n, d, causal, true)) {
        case kChooseX4: return "fa_fwd_bf16_x4_kernel";
        case kChoosePipelined4:
        case kChoosePipelined2: return "fa_fwd_bf16_pp3_kernel";
        case kChooseW4: return "fa_fwd_bf16_w4_kernel";
        case kChooseX2D128:
        case kChooseX2D64:
        case kChooseX2D32: return "fa_fwd_bf16_x2_kernel";
        default: return "fa_fwd_bf16_kernel";
    }
}

// the two-term-P kernels exist in the one-wave-per-SIMD family only (32-bit byte offsets into a slab): NB = 4 where the bf16-P dispatch
// would take it too (large non-causal d = 64 grids), NB = 2 everywhere else
bool bf16_p16_supported(const FwdParams& p, int d)
{
    return (d == 32 || d == 64 || d == 128) && ((int64_t)(p.n - 1) * p.kv_row_stride + d) * 2 < (int64_t)0xffffffffLL;
}

#if FA_ABLATION   // the fp16-P families (csrc/experiments/): replaced by P as two bf16 terms in round 4, kept for A/B runs

// d = 64 fp16 P: the NB = 4 kernel (512-row workgroups, one per CU) only where it is not behind -- a single, well filled round of
// them.  Elsewhere the NB = 2 kernel, which fits a CU twice: ms NB = 4 / NB = 2, BH x 8192: 16: 0.279 / 0.278, 12: 0.260 / 0.251,
// 32: 0.538 / 0.530, 128: 2.122 / 2.056.
bool bf16_p16_uses_x4(int64_t bh, int64_t n, int causal)
{
    const int64_t items512 = bh * ((n + 511) / 512);
    return choose_bf16(bh, n, 64, causal, true) == kChooseX4 && items512 > 192 && items512 <= 256;
}

hipError_t launch_bf16_p16(const FwdParams& p, int d, int causal, int out_f32, hipStream_t stream)
{
    if (!bf16_p16_supported(p, d)) return hipErrorInvalidValue;
    if (d == 32) return launch_bf16_x2_p16_d32(p, causal, out_f32, stream);
    if (d == 128) return launch_bf16_x2_p16_d128(p, causal, out_f32, stream);
    if (bf16_p16_uses_x4(p.bh, p.n, causal)) return launch_bf16_x4_p16(p, causal, out_f32, stream);
    return launch_bf16_x2_p16_d64(p, causal, out_f32, stream);
}

// Two fp16 terms of P: the NB = 2 kernel at every grid size.  With twice the P.V and row-sum MFMAs the loop is bound by instruction issue
// at a clock near the chip's maximum (2.3 GHz, matrix pipe 54 % busy: profiles/r03a_rocprof_summary.txt, pass pmc_acc_sq), and two
// resident workgroups per CU beat one NB = 4 workgroup wherever both were measured (ms NB = 4 / NB = 2, BH x 8192 x 64: 16: 0.374 / 0.352,
// 12: 0.37 / 0.337; 22.0 us per slab in full rounds against 23.4).
hipError_t launch_bf16_p16x2(const FwdParams& p, int d, int causal, int out_f32, hipStream_t stream)
{
    if (!bf16_p16_supported(p, d)) return hipErrorInvalidValue;
    if (d == 32) return launch_bf16_x2_p16x2_d32(p, causal, out_f32, stream);
    if (d == 128) return launch_bf16_x2_p16x2_d128(p, causal, out_f32, stream);
    return launch_bf16_x2_p16x2_d64(p, causal, out_f32, stream);
}

#endif   // FA_ABLATION

// P as bf16 hi + bf16 lo (round 4): NB = 4 on the grids the bf16-P dispatch gives 512-row workgroups, NB = 2 elsewhere.
// variant 0 = the dispatch, 1 = NB = 2 forced, 2 = NB = 4 forced (d = 64, non-causal)
bool bf16_pb2_uses_x4(int64_t bh, int64_t n, int causal) { return choose_bf16(bh, n, 64, causal, true) == kChooseX4; }

hipError_t launch_bf16_pb2(const FwdParams& p, int d, int causal, int out_f32, int variant, hipStream_t stream)
{
    if (!bf16_p16_supported(p, d) || variant < 0 || variant > 2) return hipErrorInvalidValue;
    if (d == 32) return out_f32 ? launch_bf16_x2_pb2_d32_f32out(p, causal, stream) : launch_bf16_x2_pb2_d32_bf16out(p, causal, stream);
    if (d == 128) return out_f32 ? launch_bf16_x2_pb2_d128_f32out(p, causal, stream) : launch_bf16_x2_pb2_d128_bf16out(p, causal, stream);
    if (variant == 2 || (variant == 0 && bf16_pb2_uses_x4(p.bh, p.n, causal)))
        return out_f32 ? launch_bf16_x4_pb2_f32out(p, causal, stream) : launch_bf16_x4_pb2_bf16out(p, causal, stream);
    return out_f32 ? launch_bf16_x2_pb2_d64_f32out(p, causal, stream) : launch_bf16_x2_pb2_d64_bf16out(p, causal, stream);
}

hipError_t launch_fwd_bf16(const FwdParams& p, int d, int causal, int out_f32, int variant, hipStream_t stream)
{
    if (variant == 0) {
        const bool addressable = ((int64_t)(p.n - 1) * p.kv_row_stride + d) * 2 < (int64_t)0xffffffffLL;
        switch (choose_bf16(p.bh, p.n, d, causal, addressable)) {
            case kChooseX4: return launch_bf16_x4(p, causal, out_f32, 2, stream);
            case kChoosePipelined4: return launch_bf16_pipelined(p, d, 4, causal, out_f32, 0, stream);
            case kChoosePipelined2: return launch_bf16_pipelined(p, d, 2, causal, out_f32, 0, stream);
            case kChooseW4: return launch_w4<128, 4, 2>(p, causal, out_f32, stream);
            case kChooseX2D128: return launch_bf16_x2(p, 128, causal, out_f32, 0, stream);
            case kChooseX2D64: return launch_bf16_x2(p, 64, causal, out_f32, 0, stream);
            case kChooseX2D32: return launch_bf16_x2(p, 32, causal, out_f32, 0, stream);
            default:
                if (d == 32) return launch_cfg<32, 4, 1, 4>(p, causal, out_f32, stream);
                if (d == 64) return launch_cfg<64, 4, 1, 4, 3>(p, causal, out_f32, stream);
                return launch_cfg<128, 4, 1, 2>(p, causal, out_f32, stream);
        }
    }
    if (!bf16_pipelined_supported(p, d) && d != 128) return hipErrorInvalidValue;  // the ablation variants assume 32-bit slab offsets
    switch (d) {
        case 32:
            if (variant == 50) return launch_bf16_x2(p, 32, causal, out_f32, 0, stream);
            if (variant == 52) return launch_bf16_x2(p, 32, causal, out_f32, 3, stream);
#if FA_ABLATION
            if (variant == 70) return launch_bf16_x2(p, 32, causal, out_f32, 40, stream);   // cycle-stamped NB = 2 kernel (non-causal)
#endif
            if (variant == 24) return launch_bf16_pipelined(p, 32, 2, causal, out_f32, 0, stream);
            // lockstep/pipelined kernel for the non-causal case; its 256-row workgroups waste more of the causal
            // triangle than the 128-row phase-structured kernel recovers at D = 32
            if (variant == 7) return launch_bf16_pipelined(p, 32, 4, causal, out_f32, 0, stream);
            if (variant == 1) return launch_cfg<32, 4, 1, 4>(p, causal, out_f32, stream);   // phase-structured
            return hipErrorInvalidValue;
        case 64:
            switch (variant) {
                case 1: return launch_cfg<64, 4, 1, 4, 3>(p, causal, out_f32, stream);   // phase-structured, 4 waves/SIMD
#if FA_ABLATION
                case 2: return launch_cfg<64, 4, 2, 2>(p, causal, out_f32, stream);   // phase-structured, 64 rows/wave
#endif
                case 7: return launch_bf16_pipelined(p, 64, 4, causal, out_f32, 0, stream);
                case 24: return launch_bf16_pipelined(p, 64, 2, causal, out_f32, 0, stream);
#if FA_ABLATION
                case 22: return launch_bf16_pipelined(p, 64, 4, 0, 0, 1, stream);   // in-kernel phase timers (written over the lse buffer)
#endif
                case 25: return launch_bf16_pipelined(p, 64, 4, causal, out_f32, 3, stream);  // lazily rescaled mix only
                case 26: return launch_bf16_pipelined(p, 64, 2, causal, out_f32, 3, stream);  // same, 2-wave workgroups
                case 30: return launch_bf16_x4(p, causal, out_f32, 2, stream);
                case 31: return launch_bf16_x4(p, causal, out_f32, 1, stream);
                case 42: return launch_bf16_x4(p, causal, out_f32, 3, stream);   // x4, rescaling mix only
#if FA_ABLATION
                case 53: return launch_bf16_x2(p, 64, causal, out_f32, 8, stream);    // eight waves per workgroup
                case 70: return launch_bf16_x2(p, 64, causal, out_f32, 40, stream);   // cycle-stamped NB = 2 kernel
                case 71: return launch_bf16_x2(p, 64, causal, out_f32, 41, stream);   // ... in the product's causal launch order, with a timeline
#endif
                case 50: return launch_bf16_x2(p, 64, causal, out_f32, 0, stream);   // one wave per SIMD, 64 rows per wave, 256-row tiles
                case 51: return launch_bf16_x2(p, 64, causal, out_f32, 1, stream);
                case 52: return launch_bf16_x2(p, 64, causal, out_f32, 3, stream);
#if FA_ABLATION
                case 33: return launch_bf16_x4(p, causal, out_f32, 11, stream);  // x4 timing ablations (results are garbage)
                case 34: return launch_bf16_x4(p, causal, out_f32, 12, stream);
                case 35: return launch_bf16_x4(p, causal, out_f32, 13, stream);
                case 36: return launch_bf16_x4(p, causal, out_f32, 14, stream);
                case 37: return launch_bf16_x4(p, causal, out_f32, 15, stream);
                case 38: return launch_bf16_x4(p, causal, out_f32, 16, stream);
                case 39: return launch_bf16_x4(p, causal, out_f32, 17, stream);
                case 40: return launch_bf16_x4(p, causal, out_f32, 18, stream);
                case 41: return launch_bf16_x4(p, causal, out_f32, 19, stream);
                case 43: return launch_bf16_x4(p, causal, out_f32, 20, stream);
                case 44: return launch_bf16_x4(p, causal, out_f32, 21, stream);
                case 45: return launch_bf16_x4(p, causal, out_f32, 22, stream);
                case 46: return launch_bf16_x4(p, causal, out_f32, 23, stream);
                case 47: return launch_bf16_x4(p, causal, out_f32, 24, stream);
                case 60: case 61: case 62: case 63: case 64: case 65: case 66: case 67: case 68:
                    return launch_bf16_x4(p, causal, out_f32, variant - 30, stream);   // cycle-stamped forms (fa_driver_ablation --mode prof4)
#endif
#if FA_ABLATION
                case 10: return launch_w4<64, 4, 4>(p, causal, out_f32, stream);       // 4 waves/SIMD on the VALU diet (a null result of DESIGN.md section 4; its causal form spills)
#endif
#if FA_ABLATION
                default: return launch_bf16_pp2(p, causal, out_f32, variant, stream);  // 9 = pp2 (previous generation), 6, 11..21 = its ablations
#else
                default: return hipErrorInvalidValue;   // not a shipped tiling (timing-only ablations live in the ablation build)
#endif
            }
        case 128:
            if (variant == 50) return launch_bf16_x2(p, 128, causal, out_f32, 0, stream);   // one wave per SIMD, explicit register files
            if (variant == 51) return launch_bf16_x2(p, 128, causal, out_f32, 1, stream);
            if (variant == 52) return launch_bf16_x2(p, 128, causal, out_f32, 3, stream);
#if FA_ABLATION
            if (variant == 53) return launch_bf16_x2(p, 128, causal, out_f32, 12, stream);
#endif
            if (variant == 10) return launch_w4<128, 4, 2>(p, causal, out_f32, stream);
#if FA_ABLATION
            if (variant == 23) return launch_w4<128, 4, 3>(p, causal, out_f32, stream);
#endif
            if (variant == 1) return launch_cfg<128, 4, 1, 2>(p, causal, out_f32, stream);  // phase-structured (the pipelined kernel needs > 256 VGPRs at D = 128)
            return hipErrorInvalidValue;
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fa
