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
#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

constexpr int kKvBlk = 64;  // keys per K/V tile
constexpr float kFmaExpLimit = 1024.0f;  // |scale*log2e * rowmax| above which the exponent is formed subtract-first

// XOR applied to the 16-byte slot index of K-image row `row` (see header comment).
template <int D>
__device__ __forceinline__ int k_swizzle(int row)
{
    constexpr int S = D / 8;                       // 16-byte slots per row
    constexpr int R = (S >= 16) ? 1 : 16 / S;      // rows per 256-byte LDS bank row
    constexpr int M = (S >= 16) ? 15 : S - 1;
    return (row / R) & M;
}

template <int D, int NWAVES>
struct Bf16Cfg {
    static constexpr int kRowBytes = 2 * D;
    static constexpr int kTileBytes = kKvBlk * kRowBytes;       // one K (or V) tile
    static constexpr int kStageBytes = 2 * kTileBytes;          // K + V
    static constexpr int kChunks = kTileBytes / 1024;           // 1 KiB DMA pieces per tile
    static constexpr int kChunksPerWave = kChunks / NWAVES;
    static_assert(kChunks % NWAVES == 0, "tile must split evenly over the waves");
};

// Enqueue the LDS-DMA for K/V tile starting at key kv0 into `stage` (wave-uniform LDS address).
template <int D, int NWAVES>
__device__ __forceinline__ void issue_kv_tile(const __bf16* __restrict__ kg, const __bf16* __restrict__ vg,
                                              int kv0, int n, int row_stride, char* stage, int wave, int lane)
{
    using C = Bf16Cfg<D, NWAVES>;
#pragma unroll
    for (int i = 0; i < C::kChunksPerWave; ++i) {
        const int ch = wave + i * NWAVES;
        // ---- K: row-major, slot-swizzled
        {
            const int off = ch * 1024 + lane * 16;
            const int row = off / C::kRowBytes;
            const int phys = (off % C::kRowBytes) / 16;
            const int slot = phys ^ k_swizzle<D>(row);
            const int grow = min(kv0 + row, n - 1);
            const __bf16* src = kg + (int64_t)grow * row_stride + slot * 8;
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(stage + ch * 1024), 16, 0, 0);
        }
        // ---- V: [key/4][col/16][4][16] sub-tiles (128 bytes each)
        {
            const int blk = ch * 8 + lane / 8;
            const int kg4 = blk / (D / 16), cb = blk % (D / 16);
            const int key = kg4 * 4 + (lane % 8) / 2;
            const int col = cb * 16 + (lane & 1) * 8;
            const int grow = min(kv0 + key, n - 1);
            const __bf16* src = vg + (int64_t)grow * row_stride + col;
            __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(stage + C::kTileBytes + ch * 1024), 16, 0, 0);
        }
    }
}

__device__ __forceinline__ bf16x8 pack_bf16x8(const f32x16& s, int base)
{
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)s[base + i];
    return r;
}

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
            // running max in the RAW score domain; exponent = c * (s - m)
            const float m_new = fmaxf(m[qb], mx);
            const float alpha = fast_exp2((m[qb] - m_new) * c);  // exp2(-inf) = 0 on the first tile
            m[qb] = m_new;
            const float mc = m_new * c;
            float rs = 0.0f;
            if (__builtin_expect(__any(fabsf(mc) > kFmaExpLimit), 0)) {
                // enormous scores (e.g. the iota known-answer workload of test.cu): the rounding of c*m would no longer
                // cancel inside fma(s, c, -c*m); subtract first, exactly
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = fast_exp2((s[qb][kb][r] - m_new) * c);
                        s[qb][kb][r] = pv;
                        rs += pv;
                    }
            } else {
                // one fma + one exp per score; the single rounding of c*m perturbs the exponent by <= ulp(c*m)/2
                // (<= 2^-14 at the branch limit, 2^-19 at |c*m| ~ 50): fp32-rounding class
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pv = fast_exp2(fmaf(s[qb][kb][r], c, -mc));
                        s[qb][kb][r] = pv;
                        rs += pv;
                    }
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
                p.lse[(int64_t)slab * n + qi] = m[qb] * p.scale + __builtin_amdgcn_logf(lt) * kLn2;
        }
    }
}

template <int D, int NWAVES, int QB, int MINWAVES>
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
            hipLaunchKernelGGL((fa_fwd_bf16_kernel<D, NWAVES, QB, true, true, MINWAVES>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_kernel<D, NWAVES, QB, true, false, MINWAVES>), grid, block, 0, stream, p);
    } else {
        if (out_f32)
            hipLaunchKernelGGL((fa_fwd_bf16_kernel<D, NWAVES, QB, false, true, MINWAVES>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((fa_fwd_bf16_kernel<D, NWAVES, QB, false, false, MINWAVES>), grid, block, 0, stream, p);
    }
    return hipGetLastError();
}

hipError_t launch_fwd_bf16(const FwdParams& p, int d, int causal, int out_f32, int variant, hipStream_t stream)
{
    switch (d) {
        case 32: return launch_cfg<32, 4, 1, 4>(p, causal, out_f32, stream);
        case 64:
            if (variant == 1) return launch_cfg<64, 4, 2, 2>(p, causal, out_f32, stream);
            return launch_cfg<64, 4, 1, 4>(p, causal, out_f32, stream);
        case 128: return launch_cfg<128, 4, 1, 2>(p, causal, out_f32, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fa
