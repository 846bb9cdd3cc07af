// fa_fwd_f32.hip -- fused flash-attention forward in exact fp32 (the reference's dtype) for gfx950.
//
// Replaces flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:139-579) for fp32 tensors.  Both
// contractions run on v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, bit-for-bit a k-ordered fmaf chain, at the
// fp32 vector peak (157 TF) while leaving the VALU free for the softmax.  Same skeleton as fa_fwd_bf16.hip:
//
//   workgroup   NWAVES waves x 32 query rows; K/V tiles of 32 keys, LDS-DMA double buffered, one barrier per tile.
//   K image     row-major [key][D] fp32, 16-byte slots XOR-swizzled per row -> conflict-free ds_read_b128 of
//               4 consecutive head-dim values per lane (used by 4 consecutive MFMAs).
//   V image     row-major [key][D] fp32, read column-wise with ds_read_b32 (32 consecutive floats per half-wave).
//   S^T = K Q^T swapped product: lane (q = lane&31, hi) holds scores of keys 4*hi + (r&3) + 8*(r>>2), r = 0..15.
//   O^T += V^T P^T   MFMA #r of a tile contracts over exactly the key pair {r-th key of hi=0, r-th key of hi=1}, so the
//               fp32 P values are fed to the matrix core straight from the registers they were exponentiated in.
//
// The head-dim contraction order of S is d = 8g + 4*hi + e (g-major), i.e. a permutation of the reference's
// d = 0..63 loop (flashattention.cu:236-252); results agree to fp32 rounding, not bitwise.
#include "fa_common.h"
#include "fa_kernels.h"

namespace fa {

typedef __attribute__((address_space(3))) void lds_void_f;
typedef __attribute__((address_space(1))) const void gbl_cvoid_f;

constexpr int kKvBlkF32 = 32;

template <int D>
__device__ __forceinline__ int k_swizzle_f32(int row)
{
    constexpr int RB = 4 * D;
    constexpr int S = RB / 16;
    constexpr int R = (RB >= 256) ? 1 : 256 / RB;
    constexpr int M = (S >= 16) ? 15 : S - 1;
    return (row / R) & M;
}

template <int D, int NWAVES>
struct F32Cfg {
    static constexpr int kRowBytes = 4 * D;
    static constexpr int kTileBytes = kKvBlkF32 * kRowBytes;
    static constexpr int kStageBytes = 2 * kTileBytes;
    static constexpr int kChunks = kTileBytes / 1024;
    static constexpr int kChunksPerWave = kChunks / NWAVES;
    static_assert(kChunks % NWAVES == 0, "tile must split evenly over the waves");
};

template <int D, int NWAVES>
__device__ __forceinline__ void issue_kv_tile_f32(const float* __restrict__ kg, const float* __restrict__ vg,
                                                  int kv0, int n, int row_stride, char* stage, int wave, int lane)
{
    using C = F32Cfg<D, NWAVES>;
#pragma unroll
    for (int i = 0; i < C::kChunksPerWave; ++i) {
        const int ch = wave + i * NWAVES;
        const int off = ch * 1024 + lane * 16;
        const int row = off / C::kRowBytes;
        const int phys = (off % C::kRowBytes) / 16;
        const int grow = min(kv0 + row, n - 1);
        {
            const int slot = phys ^ k_swizzle_f32<D>(row);
            const float* src = kg + (int64_t)grow * row_stride + slot * 4;
            __builtin_amdgcn_global_load_lds((gbl_cvoid_f*)src, (lds_void_f*)(stage + ch * 1024), 16, 0, 0);
        }
        {
            const float* src = vg + (int64_t)grow * row_stride + phys * 4;
            __builtin_amdgcn_global_load_lds((gbl_cvoid_f*)src, (lds_void_f*)(stage + C::kTileBytes + ch * 1024), 16, 0, 0);
        }
    }
}

template <int D, int NWAVES, bool CAUSAL, int MINWAVES>
__global__ __launch_bounds__(NWAVES* kWave, MINWAVES) void fa_fwd_f32_kernel(FwdParams p)
{
    using C = F32Cfg<D, NWAVES>;
    constexpr int G = D / 8;    // ds_read_b128 groups per key row half (4 floats each)
    constexpr int DB = D / 32;  // 32-wide head-dim blocks of O^T
    constexpr int BM = NWAVES * 32;

    __shared__ __attribute__((aligned(1024))) char smem[2 * C::kStageBytes];

    if (flag_says_skip(p)) return;   // conditional fallback behind the guarded split kernel (FA_KERNEL_AUTO, fa_api.cpp)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lq = lane & 31, hi = lane >> 5;

    const int total = p.bh * p.q_tiles;
    const int w = xcd_remap(blockIdx.x, total);
    const int slab = w / p.q_tiles;
    int qt = w % p.q_tiles;
    if (CAUSAL) qt = causal_tile(p, qt);
    const int n = p.n;
    const int q0 = qt * BM + wave * 32;

    const int b = slab / p.heads, h = slab % p.heads;
    const float* qg = (const float*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
    const float* kg = (const float*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
    const float* vg = (const float*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    float* og = (float*)p.o + b * p.o_batch_stride + h * p.o_head_stride;

    int kv_end = n;
    if (CAUSAL) kv_end = min(n, qt * BM + BM);
    const int nt = (kv_end + kKvBlkF32 - 1) / kKvBlkF32;

    issue_kv_tile_f32<D, NWAVES>(kg, vg, 0, n, p.kv_row_stride, smem, wave, lane);

    // Q fragments: lane (lq, hi) holds Q[q][8g + 4hi .. +3]; MFMA #(4g+e) uses element e
    f32x4 qf[G];
    {
        const int qrow = min(q0 + lq, n - 1);
        const float* qr = qg + (int64_t)qrow * p.q_row_stride + hi * 4;
#pragma unroll
        for (int g = 0; g < G; ++g) qf[g] = *(const f32x4*)(qr + g * 8) * p.scale_log2e;
    }
    // Q is pre-multiplied by scale*log2(e) (one fp32 rounding per element, once per workgroup), so the MFMA chain
    // delivers scores directly in the exp2 domain and p = exp2(s - m) is exact at the row maximum for any magnitude.

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m = -INFINITY, l = 0.0f;

    const int k_row_off = lq * C::kRowBytes;
    const int k_g = hi ^ k_swizzle_f32<D>(lq);
    const int v_lane_off = (4 * hi) * C::kRowBytes + lq * 4;

    for (int j = 0; j < nt; ++j) {
        wait_lds_dma();   // own LDS-DMA pieces of tile j have landed (hipcc does not insert this wait itself) ...
        __syncthreads();  // ... and so have everybody else's; all waves are done with the stage tile j+1 overwrites
        if (j + 1 < nt)
            issue_kv_tile_f32<D, NWAVES>(kg, vg, (j + 1) * kKvBlkF32, n, p.kv_row_stride,
                                         smem + ((j + 1) & 1) * C::kStageBytes, wave, lane);
        const int kv0 = j * kKvBlkF32;
        if (CAUSAL && kv0 > q0 + 31) continue;

        const char* ks_lds = smem + (j & 1) * C::kStageBytes;
        const char* vs_lds = ks_lds + C::kTileBytes;

        // ---- S^T = K Q^T : D/2 MFMAs of k = 2
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 kf = *(const f32x4*)(ks_lds + k_row_off + (((2 * g) ^ k_g) * 16));
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g][e], s, 0, 0, 0);
        }

        // ---- online softmax
        const bool need_mask = (kv0 + kKvBlkF32 > n) || (CAUSAL && (kv0 + kKvBlkF32 - 1 > q0));
        if (need_mask) {
            const int qi = q0 + lq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + 4 * hi + (r & 3) + 8 * (r >> 2);
                if ((key >= n) || (CAUSAL && key > qi)) s[r] = -INFINITY;
            }
        }
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m, mx);
        const float alpha = fast_exp2(m - m_new);
        m = m_new;
        float rs = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s[r] = fast_exp2(s[r] - m_new);
            rs += s[r];
        }
        l = fmaf(l, alpha, rs);
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;

        // ---- O^T += V^T P^T : MFMA #r contracts keys {r-th of hi=0, r-th of hi=1}
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int koff = ((r & 3) + 8 * (r >> 2)) * C::kRowBytes;
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const float vf = *(const float*)(vs_lds + v_lane_off + koff + db * 128);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], o[db], 0, 0, 0);
            }
        }
    }

    // ---- epilogue
    mfma_drain();  // the loop exit is a branch: the last P.V MFMAs may still be in flight
    const float lt = xhalf_sum(l);
    const float inv = 1.0f / lt;
    const int qi = q0 + lq;
    if (qi < n) {
        float* orow = og + (int64_t)qi * p.o_row_stride + 4 * hi;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = o[db][4 * g + e] * inv;
                *(f32x4*)(orow + db * 32 + 8 * g) = pk;
            }
        if (p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = (m + __builtin_amdgcn_logf(lt)) * kLn2;
    }
}

// MINWAVES_C: the occupancy hint of the causal instantiation (the mask code needs a few registers more: at 4 waves per SIMD, i.e.
// 128 registers, the D = 64 causal kernel spilled 56 bytes per lane)
template <int D, int NWAVES, int MINWAVES, int MINWAVES_C = MINWAVES>
static hipError_t launch_cfg_f32(const FwdParams& p0, int causal, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = NWAVES * 32;
    p.q_tiles = (p.n + BM - 1) / BM;
    const int64_t total = (int64_t)p.bh * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)total), block(NWAVES * kWave);
    // Three to six workgroups share a CU here.  Dealing a slab's tiles alternately from the heavy and the light end (causal_tile:
    // even rounds of an XCD's workgroups heavy, odd rounds light) evens out what the co-resident workgroups of a CU add up to.
    // Measured, causal, ms plain -> alternating: 16 x 4096 d = 64 0.538 -> 0.328, 8 x 8192 0.787 -> 0.617, 12 x 8192 1.321 -> 1.073,
    // 16 x 8192 d = 32 0.848 -> 0.630, 16 x 8192 d = 64 1.338 -> 1.308, 128 x 1024 0.248 -> 0.236, d = 128 2.45 -> 2.47.
    p.alt_order = causal ? 1 : 0;
    if (causal)
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, NWAVES, true, MINWAVES_C>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, NWAVES, false, MINWAVES>), grid, block, 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_fwd_f32(const FwdParams& p, int d, int causal, int variant, hipStream_t stream)
{
    (void)variant;
    switch (d) {
        case 32: return launch_cfg_f32<32, 4, 4>(p, causal, stream);
        case 64: return launch_cfg_f32<64, 4, 4, 3>(p, causal, stream);
        case 128: return launch_cfg_f32<128, 4, 2>(p, causal, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace fa
