// fa_f32_exact.h -- the exact fp32 attention of one 32-row block per wave (v_mfma_f32_32x32x2_f32 for both contractions: fp32 in,
// fp32 accumulate), as a device function: the body of fa_fwd_f32_kernel (fa_fwd_f32.hip, where the layout is described) and the
// in-kernel fallback of the guarded split kernel (fa_split_kernel.h: a tile whose logits are too wide for 16-bit operand terms is
// redone here, by the workgroup that found out, instead of by a second launch over the whole grid).
#pragma once
#include "fa_common.h"

namespace fa {

typedef __attribute__((address_space(3))) void lds_void_f;
typedef __attribute__((address_space(1))) const void gbl_cvoid_f;

constexpr int kKvBlkF32 = 32;

template <int D>
__device__ __forceinline__ int k_swizzle_f32(int row)
{
    constexpr int RB = 4 * D;
    constexpr int S = RB / 16;                       // 16-byte slots per row
    constexpr int R = (RB >= 256) ? 1 : 256 / RB;
    // the XOR must keep a slot inside its row: the mask is the largest 2^k - 1 (at most 15) with 2^k dividing S -- 7 at D = 32, 96, 160, 224
    // (8, 24, 40, 56 slots), 15 at D = 64, 128, 192, 256
    constexpr int M = (S % 16 == 0) ? 15 : (S % 8 == 0) ? 7 : (S % 4 == 0) ? 3 : 1;
    static_assert(S % (M + 1) == 0, "K swizzle leaves the row");
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

// One workgroup, NWAVES waves x 32 query rows (q0 = this wave's first row), keys [0, kv_end) of the slab or of a key share: kg / vg point
// at the share's first key, nk bounds the local key indices, kbeg is the share's first key in slab coordinates (causal: local key
// <= row - kbeg), kv_end the workgroup's causal bound in local coordinates (<= nk).  `smem`: 2 * F32Cfg::kStageBytes, 1024-byte aligned,
// not in use by anybody (callers barrier first).  Stores O (fp32, normalised) at p.o + o_slab + row * o_row_stride and, if asked, the
// log-sum-exp (natural log) at p.lse[slab * n + row].  Workgroup-uniform control flow (barriers inside).
// QREG = false (the in-kernel fallback, a cold path inside a kernel that is register-bound elsewhere): the Q fragments are re-read from
// global memory (L2) in every tile instead of living in D / 2 registers.
// bf16 tensors (round 6: head dims 96 ... 256 of bf16 tensors ran on the rung-0 kernel, 75x slower than the same call on fp32 tensors): TIN = __bf16.
// The arithmetic and the LDS images are the fp32 kernel's; the tiles come through registers (four bf16 values -> one 16-byte piece of the
// fp32 image: a shift or a mask each) instead of by LDS-DMA, Q is widened on load, FwdParams::o_is_bf16 rounds the normalised output once.
__device__ __forceinline__ f32x4 widen_bf16x4(const u32x2 v)
{
    f32x4 r;
    r[0] = __uint_as_float(v[0] << 16);
    r[1] = __uint_as_float(v[0] & 0xffff0000u);
    r[2] = __uint_as_float(v[1] << 16);
    r[3] = __uint_as_float(v[1] & 0xffff0000u);
    return r;
}
template <class TIN>
__device__ __forceinline__ f32x4 load_f32x4(const TIN* src)
{
    if constexpr (sizeof(TIN) == 4) return *(const f32x4*)src;
    else return widen_bf16x4(*(const u32x2*)src);
}

template <int D, int NWAVES, bool CAUSAL, bool QREG = true, class TIN = float>
__device__ __forceinline__ void f32_exact_rows(const FwdParams& p, char* smem, const TIN* qg, const TIN* kg, const TIN* vg, int64_t o_slab,
                                               int slab, int q0, int kbeg, int nk, int kv_end, int wave, int lane)
{
    using C = F32Cfg<D, NWAVES>;
    constexpr bool IN_BF16 = sizeof(TIN) == 2;
    constexpr int G = D / 8;    // ds_read_b128 groups per key row half (4 floats each)
    constexpr int DB = D / 32;  // 32-wide head-dim blocks of O^T
    const int lq = lane & 31, hi = lane >> 5;
    const int n = p.n;
    const int q0l = q0 - kbeg;  // this wave's first row in local key coordinates
    const int nt = (kv_end + kKvBlkF32 - 1) / kKvBlkF32;

    // bf16 tensors: this thread's 16-byte pieces of the two fp32 images of a tile, as the four bf16 values they are widened from.  Up to d = 160
    // they are requested at the top of the previous tile and held across it; above, the registers are the output accumulators' and Q's (472 -
    // 506 of 512 at d = 256): requested behind the previous tile's last product, their latency in the open.
    constexpr int NP = IN_BF16 ? C::kChunksPerWave : 1;
    constexpr bool HOLD = D <= 160;
    u32x2 kreg[NP], vreg[NP];
    auto request_tile = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int off = (wave + i * NWAVES) * 1024 + lane * 16;
            const int row = off / C::kRowBytes, phys = (off % C::kRowBytes) / 16;
            const int grow = min(kv0 + row, nk - 1);
            kreg[i] = *(const u32x2*)(kg + (int64_t)grow * p.kv_row_stride + (phys ^ k_swizzle_f32<D>(row)) * 4);
            vreg[i] = *(const u32x2*)(vg + (int64_t)grow * p.kv_row_stride + phys * 4);
        }
    };
    auto stage_tile = [&](char* stage) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int off = (wave + i * NWAVES) * 1024 + lane * 16;
            *(f32x4*)(stage + off) = widen_bf16x4(kreg[i]);
            *(f32x4*)(stage + C::kTileBytes + off) = widen_bf16x4(vreg[i]);
        }
    };
    if constexpr (IN_BF16) request_tile(0);
    else issue_kv_tile_f32<D, NWAVES>((const float*)kg, (const float*)vg, 0, nk, p.kv_row_stride, smem, wave, lane);

    // Q fragments: lane (lq, hi) holds Q[q][8g + 4hi .. +3]; MFMA #(4g+e) uses element e
    f32x4 qf[QREG ? G : 1];
    const TIN* qr = qg + (int64_t)min(q0 + lq, n - 1) * p.q_row_stride + hi * 4;
    if constexpr (QREG) {
#pragma unroll
        for (int g = 0; g < G; ++g) qf[g] = load_f32x4(qr + g * 8) * p.scale_log2e;
    }
    if constexpr (IN_BF16) stage_tile(smem);
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
        if constexpr (!IN_BF16) wait_lds_dma();   // own LDS-DMA pieces of tile j have landed (hipcc does not insert this wait itself) ...
        __syncthreads();  // ... and so have everybody else's; all waves are done with the stage tile j+1 overwrites
        if constexpr (IN_BF16) {
            if (HOLD && j + 1 < nt) request_tile((j + 1) * kKvBlkF32);
        } else {
            if (j + 1 < nt)
                issue_kv_tile_f32<D, NWAVES>((const float*)kg, (const float*)vg, (j + 1) * kKvBlkF32, nk, p.kv_row_stride,
                                             smem + ((j + 1) & 1) * C::kStageBytes, wave, lane);
        }
        const int kv0 = j * kKvBlkF32;
        if (!(CAUSAL && kv0 > q0l + 31)) {   // (else: tile above this wave's rows -- the other waves of the workgroup still need the next one staged)

        const char* ks_lds = smem + (j & 1) * C::kStageBytes;
        const char* vs_lds = ks_lds + C::kTileBytes;

        // ---- S^T = K Q^T : D/2 MFMAs of k = 2, one k-ordered chain on one accumulator (the reference's own summation order,
        // flashattention.cu:236-252).  (Round 6 tried four partial accumulators at head dims above 128, where one wave owns a SIMD: 4-19 %
        // SLOWER -- 64 more registers push the output accumulators' traffic through AGPRs; a chain on one accumulator costs nothing extra.)
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const f32x4 kf = *(const f32x4*)(ks_lds + k_row_off + (((2 * g) ^ k_g) * 16));
            f32x4 qv;
            if constexpr (QREG) qv = qf[g];
            else qv = *(const f32x4*)(qr + g * 8) * p.scale_log2e;
#pragma unroll
            for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qv[e], s, 0, 0, 0);
        }

        // ---- online softmax
        const bool need_mask = (kv0 + kKvBlkF32 > nk) || (CAUSAL && (kv0 + kKvBlkF32 - 1 > q0l));
        if (need_mask) {
            const int qil = q0l + lq;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kv0 + 4 * hi + (r & 3) + 8 * (r >> 2);
                if ((key >= nk) || (CAUSAL && key > qil)) s[r] = -INFINITY;
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
        // (the running maximum of a row settles within its first tiles: a wave whose 64 lanes all keep theirs has alpha == 1 exactly and skips the
        // D / 2 multiplications per lane -- bit-identical, and at head dims above 128, where the accumulators live in AGPRs, 3 instructions each)
        if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        }

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

        }   // tile not above the wave's rows
        if constexpr (IN_BF16) {
            if (j + 1 < nt) {
                if constexpr (!HOLD) request_tile((j + 1) * kKvBlkF32);
                stage_tile(smem + ((j + 1) & 1) * C::kStageBytes);
            }
        }
    }

    // ---- epilogue
    mfma_drain();  // the loop exit is a branch: the last P.V MFMAs may still be in flight
    const float lt = xhalf_sum(l);
    const float inv = 1.0f / lt;
    const int qi = q0 + lq;
    if (qi < n) {
        const int64_t o_off = o_slab + (int64_t)qi * p.o_row_stride + 4 * hi;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = o[db][4 * g + e] * inv;
                if (IN_BF16 && p.o_is_bf16) {   // (uniform; bf16 tensors only: one kernel per head dim serves both output types)
                    bf16x4 pb;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pb[e] = (__bf16)pk[e];
                    *(bf16x4*)((__bf16*)p.o + o_off + db * 32 + 8 * g) = pb;
                } else {
                    *(f32x4*)((float*)p.o + o_off + db * 32 + 8 * g) = pk;
                }
            }
        if (p.lse != nullptr && hi == 0) p.lse[(int64_t)slab * n + qi] = (m + __builtin_amdgcn_logf(lt)) * kLn2;
    }
}

}  // namespace fa
