// fa_bf16_common.h -- building blocks shared by the bf16 flash-attention kernels (gfx950 only).
//
// LDS images (both produced by LDS-DMA with permuted per-lane SOURCE addresses, the destination being lane-linear):
//   K tile  row-major [key][D], 16-byte slots XOR-swizzled per row  -> conflict-free ds_read_b128 A fragments
//   V tile  [key/4][col/16][4][16] sub-tiles                        -> ds_read_b64_tr_b16 hands out V^T fragments
// Fragment conventions (v_mfma_f32_32x32x16_bf16, "swapped" product S^T = K Q^T): lane (q = lane&31, hi = lane>>5) owns
// scores of keys 4*hi + (r&3) + 8*(r>>2), r = 0..15, of one query row; the same key permutation is applied to the V^T
// fragment addresses, so P feeds the second MFMA straight from registers.
#pragma once
#include "fa_common.h"

namespace fa {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

constexpr int kKvBlk = 64;  // keys per K/V tile

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

// Enqueue the LDS-DMA of the K tile starting at key kv0 into `dst` (wave-uniform LDS address): row-major, slot-swizzled.
template <int D, int NWAVES>
__device__ __forceinline__ void issue_k_tile(const __bf16* __restrict__ kg, int kv0, int n, int row_stride, char* dst, int wave,
                                             int lane)
{
    using C = Bf16Cfg<D, NWAVES>;
#pragma unroll
    for (int i = 0; i < C::kChunksPerWave; ++i) {
        const int ch = wave + i * NWAVES;
        const int off = ch * 1024 + lane * 16;
        const int row = off / C::kRowBytes;
        const int phys = (off % C::kRowBytes) / 16;
        const int slot = phys ^ k_swizzle<D>(row);
        const int grow = min(kv0 + row, n - 1);
        const __bf16* src = kg + (int64_t)grow * row_stride + slot * 8;
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(dst + ch * 1024), 16, 0, 0);
    }
}

// Same for the V tile: [key/4][col/16][4][16] sub-tiles (128 bytes each).
template <int D, int NWAVES>
__device__ __forceinline__ void issue_v_tile(const __bf16* __restrict__ vg, int kv0, int n, int row_stride, char* dst, int wave,
                                             int lane)
{
    using C = Bf16Cfg<D, NWAVES>;
#pragma unroll
    for (int i = 0; i < C::kChunksPerWave; ++i) {
        const int ch = wave + i * NWAVES;
        const int blk = ch * 8 + lane / 8;
        const int kg4 = blk / (D / 16), cb = blk % (D / 16);
        const int key = kg4 * 4 + (lane % 8) / 2;
        const int col = cb * 16 + (lane & 1) * 8;
        const int grow = min(kv0 + key, n - 1);
        const __bf16* src = vg + (int64_t)grow * row_stride + col;
        __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(dst + ch * 1024), 16, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same two LDS images through BUFFER loads (buffer_load_dwordx4 ... lds): the per-lane source permutation is computed
// once per kernel (a 32-bit byte offset per lane), the tile / chunk position travels in the scalar offset, and rows past
// the end of the slab are answered with zeros by the bounds check of the descriptor -- so a stage costs one SALU add and
// one DMA instruction per 1 KiB piece instead of ~5 VALU of 64-bit address arithmetic per piece.
// Requires (n - 1) * row_stride * 2 + 2 * D < 2^32 (checked by the launcher).
// ---------------------------------------------------------------------------------------------------------------------
template <int D, int NWAVES>
struct TileDma {
    using C = Bf16Cfg<D, NWAVES>;
    __amdgpu_buffer_rsrc_t k_rsrc, v_rsrc;
    unsigned k_voff, v_voff;   // per-lane byte offset of this wave's chunk 0 within a tile (source permutation included)
    unsigned chunk_step;       // byte distance between the sources of two consecutive chunks of this wave
    unsigned stage_step;       // byte distance between two tiles (kKvBlk rows)

    __device__ __forceinline__ void init(const __bf16* kg, const __bf16* vg, int n, int row_stride, int wave, int lane)
    {
        const unsigned bytes = ((unsigned)(n - 1) * (unsigned)row_stride + D) * 2u;
        k_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)kg, 0, bytes, 0x00020000);
        v_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)vg, 0, bytes, 0x00020000);
        {   // K image: row-major, 16-byte slots XOR-swizzled (same mapping as issue_k_tile)
            const int off = wave * 1024 + lane * 16;
            const int row = off / C::kRowBytes;
            const int slot = ((off % C::kRowBytes) / 16) ^ k_swizzle<D>(row);
            k_voff = (unsigned)(row * row_stride + slot * 8) * 2u;
        }
        {   // V image: [key/4][col/16][4][16] sub-tiles (same mapping as issue_v_tile)
            const int blk = wave * 8 + lane / 8;
            const int kg4 = blk / (D / 16), cb = blk % (D / 16);
            const int key = kg4 * 4 + (lane % 8) / 2;
            const int col = cb * 16 + (lane & 1) * 8;
            v_voff = (unsigned)(key * row_stride + col) * 2u;
        }
        // chunk i of a wave is chunk (wave + i * NWAVES) of the tile: NWAVES * 512 / D rows further down in both images,
        // and both permutations are periodic in that distance (see the static_asserts)
        static_assert(((NWAVES * 512 / D) / ((D / 8 >= 16) ? 1 : 16 / (D / 8))) % (D / 8 >= 16 ? 16 : D / 8) == 0, "K swizzle period");
        static_assert((NWAVES * 8) % (D / 16) == 0, "V sub-tile period");
        chunk_step = (unsigned)(NWAVES * 512 / D) * (unsigned)row_stride * 2u;
        stage_step = (unsigned)kKvBlk * (unsigned)row_stride * 2u;
    }
    // enqueue the K (V) tile of stage j into the LDS tile at dst (wave-uniform)
    __device__ __forceinline__ void issue_k(unsigned stage_off, char* dst, int wave) const
    {
#pragma unroll
        for (int i = 0; i < C::kChunksPerWave; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (lds_void_t*)(dst + (wave + i * NWAVES) * 1024), 16, k_voff,
                                                     stage_off + i * chunk_step, 0, 0);
    }
    __device__ __forceinline__ void issue_v(unsigned stage_off, char* dst, int wave) const
    {
#pragma unroll
        for (int i = 0; i < C::kChunksPerWave; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (lds_void_t*)(dst + (wave + i * NWAVES) * 1024), 16, v_voff,
                                                     stage_off + i * chunk_step, 0, 0);
    }
};

template <int D, int NWAVES>
__device__ __forceinline__ void issue_kv_tile(const __bf16* __restrict__ kg, const __bf16* __restrict__ vg,
                                              int kv0, int n, int row_stride, char* stage, int wave, int lane)
{
    issue_k_tile<D, NWAVES>(kg, kv0, n, row_stride, stage, wave, lane);
    issue_v_tile<D, NWAVES>(vg, kv0, n, row_stride, stage + Bf16Cfg<D, NWAVES>::kTileBytes, wave, lane);
}

__device__ __forceinline__ bf16x8 pack_bf16x8(const f32x16& s, int base)
{
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (__bf16)s[base + i];
    return r;
}

// The format of P (and of the V image) in the second contraction, `PF` throughout the one-wave-per-SIMD kernels:
//   0 = bf16;  1 = fp16;  2 = fp16 hi + fp16 lo (V copied to fp16);  3 = bf16 hi + bf16 lo (V as it is; round 4)
constexpr bool pf_f16(int PF) { return PF == 1 || PF == 2; }
constexpr int pf_terms(int PF) { return PF >= 2 ? 2 : 1; }

// P fragment in the format of the second contraction: bf16 (default) or fp16 (round to nearest even: v_cvt_pk_f16_f32)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <int PF>
__device__ __forceinline__ bf16x8 pack_p16x8(const f32x16& s, int base)
{
    if constexpr (pf_f16(PF)) {
        f16x8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (_Float16)s[base + i];
        return __builtin_bit_cast(bf16x8, r);
    } else {
        return pack_bf16x8(s, base);
    }
}

template <int D>
__device__ __forceinline__ void qk_block(const char* k_lds, int k_row_off, int k_g, const bf16x8 (&qf)[D / 16], f32x16 (&s)[2])
{
    constexpr int RB = 2 * D;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
        const int slot_off = ((2 * ks) ^ k_g) * 16;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const bf16x8 kf = *(const bf16x8*)(k_lds + k_row_off + kb * 32 * RB + slot_off);
            s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[kb], 0, 0, 0);
        }
    }
}

template <int D>
__device__ __forceinline__ void pv_block(const char* v_lds, int v_lane_off, const bf16x8 (&pf)[4], f32x16 (&o)[D / 32])
{
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int db = 0; db < D / 32; ++db) {
                const int off0 = ((kb * 8 + 4 * t + 0) * (D / 16) + 2 * db) * 128;
                const int off1 = ((kb * 8 + 4 * t + 2) * (D / 16) + 2 * db) * 128;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_lds + v_lane_off + off0));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(v_lds + v_lane_off + off1));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kb * 2 + t], o[db], 0, 0, 0);
            }
}

template <bool CAUSAL>
__device__ __forceinline__ void mask_block(f32x16 (&s)[2], int kv0, int qi, int n, int hi)
{
    asm volatile("; mask_block" ::: "memory");  // not speculatable: keeps the caller's wave-uniform `if` a real branch
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = kv0 + kb * 32 + 4 * hi + (r & 3) + 8 * (r >> 2);
            if ((key >= n) || (CAUSAL && key > qi)) s[kb][r] = -INFINITY;
        }
}

// online softmax of one 32x64 score block held in registers; leaves P packed as the four B-operand fragments of P.V

// exp2-domain slack of the lazily updated running max: p = 2^(c*s - m - kLazyThr) spans 2^-kLazyThr .. 1 for the row maximum
// between rescales.  The shift is an exact power of two in P (bf16, 8 exponent bits), in O and in the row sum (fp32), so it
// cancels exactly in O / l; it only has to keep the entries that matter (within 2^-24 of the row maximum) above the
// smallest normal bf16, 2^-126.  64 makes the rescale branch (O-wide multiply + a matrix-pipe drain) a first-tile-only
// event on real data: with 8 it fired in a quarter of all steps of a 64-row wave on unit-variance data at scale 1.
constexpr float kLazyThr = 64.0f;

// Optimistic softmax (used by the software-pipelined kernels): p = 2^(c*s - m0 - kOptBias) with m0 the row maximum of the
// FIRST sub-tile, fixed for the whole row -- no running maximum, no rescale test, no branch in the main loop.  The shift is
// an exact power of two in P (bf16, 8 exponent bits), O and the row sum (fp32) and cancels in O / l, so the only requirement
// is range: the largest term of a row is >= 2^-100, and everything within 2^-24 of it stays above the smallest normal bf16
// (2^-126); the row sum l = sum p < 2^100 at the end of the tile proves that no term exceeded 2^100 (a larger term, +inf or
// NaN fails the test; fp32 accumulators hold N * 2^100 * |v| comfortably).  A row may thus outgrow the maximum of its first
// 32 keys by a factor 2^200 before its tile is redone with the lazily rescaled softmax -- which keeps every input correct.
constexpr float kOptBias = 100.0f;   // (the one-wave-per-SIMD kernels add up to 9 -- capped -- for bf16 P on short rows and subtract for the two-term P on long ones: xn_tile)
constexpr float kOptLimit = 0x1p100f;
constexpr float kOptTinyAcc = 0x1p-116f;   // an accumulator row below this had its dominant products near (or below) fp32's subnormals

// Zero accumulators behind an optimistic attempt: an all-zero V (padding heads, masked-out slabs) or products that underflowed as a whole
// (|v| below ~2^-50).  The rare path can afford to look: is every value of this workgroup's slab / key share exactly zero (+-0)?  Then the
// zeros it has stored ARE the result and the rescaled redo (the whole tile again) is not needed.  One pass over the share's V: L2 hits.
template <int D, int NWAVES>
__device__ __forceinline__ bool bf16_v_is_zero(const FwdParams& p)
{
    const int total = p.bh * p.q_tiles;
    const int slab = xcd_remap(blockIdx.x, total) / p.q_tiles;
    const int b = slab / p.heads, h = slab % p.heads;
    const __bf16* vg = (const __bf16*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
    const int nk = p.n_kv > 0 ? min(p.n_kv, p.n_kv_total - h * p.n_kv) : p.n;
    unsigned any = 0u;
#pragma unroll 4
    for (int i = threadIdx.x; i < nk * (D / 8); i += NWAVES * kWave) {
        const int row = i / (D / 8), c8 = (i % (D / 8)) * 8;
        const u32x4 x = *(const u32x4*)(vg + (int64_t)row * p.kv_row_stride + c8);
        any |= (x[0] | x[1] | x[2] | x[3]) & 0x7fff7fffu;
    }
    return __syncthreads_or(any != 0u ? 1 : 0) == 0;
}


typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int PF = 0>
__device__ __forceinline__ bf16x8 rowsum_a_operand(int lane)
{
    const bool one = (((lane & 15) >> 2) & 1) == ((lane >> 4) & 1);
    if constexpr (pf_f16(PF)) {
        f16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = one ? (_Float16)1.0f : (_Float16)0.0f;
        return __builtin_bit_cast(bf16x8, a);
    } else {
        bf16x8 a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = one ? (__bf16)1.0f : (__bf16)0.0f;
        return a;
    }
}

struct BlockState {   // running softmax state of one 32-row block (per lane: one query row, half of its keys)
    float m;          // exponent reference: p = 2^(c*s - m - kLazyThr)
    f32x4_t lacc;     // row sum of p (all four registers hold the same, complete, row sum)
};

// max phase result -> decision -> (rare) rescale.  Returns the exponent offset to use for this tile.
__device__ __forceinline__ void sum_block(const bf16x8& ones_a, const bf16x8 (&pf)[4], BlockState& st)
{
#pragma unroll
    for (int f = 0; f < 4; ++f) st.lacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones_a, pf[f], st.lacc, 0, 0, 0);
}


}  // namespace fa
