// fa_common.h -- shared host/device helpers for the gfx950 flash-attention forward kernels.
//
// Everything in csrc/ is written for CDNA4 (gfx950, wave64) only: MFMA builtins, LDS-DMA
// (global_load_lds), ds_read_b64_tr_b16 and the 64-lane cross-lane ops are used directly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

constexpr int kWave = 64;          // CDNA wavefront width
constexpr int kNumXcd = 8;         // MI355X: 8 XCDs, each with a private 4 MiB L2
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// Launch parameters shared by every kernel (all strides in elements).
struct FwdParams {
    const void* q;
    const void* k;
    const void* v;
    void* o;
    float* lse;            // nullable, (bh, n) fp32, natural log
    int64_t q_batch_stride; // elements between consecutive (batch*head) slabs of Q
    int64_t kv_batch_stride;
    int64_t o_batch_stride;
    int32_t q_row_stride;   // elements between consecutive sequence positions
    int32_t kv_row_stride;
    int32_t o_row_stride;
    int32_t n;              // sequence length
    int32_t bh;             // batch * heads
    int32_t q_tiles;        // ceil(n / rows-per-workgroup)
    float scale_log2e;      // scale * log2(e): softmax runs in the exp2 domain
    float scale;
    // packed-QKV (llm.c layout) addressing: slab index b*NH + h -> b * batch_stride + h * head_stride
    int32_t heads;          // NH (1 for the plain (BH, N, d) layout)
    int64_t q_head_stride;
    int64_t kv_head_stride;
    int64_t o_head_stride;
    int32_t o_is_bf16;      // split kernel only: O is bf16 (bf16 tensors); 0 = fp32
    // The report word of an fp32 FA_KERNEL_AUTO forward, and the ablation library's conditional launch chains (fa_counters.cpp, fa_launch.cpp): one 32-bit
    // device word per call; "set" means *flag == flag_serial
    // (serials are unique per call, so the word never needs clearing).
    //   flag_mode 0  ignore the word;   1  run only while the word is NOT set;   2  run only if the word IS set;
    //   flag_mode 3  (fp32 split kernel) always run, and set the word when the logits are too wide for 16-bit operands.
    //   flag_mode 4  (fp32 split kernel, FA_KERNEL_AUTO) always run; a workgroup whose logits are too wide redoes its own rows in fp32
    //                arithmetic on the spot (fa_f32_exact.h) and sets the word, if there is one (flag may be null), as a report.
    uint32_t* flag;
    uint32_t flag_serial;
    int32_t flag_mode;
    // fa_fwd_f32_t3_kernel only: the low bf16 terms of the pre-split K and V (p.k / p.v hold the high terms) and the pre-pass maxima
    const void* k_lo;
    const void* v_lo;
    int32_t n_kv, n_kv_total;   // key-split launches (non-causal NB = 2 kernels): a workgroup whose "head" index is s reads the keys
                                // [s * n_kv, min((s + 1) * n_kv, n_kv_total)) -- kv_head_stride carries the offset; 0 = all n keys
    int32_t alt_order;     // causal NB = 2 launches with two workgroups per CU: odd rounds of a CU's workgroups walk their slab light-to-heavy
    int32_t take_turns;    // non-causal NB = 2 launches whose whole grid is resident, two workgroups per CU: the pair shares the issue priority by the clock
    const unsigned long long* stats;   // [0] (serial << 32) | bits of max |k|,  [1] (serial << 32) | bits of max |q * scale * log2 e|_2^2
    // Nullable: two 64-bit counters in the device's memory (fa_counters.cpp: cliff_counters) that the kernels bump on their RARE slow paths --
    // [0] tiles redone with the rescaled / textbook mix behind a failed optimistic attempt, [1] workgroups of an fp32 FA_KERNEL_AUTO
    // forward redone in fp32 arithmetic -- so that the performance cliffs of DESIGN.md section 5 show up in fa_get_stats().
    unsigned long long* cliffs;
};

// one device-scope atomic per slow-path event, from one lane of the workgroup (see FwdParams::cliffs)
__device__ __forceinline__ void count_cliff(const FwdParams& p, int which)
{
    if (p.cliffs != nullptr && threadIdx.x == 0) __hip_atomic_fetch_add(p.cliffs + which, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Early exit of a conditionally launched kernel (wave-uniform scalar load; see FwdParams::flag_mode).
__device__ __forceinline__ bool flag_says_skip(const FwdParams& p)
{
    if (p.flag_mode != 1 && p.flag_mode != 2) return false;
    const uint32_t f = __hip_atomic_load(p.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (f == p.flag_serial) == (p.flag_mode == 1);
}

// Map the linear workgroup id onto (slab, q-tile) so that each XCD owns a contiguous range of work items
// and therefore whole (batch*head) slabs: the dispatcher places workgroup b on XCD b % 8 (observed, used for
// L2 locality only -- any placement is correct).  Bijective for every grid size.
__device__ __forceinline__ int xcd_remap(int bid, int total)
{
    const int q = total / kNumXcd, r = total % kNumXcd;
    const int xcd = bid % kNumXcd, idx = bid / kNumXcd;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Tile of a causal workgroup.  Default: walk each slab from its heaviest tile (light tiles then follow heavy ones onto a CU).
// alt_order (set by the launcher when the WHOLE grid is resident with two workgroups per CU: at most 64 per XCD): the dispatcher
// deals an XCD's workgroups over its 32 CUs in order, so positions pos and pos + 32 share a CU.  Within the stretch of a slab that
// an XCD owns (xcd_remap gives every XCD a contiguous range of items; a slab may be cut by the boundary), even rounds take that
// stretch's tiles from the heavy end, odd rounds from the light end: the two tiles of a CU add up to about the same work
// everywhere.  Each stretch maps onto its own set of tiles, so the whole is a bijection for every bh and tile count.
__device__ __forceinline__ int causal_tile(const FwdParams& p, int qt)
{
    if (!p.alt_order) return p.q_tiles - 1 - qt;
    const int total = p.bh * p.q_tiles, q = total / kNumXcd, r = total % kNumXcd;
    const int xcd = blockIdx.x % kNumXcd, pos = blockIdx.x / kNumXcd;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q, cnt = q + (xcd < r ? 1 : 0);
    const int w = base + pos, k = w % p.q_tiles, slab0 = w - k;
    const int seg0 = slab0 > base ? slab0 : base;                                               // first item of this stretch
    const int seg1 = slab0 + p.q_tiles < base + cnt ? slab0 + p.q_tiles : base + cnt;           // one past its last
    const int k_lo = seg0 - slab0, len = seg1 - seg0, pos0 = seg0 - base, j = w - seg0;
    auto even_before = [](int x) { return (x / 64) * 32 + (x % 64 < 32 ? x % 64 : 32); };
    const int e = even_before(pos) - even_before(pos0);                                         // even-round items of the stretch before this one
    const int hi = p.q_tiles - 1 - k_lo, lo = p.q_tiles - k_lo - len;                           // the stretch's tiles: lo .. hi
    return ((pos / 32) & 1) ? lo + (j - e) : hi - e;
}

// Drain this wave's outstanding LDS-DMA (global_load_lds) transfers.  LDS-DMA completion is tracked by vmcnt; a
// barrier does not wait for it, and the compiler only inserts the wait in front of LDS reads it thinks may alias.
__device__ __forceinline__ void wait_lds_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Let every matrix-core instruction in flight retire before VALU code touches its destination registers.  hipcc pads
// "MFMA write -> VALU read/write" hazards inside a basic block, but not across a branch or a loop exit (observed on
// ROCm 7.2: an epilogue or a rarely taken branch that starts right after the last MFMA read the accumulator one MFMA
// short).  64 idle states cover the longest (16-pass) instruction; use it only on cold paths.
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// min(2^x, 1): fmed3(., 0, 1) folds into the VOP3 clamp bit of v_exp_f32 -- one instruction
__device__ __forceinline__ float exp2_clamp01(float x) { return __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(x), 0.0f, 1.0f); }

// max(a, b, c) as ONE v_max3_f32.  Written as asm because hipcc puts a canonicalising v_max_f32 x, x, x in front of every fmaxf
// whose operand comes straight out of an MFMA accumulator (24 extra VALU per tile pair in the attention loop).
// ONLY for operands whose producing MFMA retired long ago (a whole pipeline phase earlier): hipcc pads the
// "MFMA write -> VALU read" hazard for its own instructions, never for the inside of an asm statement.  Reading the
// scores with this helper right after K.Q^T returned partial sums (observed: wrong row maxima -> skipped rescale).
// max3_safe() is the compiler-visible form for those places.
__device__ __forceinline__ float max3_raw(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float max3_safe(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// max / sum across the two half-waves (lane l <-> lane l^32) with one v_permlane32_swap (VALU, no LDS trip):
// swap(a, b) exchanges a[32..63] with b[0..31]; fed the same value twice it returns {lo, lo} and {hi, hi}.
__device__ __forceinline__ float xhalf_max(float x)
{
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x)
{
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

}  // namespace fa
