// fa_fwd_f32_kernel.h -- fused flash-attention forward in exact fp32 (the reference's dtype) for gfx950: the kernel and its launcher,
// instantiated per head dim in fa_fwd_f32.hip (32, 64, 128), fa_fwd_f32_wide.hip (96, 160, 192, 224, 256) and, for bf16 tensors at those five,
// fa_fwd_f32_wide_bf16.hip.
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
#pragma once
#include "fa_common.h"
#include "fa_f32_exact.h"
#include "fa_kernels.h"

namespace fa {

// PAIRED (causal only): one workgroup computes TWO q-tiles, the heavy tile T - 1 - i and then the light tile i of its slab, so that every
// workgroup of the launch does the same T + 1 tile-steps of work whatever its position (round 5).  Unpaired, a causal launch is a bag
// of tiles of 1 .. T steps on three to four co-resident workgroups per CU: c3-causal read 0.67 of the fp32 MFMA peak where the
// non-causal launch reads 0.85, all of it load imbalance (the masked diagonal costs ~1 %).
// Key shares (FwdParams::n_kv > 0, round 5): as in the split kernel -- the "head" index of a slab is the share, kv_head_stride carries the
// key offset, the kernel works in the share's local key coordinates and leaves a normalised partial + its log-sum-exp for the combine.
// TIN = __bf16 (round 6, the wide head dims only): bf16 tensors widened on load, fp32 arithmetic; FwdParams::o_is_bf16: the output rounded once.
template <int D, int NWAVES, bool CAUSAL, int MINWAVES, bool PAIRED, class TIN = float>
__global__ __launch_bounds__(NWAVES* kWave, MINWAVES) void fa_fwd_f32_kernel(FwdParams p)
{
    using C = F32Cfg<D, NWAVES>;
    constexpr int BM = NWAVES * 32;
    static_assert(CAUSAL || !PAIRED, "pairing balances causal launches only");

    __shared__ __attribute__((aligned(1024))) char smem[2 * C::kStageBytes];

    if (flag_says_skip(p)) return;   // conditional launch of a chain (the ablation library's chains; FA_KERNEL_AUTO falls back inside the split kernel)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const int n = p.n;
    const int tiles = (n + BM - 1) / BM;
    // PAIRED over key shares (n_kv > 0): the items of one slab are its S x tiles (share, row tile) pairs, item h * tiles + t; the work of an
    // item and of its complement (S - 1 - h, tiles - 1 - t) = item N - 1 - idx adds up to n_kv + one tile for EVERY item (a share below a row
    // tile is full, above it empty, and the partial ones mirror each other), so a workgroup that computes both does the same work as every
    // other one -- 4 x 8192 causal over 4 shares: 0.526 -> 0.30 ms (profiles/r05_exact_share_pairs.txt).
    const bool shares = PAIRED && p.n_kv > 0;
    const int S = shares ? p.heads : 1;
    const int N = S * tiles;               // items of one unit (unit: a slab; over key shares: a slab with all its shares)
    int unit, item;
    {
        const int total = (shares ? p.bh / p.heads : p.bh) * p.q_tiles;   // PAIRED: q_tiles counts pairs of items
        const int w = xcd_remap(blockIdx.x, total);
        unit = w / p.q_tiles;
        item = w % p.q_tiles;
        if (CAUSAL && !PAIRED) item = causal_tile(p, item);
    }

#pragma unroll 1
    for (int half = 0; half < (PAIRED ? 2 : 1); ++half) {
        int idx = item;
        if constexpr (PAIRED) {
            idx = half == 0 ? N - 1 - item : item;
            if (half == 1) {
                if (item == N - 1 - item) break;   // odd item count: the middle item is its own pair
                __syncthreads();                   // every wave is out of the first item's last stage
            }
        }
        const int t = idx % tiles;
        const int slab = unit * S + idx / tiles;
        const int b = slab / p.heads, h = slab % p.heads;
        const TIN* qg = (const TIN*)p.q + b * p.q_batch_stride + h * p.q_head_stride;
        const TIN* kg = (const TIN*)p.k + b * p.kv_batch_stride + h * p.kv_head_stride;
        const TIN* vg = (const TIN*)p.v + b * p.kv_batch_stride + h * p.kv_head_stride;
        const int64_t o_slab = b * p.o_batch_stride + h * p.o_head_stride;
        const int kbeg = p.n_kv > 0 ? h * p.n_kv : 0;
        const int nk = p.n_kv > 0 ? min(p.n_kv, p.n_kv_total - kbeg) : n;
        const int q0 = t * BM + wave * 32;
        int kv_end = nk;
        if (CAUSAL) kv_end = min(nk, t * BM + BM - kbeg);
        if (CAUSAL && kv_end <= 0) {   // a key share entirely above this tile: weight 0 in the combine (workgroup-uniform)
            const int qi = q0 + (lane & 31);
            if (qi < n && p.lse != nullptr && lane < 32) p.lse[(int64_t)slab * n + qi] = -INFINITY;
            continue;
        }
        // (Q fragments in registers at every head dim: at 256 they take 128 beside 128 of output accumulators -- one wave owns its SIMD's 512 there.
        //  The first form of the wide instantiations re-read them from L2 in every tile: 32 exposed loads per tile, d = 256 at 0.56 of the fp32
        //  peak instead of 0.80: profiles/r06_exp5_exact_alpha_skip.txt)
        f32_exact_rows<D, NWAVES, CAUSAL, true, TIN>(p, smem, qg, kg, vg, o_slab, slab, q0, kbeg, nk, kv_end, wave, lane);
    }
}

// MINWAVES_C: the occupancy hint of the causal instantiation (the mask code needs a few registers more: at 4 waves per SIMD, i.e.
// 128 registers, the D = 64 causal kernel spilled 56 bytes per lane)
// order: 0 = the product choice, 1 = one tile per workgroup, 2 = paired tiles (causal only)
template <int D, int NWAVES, int MINWAVES, int MINWAVES_C = MINWAVES, class TIN = float>
static hipError_t launch_cfg_f32(const FwdParams& p0, int causal, int order, hipStream_t stream)
{
    FwdParams p = p0;
    constexpr int BM = NWAVES * 32;
    const int tiles = (p.n + BM - 1) / BM;
    // Pairing halves the number of workgroups and makes them equally long.  Sweep over 190 causal shapes (profiles/r05_exact_causal_sweep.txt,
    // ms paired / one tile per workgroup): from ~480 pairs on it wins or ties at every head dim (d = 64: 768 pairs 0.73-0.78, 1024 0.73-0.87,
    // 2048 0.87-0.91, 8192 0.97; c3-causal, 512 pairs: 1.00; d = 128 16 x 8192 0.87); one round of pairs on most of the CUs (144 .. 256) is
    // never more than 3 % behind and up to 1.5x ahead where the alternating order of single tiles lands badly (32 x 1500 0.67, 12 x 4096 0.71,
    // 40 x 1500 0.68); fewer pairs than that leave CUs idle (1.05-1.28), and in between (257 .. 479) single tiles are 2-6 % ahead.
    // Key shares (n_kv > 0): always paired -- an item with its complement over shares AND row tiles (see the kernel).
    const bool shares = p.n_kv > 0 && p.heads > 1;
    const int pairs = shares ? (p.heads * tiles + 1) / 2 : (tiles + 1) / 2;
    const int64_t units = shares ? p.bh / p.heads : p.bh;
    const int64_t npairs = units * pairs;
    const bool pair_auto = shares || npairs >= 480 || (npairs >= 144 && npairs <= 256 && tiles >= 8);
    const bool paired = causal && (order == 2 || (order == 0 && pair_auto)) && (p.n_kv == 0 || shares);
    p.q_tiles = paired ? pairs : tiles;
    const int64_t total = (paired ? units : (int64_t)p.bh) * p.q_tiles;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)total), block(NWAVES * kWave);
    // Unpaired causal launches: three to six workgroups share a CU.  Dealing a slab's tiles alternately from the heavy and the light end
    // (causal_tile: even rounds of an XCD's workgroups heavy, odd rounds light) evens out what the co-resident workgroups of a CU add up to.
    // Measured, causal, ms plain -> alternating: 16 x 4096 d = 64 0.538 -> 0.328, 8 x 8192 0.787 -> 0.617, 12 x 8192 1.321 -> 1.073,
    // 16 x 8192 d = 32 0.848 -> 0.630, 16 x 8192 d = 64 1.338 -> 1.308, 128 x 1024 0.248 -> 0.236, d = 128 2.45 -> 2.47.
    p.alt_order = (causal && !paired) ? 1 : 0;
    if (paired)
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, NWAVES, true, MINWAVES_C, true, TIN>), grid, block, 0, stream, p);
    else if (causal)
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, NWAVES, true, MINWAVES_C, false, TIN>), grid, block, 0, stream, p);
    else
        hipLaunchKernelGGL((fa_fwd_f32_kernel<D, NWAVES, false, MINWAVES, false, TIN>), grid, block, 0, stream, p);
    return hipGetLastError();
}

}  // namespace fa
