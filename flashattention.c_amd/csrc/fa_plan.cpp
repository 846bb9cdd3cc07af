// fa_plan.cpp -- argument validation, the parameter block, key-split arithmetic and the plan of one forward (fa_host.h).
// Host-side counterpart of forward() + run_flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:590-617): unlike the
// reference nothing here allocates, synchronises or asserts; errors are a return code + a thread-local message.
#include "fa_host.h"

#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace fa_host {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// head dims every kernel family is instantiated for / the exact fp32 kernel is (fa_fwd_f32_wide.hip: the other multiples of 32 up to 256,
// the head dims the reference can be compiled for by editing `#define d`, flashattention.cu:15) / the rung-0 kernel takes
bool head_dim_supported(int d) { return d == 32 || d == 64 || d == 128; }
bool head_dim_exact_f32(int d) { return d >= 32 && d <= 256 && d % 32 == 0; }
bool head_dim_naive(int d) { return d >= 1 && d <= 256; }


int validate_common(const void* q, const void* k, const void* v, const void* o, int64_t bh, int64_t n, int32_t d,
                    float scale, int32_t dtype)
{
    if (!q || !k || !v || !o) return fail(FA_ERR_INVALID_ARGUMENT, "null tensor pointer (q=%p k=%p v=%p o=%p)", q, k, v, o);
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(o))
        return fail(FA_ERR_INVALID_ARGUMENT, "tensor pointers must be 16-byte aligned");
    if (bh < 1 || n < 1) return fail(FA_ERR_INVALID_ARGUMENT, "bh (%lld) and n (%lld) must be >= 1", (long long)bh, (long long)n);
    if (bh > 0x7fffffffLL || n > (1LL << 24))
        return fail(FA_ERR_INVALID_ARGUMENT, "bh (%lld) or n (%lld) out of range", (long long)bh, (long long)n);
    if (d < 1) return fail(FA_ERR_INVALID_ARGUMENT, "head dim %d must be >= 1", d);
    if (!(scale > 0.0f) || !std::isfinite(scale)) return fail(FA_ERR_INVALID_ARGUMENT, "scale must be finite and > 0 (got %g)", (double)scale);
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_BF16_OUT_F32)
        return fail(FA_ERR_UNSUPPORTED, "unknown dtype %d", dtype);
    // o must not overlap an input: a tile whose optimistic pass fails its verification is recomputed from q, k, v AFTER the
    // first attempt was stored
    const uint64_t elems = (uint64_t)bh * (uint64_t)n * (uint64_t)d;
    const uint64_t in_bytes = elems * (dtype == FA_DTYPE_F32 ? 4u : 2u), out_bytes = elems * (dtype == FA_DTYPE_BF16 ? 2u : 4u);
    const uintptr_t ob = reinterpret_cast<uintptr_t>(o);
    for (const void* t : {q, k, v}) {
        const uintptr_t tb = reinterpret_cast<uintptr_t>(t);
        if (ob < tb + in_bytes && tb < ob + out_bytes) return fail(FA_ERR_INVALID_ARGUMENT, "o overlaps an input tensor (q, k or v)");
    }
    return FA_OK;
}

fa::FwdParams make_params(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int32_t d,
                          float scale)
{
    fa::FwdParams p{};
    memset(&p, 0, sizeof(p));
    p.q = q;
    p.k = k;
    p.v = v;
    p.o = o;
    p.lse = lse;
    p.q_batch_stride = p.kv_batch_stride = p.o_batch_stride = n * d;  // batch_stride of flashattention.cu:593
    p.q_row_stride = p.kv_row_stride = p.o_row_stride = d;
    p.n = (int32_t)n;
    p.bh = (int32_t)bh;
    p.scale = scale;
    p.scale_log2e = scale * fa::kLog2e;
    p.heads = 1;
    return p;
}

// FA_F32_AUTO=exact in the environment makes FA_KERNEL_AUTO compute fp32 tensors in fp32 arithmetic (FA_KERNEL_MFMA) process-wide:
// the switch for a deployment whose logits are too wide for 16-bit operands, without touching call sites.  Read once.
bool f32_auto_is_exact()
{
    static const bool exact = [] {
        const char* e = getenv("FA_F32_AUTO");
        return e != nullptr && strcmp(e, "exact") == 0;
    }();
    return exact;
}

// dense (bh, n, d) tensors: what make_params() builds.  The scratch paths (fp16 copy of V, key-split partials) index dense arrays
bool dense_layout(const fa::FwdParams& p, int32_t d)
{
    return p.heads == 1 && p.q_row_stride == d && p.kv_row_stride == d && p.o_row_stride == d && p.q_batch_stride == (int64_t)p.n * d &&
           p.kv_batch_stride == (int64_t)p.n * d && p.o_batch_stride == (int64_t)p.n * d;
}

// Key-split launch for grids that leave most of the chip idle (FlashDecoding-style): bf16 tensors, dense (bh, n, d) layout.  A lone
// 256-row tile over 8192 keys takes 0.108 ms whatever bh is (1, 2 or 4 slabs: the launch is one tile long), so S = 2 .. 8 workgroups per
// q-tile each take n / S keys (>= 1024) through the NB = 2 kernel -- the split index rides on the "head" index of FwdParams,
// kv_head_stride carries the key offset, the partial outputs (fp32, normalised) and their log-sum-exps go to scratch -- and
// fa_combine_splits_kernel merges them.  Non-causal: at most 128 tiles of 256 rows.  Causal (round 3): a launch lasts as long as its
// heaviest tile (all n keys) while the average tile has half of them, so up to 256 tiles are split; the shares are multiples of the
// tile height (a share then starts at or below a tile's first row, or lies entirely above the tile: an empty share that stores lse = -inf
// and costs a few microseconds of an otherwise idle CU).
int keysplit_rows(const fa::FwdParams& p, int S, int32_t causal)   // keys per share
{
    const int unit = causal ? 256 : 64;
    return ((p.n + S - 1) / S + unit - 1) / unit * unit;
}
int keysplit_factor(const fa::FwdParams& p, int32_t d, int32_t causal, bool f32, bool pb2)
{
    // Two-term P (FA_KERNEL_PB2; AUTO for an fp32 output), non-causal rows of 1024 .. 4095 keys on at most 64 tiles: its only tiling is the
    // 256-row workgroup, so such a launch leaves three quarters of the chip idle where the bf16-P dispatch has finer tilings to fall back
    // on.  Shares of >= 256 keys, up to 256 workgroups -- ms unsplit / key-split at d = 64, BH x N: 4 x 2048 0.045 / 0.024, 8 x 1024
    // 0.026 / 0.019, 8 x 2048 0.045 / 0.030, 16 x 1024 0.026 / 0.024, 16 x 2048 0.047 / 0.044, 1 x 2048 0.045 / 0.016; d = 32 8 x 1024 0.021 / 0.013;
    // d = 128 8 x 2048 0.070 / 0.049; from 128 tiles on the split loses (32 x 1024 0.028 / 0.033): profiles/r04_experiments.txt, fourth part.
    if (pb2 && !causal && dense_layout(p, d) && p.n >= 1024 && p.n < 4096) {
        const int64_t tiles = (int64_t)p.bh * ((p.n + 255) / 256);
        if (tiles > 64 || (d == 128 && p.n < 2048 && tiles > 32)) return 1;   // (d = 128, 16 x 1024: 0.040 / 0.042)
        int S = 1;
        while (S < 8 && tiles * (2 * S) <= 256 && p.n / (2 * S) >= 256) S *= 2;
        while (S > 1 && (int64_t)(S - 1) * keysplit_rows(p, S, causal) >= p.n) --S;
        return S;
    }
    if (!dense_layout(p, d) || p.n < 4096) return 1;
    if (((int64_t)(p.n - 1) * p.kv_row_stride + d) * 2 >= (int64_t)0xffffffffLL) return 1;   // the NB = 2 kernels' 32-bit slab offsets
#if FA_ABLATION
    {   // experiment switch (ablation library): FA_EXP_FORCE_S = S forces S key shares on causal launches of any grid size
        static const int force = [] { const char* e = getenv("FA_EXP_FORCE_S"); return e ? atoi(e) : 0; }();
        if (force > 1 && causal) return force;
    }
#endif
    const int64_t tiles = (int64_t)p.bh * ((p.n + 255) / 256);
    // bf16 tensors: causal launches of up to a full round of 256-row tiles are split (a causal launch lasts as long as its heaviest tile).
    // fp32 tensors (split kernel): its 128-row tiling, two workgroups per CU in the paired order, balances a causal round by itself --
    // ms unsplit / key-split at d = 64, BH x N: 16 x 4096 causal 0.126 / 0.151, 8 x 8192 causal 0.222 / 0.247; 8 x 4096 0.097 / 0.090,
    // 4 x 8192 0.197 / 0.191, 2 x 16384 0.369 / 0.342; 1 x 8192 0.179 / 0.059 -- and at d = 128 the split stops paying at 128 tiles
    // (8 x 4096 0.196 / 0.206, 4 x 8192 0.361 / 0.361; 4 x 4096 0.163 / 0.122): profiles/r03_short_rows.txt, third part.
    // bf16, causal, more than 128 tiles (ms unsplit / key-split): 8 x 8192 0.128 / 0.098 (d = 32 0.099 / 0.072, d = 128 0.191 / 0.170), 4 x 16384
    // 0.235 / 0.219; but 16 x 4096 0.071 / 0.077 (d = 128 0.108 / 0.131), 12 x 4096 0.070 / 0.074, d = 128 4 x 16384 0.358 / 0.384
    const bool long_causal = causal && p.n >= 8192 && (d < 128 || p.n < 16384);
    const int64_t cap = f32 ? (d == 128 ? 64 : 128) : (long_causal ? 256 : 128);
    if (tiles > cap) return 1;
    int S = 1;
    while (S < 8 && tiles * (2 * S) <= 2 * cap && p.n / (2 * S) >= 1024) S *= 2;
    while (S > 1 && (int64_t)(S - 1) * keysplit_rows(p, S, causal) >= p.n) --S;   // every split owns at least one key
    return S;
}

// Exact fp32 arithmetic (FA_KERNEL_MFMA; 128-row workgroups, one per CU already reads 0.78 of the fp32 MFMA peak -- BH x N = 4 x 8192
// 0.559 ms, 8 x 8192 1.055, 16 x 8192 2.061): a grid of fewer than 256 tiles leaves CUs idle, so its rows are cut into S <= 8 key shares
// of >= 1024 keys until the launch has 256 .. 512 workgroups (round 5: 1 x 8192 took 0.555 ms unsplit, as long as 4 x 8192).
int keysplit_factor_exact(const fa::FwdParams& p, int32_t d, int32_t causal)
{
    if (!dense_layout(p, d) || p.n < 2048) return 1;
    const int64_t tiles = (int64_t)p.bh * ((p.n + 127) / 128);
    // (a causal launch of one tile per CU lasts as long as its heaviest tile -- 4 x 8192 causal 0.552 ms, the non-causal launch's 0.559 --:
    // a full round of causal tiles is still split)
    if (tiles > (causal ? 256 : 255)) return 1;
    int S = 1;
    while (S < 8 && tiles * (2 * S) <= (causal ? 1024 : 512) && p.n / (2 * S) >= 1024) S *= 2;
    while (S > 1 && (int64_t)(S - 1) * keysplit_rows(p, S, causal) >= p.n) --S;   // every share owns at least one key
    return S;
}

// FA_KERNEL_AUTO, bf16 tensors, fp32 output (round 4): P as bf16 hi + bf16 lo in the one-wave-per-SIMD kernel (FA_KERNEL_PB2) -- one launch,
// V as it is, no scratch, at every launch size: ms at BH x N x d against round 3's chain (V -> fp16 copy, two fp16 terms of P, empty
// fallback launch), same box: 16 x 8192 x 64 0.352 / 0.367, 128 x 8192 x 64 2.78 / 2.79, causal 16 x 8192 x 64 0.205 / 0.227, 16 x 8192 x 128
// 0.589 / 0.631, 16 x 8192 x 32 0.260 / 0.272, 128 x 1024 x 64 0.060 / 0.072, 16 x 1024 x 64 0.028 / 0.037, 1 x 8192 x 64 (key-split) 0.043 / 0.051
// (profiles/r04_pb2_ab.txt), at 2.4e-5 against 3.1e-5 of the fp32 reference on c4.  Q.K^T is one bf16 product, exact in the fp32
// accumulator, so the error does not grow with the logit width (the split kernel's 16-bit Q' does: round 3's soak read 6.5e-4 from it
// at x3 logits); the split kernel remains the choice for slabs beyond 32-bit byte offsets.
#if FA_ABLATION
bool p16_available(const fa::FwdParams& p, int32_t d) { return dense_layout(p, d) && fa::bf16_p16_supported(p, d); }
#endif

// ---- the plan of one forward: which launches, how much scratch ---------------------------------------------------------------------
// One function decides for fa_workspace_bytes, fa_forward_ws and the convenience entries alike, so the size a caller is told is the
// size the launch uses.
Plan make_plan(const fa::FwdParams& p, int32_t d, int32_t causal, int32_t dtype, int32_t kernel, bool scratch_ok)
{
    Plan pl;
    const KernelSel sel = decode_kernel(kernel);
    if (sel.kind == FA_KERNEL_NAIVE) {
        if (!head_dim_naive(d)) pl.status = fail(FA_ERR_UNSUPPORTED, "the rung-0 kernel supports head dims 1 .. 256 (got %d)", d);
        pl.route = kRouteNaive;
        return pl;
    }
    if (sel.kind != FA_KERNEL_AUTO && sel.kind != FA_KERNEL_MFMA && sel.kind != FA_KERNEL_SPLIT && sel.kind != kKernelP16 && sel.kind != kKernelP16x2 &&
        sel.kind != FA_KERNEL_PB2) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "unknown kernel id %d", sel.kind);
        return pl;
    }
    if (!head_dim_supported(d)) {
        // Head dims outside {32, 64, 128} (round 6): the reference is generic over d % 32 == 0 by editing one macro (flashattention.cu:15,164).
        // d a multiple of 32 up to 256: the exact fp32 MFMA kernel (AUTO and MFMA alike) -- fp32 tensors as they are, bf16 tensors widened on
        // load (fp32 arithmetic on both); every other head dim up to 256: FA_KERNEL_AUTO runs the rung-0 kernel (fp32 arithmetic, correct, slow)
        // instead of refusing the call.
        if (head_dim_exact_f32(d) && sel.variant == 0 && (sel.kind == FA_KERNEL_AUTO || sel.kind == FA_KERNEL_MFMA)) {
            pl.route = kRouteF32Exact;
            const int S = scratch_ok ? keysplit_factor_exact(p, d, causal) : 1;
            if (S > 1) {
                pl.S = S;
                pl.part_off = kWsHeader;
                pl.part_bytes = (size_t)S * p.bh * p.n * d * 4u + (size_t)S * p.bh * p.n * 4u;
                pl.total = pl.part_off + align256(pl.part_bytes);
            }
            return pl;
        }
        if (sel.kind == FA_KERNEL_AUTO && sel.variant == 0 && head_dim_naive(d)) {
            pl.route = kRouteNaive;
            return pl;
        }
        pl.status = fail(FA_ERR_UNSUPPORTED, "head dim %d: this kernel family is instantiated for 32, 64, 128 (FA_KERNEL_AUTO takes any head dim up to 256; "
                                             "FA_KERNEL_MFMA multiples of 32 up to 256)", d);
        return pl;
    }
    if (dtype == FA_DTYPE_F32) {
        if (sel.kind == kKernelP16 || sel.kind == kKernelP16x2 || sel.kind == FA_KERNEL_PB2)
            pl.status = fail(FA_ERR_UNSUPPORTED, "FA_KERNEL_PB2 (and the ablation library's fp16-P kernels, ids 4 / 5) are bf16-tensor kernels");
        else if (sel.kind == FA_KERNEL_MFMA || (sel.kind == FA_KERNEL_AUTO && f32_auto_is_exact())) {
            pl.route = kRouteF32Exact;
            const int S = (scratch_ok && sel.variant == 0) ? keysplit_factor_exact(p, d, causal) : 1;
            if (S > 1) {   // idle grids: key shares + combine
                pl.S = S;
                pl.part_off = kWsHeader;
                pl.part_bytes = (size_t)S * p.bh * p.n * d * 4u + (size_t)S * p.bh * p.n * 4u;
                pl.total = pl.part_off + align256(pl.part_bytes);
            }
        }
        else if (sel.kind == FA_KERNEL_AUTO && sel.variant == 0) {
            pl.route = kRouteF32Guarded;
            // grids that leave the chip idle: the split kernel over key shares + combine (flag_mode 4: the workgroups of a share guard the
            // keys of THAT share and redo their own partial rows in fp32 arithmetic; the combine merges both kinds; the word only reports)
            const int S = scratch_ok ? keysplit_factor(p, d, causal, true) : 1;
            if (S > 1) {
                pl.S = S;
                pl.part_off = kWsHeader;
                pl.part_bytes = (size_t)S * p.bh * p.n * d * 4u + (size_t)S * p.bh * p.n * 4u;
                pl.total = pl.part_off + align256(pl.part_bytes);
            }
        }
#if FA_ABLATION
        else if (sel.kind == FA_KERNEL_SPLIT && sel.variant >= 8 && sel.variant < 32) {
            pl.route = kRouteF32T3;
            if (!fa::f32_t3_supported(p, d, causal))
                pl.status = fail(FA_ERR_UNSUPPORTED, "fa_fwd_f32_t3_kernel covers head dim 64, non-causal, N a multiple of 64, plain layout");
            pl.part_off = kWsHeader;
            pl.part_bytes = (size_t)p.bh * p.n * d * 8u;   // four bf16 arrays
            pl.total = pl.part_off + align256(pl.part_bytes);
        }
#endif
        else pl.route = kRouteF32Split;
        return pl;
    }
    // bf16 tensors.  AUTO: a caller who asks for the fp32 accumulator gets the accurate P (two bf16 terms: ~3e-5); a bf16 output rounds
    // at 2^-9 of |O| anyway and takes the fastest kernels (bf16 P).  MFMA / SPLIT / PB2 force one family.
    const int out_f32 = dtype == FA_DTYPE_BF16_OUT_F32 ? 1 : 0;
    const bool p16_kind = sel.kind == kKernelP16 || sel.kind == kKernelP16x2;
#if !FA_ABLATION
    if (p16_kind) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "kernel ids 4 / 5 (P and V in fp16, rounds 2-3) were replaced by FA_KERNEL_PB2 (P as two bf16 terms: faster, one launch, "
                                             "no scratch) and are built into libflashattn_amd_ablation.so only");
        return pl;
    }
#else
    if (p16_kind && !p16_available(p, d)) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "the fp16-P kernels need dense (bh, n, d) tensors and address a slab with 32-bit byte offsets (slabs below 4 GiB; got n = %d, d = %d)", p.n, d);
        return pl;
    }
    if (p16_kind && !scratch_ok) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "the fp16-P kernels need scratch, and stream-ordered allocations are not reliable inside a captured graph on this runtime: "
                                             "call fa_forward_ws with a workspace of fa_workspace_bytes() (legal under capture), or use FA_KERNEL_AUTO, which picks a "
                                             "kernel without scratch while the stream is capturing");
        return pl;
    }
#endif
    if (sel.kind == FA_KERNEL_PB2 && !fa::bf16_p16_supported(p, d)) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "FA_KERNEL_PB2 addresses a slab with 32-bit byte offsets (slabs below 4 GiB; got n = %d, d = %d)", p.n, d);
        return pl;
    }
    const bool pb2_route = sel.kind == FA_KERNEL_PB2 || (sel.kind == FA_KERNEL_AUTO && out_f32 && sel.variant == 0 && fa::bf16_p16_supported(p, d));
    const int S = (scratch_ok && sel.variant == 0) ? keysplit_factor(p, d, causal, false, pb2_route) : 1;
    // AUTO for an fp32 output (round 4): hi + lo bf16 terms of P in the one-wave-per-SIMD kernel -- one launch, V as it is, any layout
    // (slabs beyond 32-bit byte offsets: the split kernel below)
    if (sel.kind == FA_KERNEL_PB2 || (sel.kind == FA_KERNEL_AUTO && out_f32 && sel.variant == 0 && fa::bf16_p16_supported(p, d))) {
        pl.route = kRouteBf16Pb2;
        if (S > 1) {
            pl.S = S;
            pl.part_off = kWsHeader;
            pl.part_bytes = (size_t)S * p.bh * p.n * d * 4u + (size_t)S * p.bh * p.n * 4u;
            pl.total = pl.part_off + align256(pl.part_bytes);
        }
        return pl;
    }
    const size_t part = (size_t)S * p.bh * p.n * d * 4u + (size_t)S * p.bh * p.n * 4u;
    if (p16_kind) {
        pl.route = kRouteP16Chain;
        pl.terms = sel.kind == kKernelP16 ? 1 : 2;
        pl.S = S;
        pl.v16_off = kWsHeader;
        pl.v16_bytes = (size_t)p.bh * p.n * d * 2u;
        pl.total = pl.v16_off + align256(pl.v16_bytes);
        if (S > 1) {
            pl.part_off = pl.total;
            pl.part_bytes = part;
            pl.total += align256(part);
        }
    } else if (sel.kind == FA_KERNEL_SPLIT || (sel.kind == FA_KERNEL_AUTO && out_f32 && sel.variant == 0)) {
        pl.route = kRouteBf16Split;
    } else if (S > 1 && (sel.kind == FA_KERNEL_AUTO || sel.kind == FA_KERNEL_MFMA)) {
        pl.route = kRouteBf16KeySplit;
        pl.S = S;
        pl.part_off = kWsHeader;
        pl.part_bytes = part;
        pl.total = pl.part_off + align256(part);
    } else {
        pl.route = kRouteBf16Plain;
    }
    return pl;
}

}  // namespace fa_host
