// fa_api.cpp -- the extern "C" boundary declared in include/flashattn_amd.h.
//
// Host-side counterpart of forward() + run_flash_tiled_coarse{,_causal}
// (/root/reference/src/flashattention.cu:590-617): argument validation, parameter block, kernel choice, launch.
// Unlike the reference it never allocates, never synchronises (except fa_time_forward) and reports errors by
// return code + thread-local message instead of assert().
#include "../../include/flashattn_amd.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "fa_kernels.h"

namespace {

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

bool head_dim_supported(int d) { return d == 32 || d == 64 || d == 128; }

// Decode the `kernel` argument: low byte = fa_kernel, bits 8.. = tiling variant (ablation driver only).
struct KernelSel {
    int kind;
    int variant;
};
KernelSel decode_kernel(int32_t kernel) { return KernelSel{kernel & 0xff, (kernel >> 8) & 0xff}; }

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

// ---- conditional launch chains ------------------------------------------------------------------------------------------------
// Two FA_KERNEL_AUTO paths are chains of launches on the caller's stream in which a later kernel runs or skips itself depending on
// what an earlier one found on the device (nothing is read back, nothing synchronises):
//   fp32 tensors   split kernel (bf16 pipe, 16-bit operand terms; raises the word when the logits are too wide for that)
//                  -> exact fp32 kernel, only if the word is raised;
//   bf16 tensors with fp32 output   V -> fp16 copy (raises the word when some |v| >= 2^16) -> fp16-P kernel unless raised
//                  -> split kernel (hi + lo bf16 terms of P) only if raised.
// The word lives in a per-device ring of 32-bit slots; "raised" means "equals this call's serial number", so a slot never needs
// clearing and concurrent calls (other streams, other threads) cannot see each other's verdicts.  A replayed hipGraph reuses its
// captured slot and serial: a verdict left by an earlier replay can only send a later one down the slower, more careful kernel.
constexpr int kFlagSlots = 4096;
__device__ uint32_t g_flag_ring[kFlagSlots];
__device__ unsigned long long g_stat_ring[kFlagSlots][2];   // pre-pass maxima of the t3 chain, tagged with the call's serial (fa_cvt.hip)
constexpr int kMaxDevices = 64;
std::atomic<uint32_t*> g_ring_base[kMaxDevices];
std::atomic<unsigned long long*> g_stat_base[kMaxDevices];
std::atomic<bool> g_pool_tuned[kMaxDevices];
std::atomic<uint32_t> g_serial{1};

struct FlagRef {
    uint32_t* word = nullptr;
    uint32_t serial = 0;
    unsigned long long* stats = nullptr;   // two 64-bit words of the same slot
};
thread_local FlagRef t_last_flag;   // chain state of this thread's most recent forward (fa_last_forward_route)
thread_local int t_last_chain = 0;  // 0 = no chain, 1 = fp32 guard, 2 = fp16-P

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}

bool next_flag(FlagRef& f)
{
    const int dev = current_device();
    if (dev < 0) return false;
    uint32_t* base = g_ring_base[dev].load(std::memory_order_acquire);
    if (base == nullptr) {
        void* sym = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_flag_ring)) != hipSuccess || sym == nullptr) return false;
        base = static_cast<uint32_t*>(sym);
        void* sym2 = nullptr;
        if (hipGetSymbolAddress(&sym2, HIP_SYMBOL(g_stat_ring)) != hipSuccess || sym2 == nullptr) return false;
        g_stat_base[dev].store(static_cast<unsigned long long*>(sym2), std::memory_order_release);
        g_ring_base[dev].store(base, std::memory_order_release);
    }
    uint32_t serial = g_serial.fetch_add(1, std::memory_order_relaxed);
    if (serial == 0) serial = g_serial.fetch_add(1, std::memory_order_relaxed);   // 0 is the ring's initial content
    f.word = base + serial % kFlagSlots;
    f.serial = serial;
    f.stats = g_stat_base[dev].load(std::memory_order_acquire) + 2 * (size_t)(serial % kFlagSlots);
    return true;
}

// Stream-ordered scratch (the fp16 copy of V): hipMallocAsync from the device's default pool, released behind the last kernel that
// reads it.  The pool keeps what it has handed out (release threshold raised once per device), so steady-state calls do not
// reach the driver.  Capturable: inside a stream capture the pair becomes graph memory nodes.
hipError_t scratch_alloc(void** ptr, size_t bytes, hipStream_t stream)
{
    const int dev = current_device();
    if (dev >= 0 && !g_pool_tuned[dev].exchange(true)) {
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, dev) == hipSuccess && pool != nullptr) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
    }
    return hipMallocAsync(ptr, bytes, stream);
}

// Stream-ordered scratch inside a captured graph is not reliable on this runtime (ROCm 7.2, MI355X): the first kernels that touch a
// graph allocation of more than a few MB after the graph starts did not see / keep their data (fp16-P chain at 16 x 8192: output
// unwritten on every replay of a one-launch graph; key-split launches: intermittently; tests/ + DESIGN.md section 1.2).  So the
// paths that need scratch are not taken while `stream` is capturing: FA_KERNEL_AUTO falls back to kernels without scratch (same or
// better accuracy, slower), an explicit FA_KERNEL_P16 is refused.
bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
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

// Key-split launch for grids that leave most of the chip idle (FlashDecoding-style): bf16 tensors, bf16 P, non-causal, plain
// (bh, n, d) layout, at most 128 tiles of 256 rows.  A lone 256-row tile over 8192 keys takes 0.108 ms whatever bh is (1, 2 or 4
// slabs: the launch is one tile long), so S = 2 .. 8 workgroups per q-tile each take n / S keys (>= 1024) through the NB = 2
// kernel -- the split index rides on the "head" index of FwdParams, kv_head_stride carries the key offset, the partial outputs
// (fp32, normalised) and their log-sum-exps go to stream-ordered scratch -- and fa_combine_splits_kernel merges them.
int keysplit_factor(const fa::FwdParams& p, int32_t d, int32_t causal)
{
    if (causal || p.heads != 1 || p.n < 4096) return 1;
    if (((int64_t)(p.n - 1) * p.kv_row_stride + d) * 2 >= (int64_t)0xffffffffLL) return 1;   // the NB = 2 kernels' 32-bit slab offsets
    const int64_t tiles = (int64_t)p.bh * ((p.n + 255) / 256);
    if (tiles > 128) return 1;
    int S = 1;
    while (S < 8 && tiles * (2 * S) <= 256 && p.n / (2 * S) >= 1024) S *= 2;
    return S;
}

// p16: p0.v is the fp16 copy of V and p0 carries the chain's flag fields (the fp16-P kernel skips itself when the copy overflowed; the
// combine then merges garbage, which the chain's fallback launch overwrites -- it runs after this function)
int launch_bf16_keysplit(const fa::FwdParams& p0, int32_t d, int32_t out_f32, int S, hipStream_t stream, bool p16 = false)
{
    int n_kv = ((p0.n + S - 1) / S + 63) / 64 * 64;
    while (S > 1 && (int64_t)(S - 1) * n_kv >= p0.n) --S;        // every split owns at least one key
    const size_t o_bytes = (size_t)S * p0.bh * p0.n * d * 4u, l_bytes = (size_t)S * p0.bh * p0.n * 4u;
    void* scratch = nullptr;
    hipError_t e = scratch_alloc(&scratch, o_bytes + l_bytes, stream);
    if (e != hipSuccess) return fail(FA_ERR_HIP, "hipMallocAsync(%zu bytes) for the key-split partials failed: %s", o_bytes + l_bytes, hipGetErrorString(e));
    float* o_part = (float*)scratch;
    float* lse_part = (float*)((char*)scratch + o_bytes);
    fa::FwdParams p = p0;
    p.bh = p0.bh * S;
    p.heads = S;
    p.q_head_stride = 0;
    p.kv_head_stride = (int64_t)n_kv * p0.kv_row_stride;
    p.o = o_part;
    p.o_batch_stride = (int64_t)p0.n * d;
    p.o_head_stride = (int64_t)p0.bh * p0.n * d;
    p.o_row_stride = d;
    p.lse = lse_part;
    p.n_kv = n_kv;
    p.n_kv_total = p0.n;
    if (!p16) e = fa::launch_bf16_x2(p, d, 0, 1, 0, stream);
    else e = d == 32 ? fa::launch_bf16_x2_p16_d32(p, 0, 1, stream) : d == 64 ? fa::launch_bf16_x2_p16_d64(p, 0, 1, stream) : fa::launch_bf16_x2_p16_d128(p, 0, 1, stream);
    fa::FwdParams pc = p0;
    pc.flag_mode = 0;
    if (e == hipSuccess) e = fa::launch_combine_splits(pc, o_part, lse_part, S, d, out_f32, stream);
    const hipError_t ef = hipFreeAsync(scratch, stream);
    if (e == hipSuccess) e = ef;
    if (e != hipSuccess) return fail(FA_ERR_HIP, "key-split launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

// bf16 tensors, fp16 P: V -> fp16 scratch copy, fp16-P kernel, split kernel as the conditional fallback
int launch_p16_chain(const fa::FwdParams& p0, int32_t d, int32_t causal, int32_t out_f32, hipStream_t stream)
{
    FlagRef f;
    if (!next_flag(f)) return fail(FA_ERR_HIP, "no device flag ring (hipGetSymbolAddress failed)");
    const size_t bytes = (size_t)p0.bh * (size_t)p0.n * (size_t)d * 2u;
    void* v16 = nullptr;
    hipError_t e = scratch_alloc(&v16, bytes, stream);
    if (e != hipSuccess) return fail(FA_ERR_HIP, "hipMallocAsync(%zu bytes) for the fp16 copy of V failed: %s", bytes, hipGetErrorString(e));
    e = fa::launch_cvt_v_f16(p0.v, v16, (int64_t)p0.bh * p0.n * d, f.word, f.serial, stream);
    if (e == hipSuccess) {
        fa::FwdParams p = p0;
        p.v = v16;
        p.flag = f.word;
        p.flag_serial = f.serial;
        p.flag_mode = 1;   // skip if the copy found a value fp16 cannot hold
        const int S = keysplit_factor(p0, d, causal);
        if (S > 1) {
            if (launch_bf16_keysplit(p, d, out_f32, S, stream, true) != FA_OK) e = hipErrorUnknown;   // fail() has recorded the message
        } else {
            e = fa::launch_bf16_p16(p, d, causal ? 1 : 0, out_f32, stream);
        }
    }
    if (e == hipSuccess) {
        fa::FwdParams p = p0;
        p.flag = f.word;
        p.flag_serial = f.serial;
        p.flag_mode = 2;   // run only in that case
        e = fa::launch_bf16_split(p, d, causal ? 1 : 0, out_f32, 0, stream);
    }
    const hipError_t ef = hipFreeAsync(v16, stream);
    if (e == hipSuccess) e = ef;
    if (e != hipSuccess) return fail(FA_ERR_HIP, "fp16-P launch chain failed: %s", hipGetErrorString(e));
    t_last_flag = f;
    t_last_chain = 2;
    return FA_OK;
}

#if FA_ABLATION
// fp32 tensors, long non-causal rows at head dim 64: K / V split once per launch into stream-ordered scratch (the same pass bounds the
// logit width), then the static-slot three-product kernel; guard, range or finiteness trouble raises the flag -> exact kernel.
// The pre-pass moves 2.5 x sizeof(K + V) + sizeof(Q) through HBM (~40 us at c3).
// The experimental three-product kernel of fa_f32_t3_kernel.h (ablation library only; FA_KERNEL_SPLIT tilings 8 = guarded chain with the exact
// kernel as fallback, 9 = the kernel alone, 16 + a = timing-only ablation a of the kernel alone)
int launch_f32_t3_chain(const fa::FwdParams& p0, int32_t d, hipStream_t stream, bool guarded, int abl)
{
    FlagRef f;
    if (!next_flag(f)) return fail(FA_ERR_HIP, "no device flag ring (hipGetSymbolAddress failed)");
    const int64_t count = (int64_t)p0.bh * p0.n * d;
    void* scratch = nullptr;
    hipError_t e = scratch_alloc(&scratch, (size_t)count * 8u, stream);   // four bf16 arrays
    if (e != hipSuccess) return fail(FA_ERR_HIP, "hipMallocAsync(%zu bytes) for the split K / V failed: %s", (size_t)count * 8u, hipGetErrorString(e));
    e = fa::launch_t3_prepass(p0.q, p0.k, p0.v, scratch, count, p0.scale_log2e, f.stats, f.serial, stream);
    if (e == hipSuccess) {
        fa::FwdParams p = p0;
        char* s = static_cast<char*>(scratch);
        p.k = s;
        p.k_lo = s + count * 2;
        p.v = s + count * 4;
        p.v_lo = s + count * 6;
        p.stats = f.stats;
        p.flag = f.word;
        p.flag_serial = f.serial;
        p.flag_mode = guarded ? 3 : 0;
        e = fa::launch_f32_t3(p, abl, stream);
    }
    if (e == hipSuccess && guarded) {
        fa::FwdParams p = p0;
        p.flag = f.word;
        p.flag_serial = f.serial;
        p.flag_mode = 2;
        e = fa::launch_fwd_f32(p, d, 0, 0, stream);
    }
    const hipError_t ef = hipFreeAsync(scratch, stream);
    if (e == hipSuccess) e = ef;
    if (e != hipSuccess) return fail(FA_ERR_HIP, "fp32 three-product launch chain failed: %s", hipGetErrorString(e));
    t_last_flag = f;
    t_last_chain = 1;
    return FA_OK;
}

#endif

// fp32 tensors, FA_KERNEL_AUTO: split kernel with the logit-width guard, exact kernel as the conditional fallback
int launch_f32_guarded(const fa::FwdParams& p0, int32_t d, int32_t causal, hipStream_t stream)
{
    FlagRef f;
    if (!next_flag(f)) return fail(FA_ERR_HIP, "no device flag ring (hipGetSymbolAddress failed)");
    fa::FwdParams p = p0;
    p.flag = f.word;
    p.flag_serial = f.serial;
    p.flag_mode = 3;
    hipError_t e = fa::launch_f32_split(p, d, causal ? 1 : 0, 0, stream);
    if (e == hipSuccess) {
        p.flag_mode = 2;
        e = fa::launch_fwd_f32(p, d, causal ? 1 : 0, 0, stream);
    }
    if (e != hipSuccess) return fail(FA_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    t_last_flag = f;
    t_last_chain = 1;
    return FA_OK;
}

bool p16_available(const fa::FwdParams& p, int32_t d) { return fa::bf16_p16_supported(p, d); }

// FA_KERNEL_AUTO, bf16 tensors, fp32 output: fp16 P or hi + lo bf16 terms?  The fp16 chain has a fixed cost the split kernel does
// not have -- the V copy (2 x sizeof(V) of HBM traffic) and two extra launches, ~15 us together -- and a faster kernel.  Measured
// on MI355X (ms, fp16 P / split): BH x N x d = 128 x 2048 x 64: 0.187 / 0.244, 32 x 4096 x 64: 0.169 / 0.234, 128 x 1024 x 128: 0.118 /
// 0.129, 128 x 1024 x 64: 0.074 / 0.077, the same causal: 0.075 / 0.064, 16 x 1024 x 64: 0.036 / 0.019, 128 x 1024 x 32: 0.046 / 0.056.
// Rule: fp16 P from 6e9 multiply-adds per contraction on (a causal launch counts half), from 2e9 at d = 32 (where the split kernel is
// slowest); the split kernel below that.
bool p16_worthwhile(const fa::FwdParams& p, int32_t d, int32_t causal)
{
    const double macs = (double)p.bh * (double)p.n * (double)p.n * (double)d * (causal ? 0.5 : 1.0);
    return macs >= (d == 32 ? 2e9 : 6e9);
}

int launch(const fa::FwdParams& p, int32_t d, int32_t causal, int32_t dtype, int32_t kernel, hipStream_t stream)
{
    const KernelSel sel = decode_kernel(kernel);
    hipError_t e = hipSuccess;
    t_last_chain = 0;
    if (sel.kind == FA_KERNEL_NAIVE) {
        if (dtype != FA_DTYPE_F32) return fail(FA_ERR_UNSUPPORTED, "the naive kernel is fp32 only");
        if (d > 256) return fail(FA_ERR_UNSUPPORTED, "naive kernel supports head dim <= 256 (got %d)", d);
        e = fa::launch_naive_f32(p, d, causal ? 1 : 0, stream);
    } else if (sel.kind == FA_KERNEL_AUTO || sel.kind == FA_KERNEL_MFMA || sel.kind == FA_KERNEL_SPLIT || sel.kind == FA_KERNEL_P16) {
        if (!head_dim_supported(d))
            return fail(FA_ERR_UNSUPPORTED, "head dim %d not instantiated for the MFMA kernels (32, 64, 128)", d);
        const int out_f32 = dtype == FA_DTYPE_BF16_OUT_F32 ? 1 : 0;
        if (dtype != FA_DTYPE_F32) {
            // bf16 tensors.  AUTO: a caller who asks for the fp32 accumulator gets the accurate P (fp16 where instantiated, hi + lo
            // bf16 terms elsewhere: within 1e-3 of the fp32 reference at scale 1); a bf16 output rounds at 2^-9 of |O| anyway
            // and takes the fastest kernels (bf16 P).  MFMA / SPLIT / P16 force one family.
            if (sel.kind == FA_KERNEL_P16 && !p16_available(p, d))
                return fail(FA_ERR_UNSUPPORTED, "the fp16-P kernels address a slab with 32-bit byte offsets (slabs below 4 GiB; got n = %d, d = %d)", p.n, d);
            const bool capturing = stream_is_capturing(stream);
            if (sel.kind == FA_KERNEL_P16 && capturing)
                return fail(FA_ERR_UNSUPPORTED, "FA_KERNEL_P16 needs stream-ordered scratch, which is not reliable inside a captured graph on this runtime; "
                                                "FA_KERNEL_AUTO picks a kernel without scratch while the stream is capturing");
            if (sel.kind == FA_KERNEL_P16 || (!capturing && sel.kind == FA_KERNEL_AUTO && out_f32 && sel.variant == 0 && p16_available(p, d) && (p16_worthwhile(p, d, causal) || keysplit_factor(p, d, causal) > 1)))
                return launch_p16_chain(p, d, causal, out_f32, stream);
            if (sel.kind == FA_KERNEL_SPLIT || (sel.kind == FA_KERNEL_AUTO && out_f32 && sel.variant == 0))
                e = fa::launch_bf16_split(p, d, causal ? 1 : 0, out_f32, sel.variant, stream);
            else {
                const int S = (!capturing && sel.variant == 0 && (sel.kind == FA_KERNEL_AUTO || sel.kind == FA_KERNEL_MFMA)) ? keysplit_factor(p, d, causal) : 1;
                if (S > 1) return launch_bf16_keysplit(p, d, out_f32, S, stream);
                e = fa::launch_fwd_bf16(p, d, causal ? 1 : 0, out_f32, sel.variant, stream);
            }
        } else if (sel.kind == FA_KERNEL_P16) {
            return fail(FA_ERR_UNSUPPORTED, "FA_KERNEL_P16 is a bf16-tensor kernel");
        } else if (sel.kind == FA_KERNEL_MFMA || (sel.kind == FA_KERNEL_AUTO && f32_auto_is_exact())) {
            e = fa::launch_fwd_f32(p, d, causal ? 1 : 0, sel.variant, stream);       // exact fp32 arithmetic
        } else if (sel.kind == FA_KERNEL_AUTO && sel.variant == 0) {
            return launch_f32_guarded(p, d, causal, stream);                         // split products behind the logit-width guard
#if FA_ABLATION
        } else if (sel.kind == FA_KERNEL_SPLIT && sel.variant >= 8 && sel.variant < 32) {   // the experimental three-product kernel
            if (!fa::f32_t3_supported(p, d, causal))
                return fail(FA_ERR_UNSUPPORTED, "fa_fwd_f32_t3_kernel covers head dim 64, non-causal, N a multiple of 64, plain layout");
            return launch_f32_t3_chain(p, d, stream, sel.variant == 8, sel.variant >= 16 ? sel.variant - 16 : 0);
#endif
        } else {
            e = fa::launch_f32_split(p, d, causal ? 1 : 0, sel.variant, stream);     // SPLIT: fp32 tensors on the bf16 pipe, unguarded
        }
    } else {
        return fail(FA_ERR_UNSUPPORTED, "unknown kernel id %d", sel.kind);
    }
    if (e == hipErrorInvalidValue && sel.variant != 0)
        return fail(FA_ERR_UNSUPPORTED, "tiling %d is not a shipped tiling of kernel family %d for head dim %d (timing-only ablations "
                                        "are built into libflashattn_amd_ablation.so only)", sel.variant, sel.kind, d);
    if (e != hipSuccess) return fail(FA_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

}  // namespace

extern "C" {

int fa_forward_ex(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int32_t d, float scale,
                  int32_t causal, int32_t dtype, int32_t kernel, void* stream)
{
    g_err[0] = 0;
    if (int rc = validate_common(q, k, v, o, bh, n, d, scale, dtype)) return rc;
    const fa::FwdParams p = make_params(q, k, v, o, lse, bh, n, d, scale);
    return launch(p, d, causal, dtype, kernel, static_cast<hipStream_t>(stream));
}

int fa_forward(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
               int32_t dtype, void* stream)
{
    return fa_forward_ex(q, k, v, o, nullptr, bh, n, d, scale, causal, dtype, FA_KERNEL_AUTO, stream);
}

int fa_forward_sharded(int32_t n_shards, const int32_t* device_ids, const void* const* q, const void* const* k, const void* const* v,
                       void* const* o, const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                       void* const* streams)
{
    g_err[0] = 0;
    if (n_shards < 1 || !device_ids || !q || !k || !v || !o || !bh)
        return fail(FA_ERR_INVALID_ARGUMENT, "fa_forward_sharded: bad shard table");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FA_ERR_NO_DEVICE, "no HIP device visible");
    for (int i = 0; i < n_shards; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev)
            return fail(FA_ERR_INVALID_ARGUMENT, "shard %d: device %d not in [0, %d)", i, device_ids[i], ndev);
        if (bh[i] < 0) return fail(FA_ERR_INVALID_ARGUMENT, "shard %d: negative bh", i);
        if (bh[i] == 0) continue;
        if (int rc = validate_common(q[i], k[i], v[i], o[i], bh[i], n, d, scale, dtype)) return rc;
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return fail(FA_ERR_HIP, "hipGetDevice failed");
    int rc = FA_OK;
    for (int i = 0; i < n_shards && rc == FA_OK; ++i) {
        if (bh[i] == 0) continue;  // more devices than slabs: this shard is empty
        hipError_t e = hipSetDevice(device_ids[i]);
        if (e != hipSuccess) {
            rc = fail(FA_ERR_HIP, "hipSetDevice(%d): %s", device_ids[i], hipGetErrorString(e));
            break;
        }
        const fa::FwdParams p = make_params(q[i], k[i], v[i], o[i], nullptr, bh[i], n, d, scale);
        rc = launch(p, d, causal, dtype, FA_KERNEL_AUTO, streams ? static_cast<hipStream_t>(streams[i]) : nullptr);
    }
    (void)hipSetDevice(prev);
    return rc;
}

int fa_forward_packed_qkv(const float* inp, float* out, int32_t B, int32_t T, int32_t C, int32_t NH, void* stream)
{
    g_err[0] = 0;
    if (!inp || !out) return fail(FA_ERR_INVALID_ARGUMENT, "null pointer");
    if (B < 1 || T < 1 || C < 1 || NH < 1 || C % NH != 0)
        return fail(FA_ERR_INVALID_ARGUMENT, "bad shape B=%d T=%d C=%d NH=%d", B, T, C, NH);
    const int hs = C / NH;
    if (!head_dim_supported(hs)) return fail(FA_ERR_UNSUPPORTED, "head size %d not instantiated (32, 64, 128)", hs);
    if (!aligned16(inp) || !aligned16(out)) return fail(FA_ERR_INVALID_ARGUMENT, "buffers must be 16-byte aligned");
    if ((int64_t)B * NH > 0x7fffffffLL) return fail(FA_ERR_INVALID_ARGUMENT, "B*NH too large");
    // (B, T, 3C): q at column h*hs, k at C + h*hs, v at 2C + h*hs of each token row
    // (attention_forward_cpu, /root/reference/src/llm.c/attention_forward.cu:66,74,115)
    fa::FwdParams p{};
    memset(&p, 0, sizeof(p));
    p.q = inp;
    p.k = inp + C;
    p.v = inp + 2 * (int64_t)C;
    p.o = out;
    p.lse = nullptr;
    p.q_batch_stride = p.kv_batch_stride = (int64_t)T * 3 * C;
    p.o_batch_stride = (int64_t)T * C;
    p.q_row_stride = p.kv_row_stride = 3 * C;
    p.o_row_stride = C;
    p.q_head_stride = p.kv_head_stride = p.o_head_stride = hs;
    p.heads = NH;
    p.n = T;
    p.bh = B * NH;
    p.scale = 1.0f / sqrtf((float)hs);  // attention_forward.cu:61,1123
    p.scale_log2e = p.scale * fa::kLog2e;
    return launch(p, hs, /*causal=*/1, FA_DTYPE_F32, FA_KERNEL_AUTO, static_cast<hipStream_t>(stream));
}

static int time_forward_impl(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale,
                             int32_t causal, int32_t dtype, int32_t kernel, void* stream, int32_t warmup, int32_t iters,
                             float* ms_per_forward, bool graph_replay)
{
    g_err[0] = 0;
    if (!ms_per_forward || iters < 1 || warmup < 0) return fail(FA_ERR_INVALID_ARGUMENT, "bad timing arguments");
    if (int rc = validate_common(q, k, v, o, bh, n, d, scale, dtype)) return rc;
    const fa::FwdParams p = make_params(q, k, v, o, nullptr, bh, n, d, scale);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        return fail(FA_ERR_HIP, "hipEventCreate failed");
    }
    int rc = FA_OK;
    for (int i = 0; i < warmup && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, s);
    if (rc == FA_OK && graph_replay) {  // the `iters` launches captured into one hipGraph, one replay timed
        hipStream_t cs = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamCreate(&cs) != hipSuccess) rc = fail(FA_ERR_HIP, "hipStreamCreate failed");
        if (rc == FA_OK && hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal) != hipSuccess) rc = fail(FA_ERR_HIP, "begin capture failed");
        for (int i = 0; i < iters && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, cs);
        if (rc == FA_OK && hipStreamEndCapture(cs, &graph) != hipSuccess) rc = fail(FA_ERR_HIP, "end capture failed");
        if (rc == FA_OK && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(FA_ERR_HIP, "graph instantiate failed");
        if (rc == FA_OK) {
            (void)hipGraphLaunch(exec, cs);   // warm replay
            (void)hipStreamSynchronize(cs);
            (void)hipEventRecord(e0, cs);
            (void)hipGraphLaunch(exec, cs);
            (void)hipEventRecord(e1, cs);
            const hipError_t e = hipEventSynchronize(e1);
            if (e != hipSuccess) rc = fail(FA_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
            if (rc == FA_OK) {
                float ms = 0.0f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                *ms_per_forward = ms / (float)iters;
            }
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (cs) (void)hipStreamDestroy(cs);
    } else if (rc == FA_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, s);
        (void)hipEventRecord(e1, s);
        const hipError_t e = hipEventSynchronize(e1);
        if (rc == FA_OK && e != hipSuccess) rc = fail(FA_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
        if (rc == FA_OK) {
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            *ms_per_forward = ms / (float)iters;
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int fa_time_forward(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale,
                    int32_t causal, int32_t dtype, int32_t kernel, void* stream, int32_t warmup, int32_t iters, float* ms_per_forward)
{
    const char* env = getenv("FA_TIME_GRAPH");  // experiment switch of the C driver
    return time_forward_impl(q, k, v, o, bh, n, d, scale, causal, dtype, kernel, stream, warmup, iters, ms_per_forward,
                             env && env[0] == '1');
}

int fa_time_forward_graph(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale,
                          int32_t causal, int32_t dtype, int32_t kernel, int32_t warmup, int32_t iters, float* ms_per_forward)
{
    return time_forward_impl(q, k, v, o, bh, n, d, scale, causal, dtype, kernel, nullptr, warmup, iters, ms_per_forward, true);
}

int fa_last_forward_route(void* stream, int32_t* route)
{
    g_err[0] = 0;
    if (!route) return fail(FA_ERR_INVALID_ARGUMENT, "null route pointer");
    *route = 0;
    if (t_last_chain == 0) return FA_OK;
    uint32_t word = 0;
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipMemcpy(&word, t_last_flag.word, sizeof(word), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(FA_ERR_HIP, "reading the chain's flag word failed: %s", hipGetErrorString(e));
    *route = word == t_last_flag.serial ? 2 : 1;
    return FA_OK;
}

const char* fa_last_error(void) { return g_err; }

int fa_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* fa_version(void)
{
#if FA_ABLATION
    return "flashattn_amd abi 2 gfx950 (hip, mfma f32 32x32x2 / bf16, f16 32x32x16, lds-dma) +ablation";
#else
    return "flashattn_amd abi 2 gfx950 (hip, mfma f32 32x32x2 / bf16, f16 32x32x16, lds-dma)";
#endif
}

const char* fa_kernel_name_for(int32_t dtype, int32_t d, int32_t causal, int64_t bh, int64_t n)
{
    if (!head_dim_supported(d) || bh < 1 || n < 1) return nullptr;
    if (dtype == FA_DTYPE_F32) {
        if (f32_auto_is_exact()) return "fa_fwd_f32_kernel";
        return "fa_fwd_f32_split_kernel";
    }
    if (dtype == FA_DTYPE_BF16) {
        fa::FwdParams pk{};
        memset(&pk, 0, sizeof(pk));
        pk.bh = (int32_t)bh, pk.n = (int32_t)n, pk.heads = 1, pk.kv_row_stride = d;
        if (keysplit_factor(pk, d, causal) > 1) return "fa_fwd_bf16_x2_kernel";       // small grids: key-split launch of the NB = 2 kernel
        return fa::bf16_kernel_name(bh, n, d, causal);
    }
    if (dtype == FA_DTYPE_BF16_OUT_F32) {   // the accurate P (see fa_dtype): fp16 (slabs below 4 GiB, launches large enough), hi + lo bf16 terms otherwise
        if (((n - 1) * d + d) * 2 >= 0xffffffffLL) return "fa_fwd_f32_split_kernel";
        fa::FwdParams pk{};
        memset(&pk, 0, sizeof(pk));
        pk.bh = (int32_t)bh, pk.n = (int32_t)n, pk.heads = 1, pk.kv_row_stride = d;
        if (keysplit_factor(pk, d, causal) > 1) return "fa_fwd_bf16_x2_p16_kernel";   // small grids: key-split launch of the NB = 2 kernel
        if ((double)bh * (double)n * (double)n * (double)d * (causal ? 0.5 : 1.0) < (d == 32 ? 2e9 : 6e9)) return "fa_fwd_f32_split_kernel";
        return (d == 64 && fa::bf16_p16_uses_x4(bh, n, causal)) ? "fa_fwd_bf16_x4_p16_kernel" : "fa_fwd_bf16_x2_p16_kernel";
    }
    return nullptr;
}

const char* fa_kernel_name(int32_t dtype, int32_t d, int32_t causal) { return fa_kernel_name_for(dtype, d, causal, 16, 8192); }

}  // extern "C"
