// fa_launch.cpp -- one forward: its plan (fa_plan.cpp) executed on a stream -- the kernel families' launchers, key shares + combine,
// the report word of an fp32 FA_KERNEL_AUTO forward (fa_counters.cpp), and (ablation library) the conditional launch chains (fa_host.h).
// Replaces run_flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:590-602): no device sync, status codes.
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

namespace {
// key-split launch; p16: p0.v is the fp16 copy of V and p0 carries the chain's flag fields (the fp16-P kernel skips itself when the
// copy overflowed; the combine then merges garbage, which the chain's fallback launch overwrites -- it runs after this function)
hipError_t launch_bf16_keysplit(const fa::FwdParams& p0, int32_t d, int32_t causal, int32_t out_f32, int S, char* part, hipStream_t stream, int p16 = 0)
{
    const int n_kv = keysplit_rows(p0, S, causal);
    const int c = causal ? 1 : 0;
    const size_t o_bytes = (size_t)S * p0.bh * p0.n * d * 4u;
    float* o_part = (float*)part;
    float* lse_part = (float*)(part + o_bytes);
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
    hipError_t e;
    if (p16 == 0) e = fa::launch_bf16_x2(p, d, c, 1, 0, stream);
    else if (p16 == 3) e = fa::launch_bf16_pb2(p, d, c, 1, 1, stream);   // bf16 hi + lo terms of P, NB = 2
#if FA_ABLATION
    else if (p16 == 1) e = d == 32 ? fa::launch_bf16_x2_p16_d32(p, c, 1, stream) : d == 64 ? fa::launch_bf16_x2_p16_d64(p, c, 1, stream) : fa::launch_bf16_x2_p16_d128(p, c, 1, stream);
    else e = d == 32 ? fa::launch_bf16_x2_p16x2_d32(p, c, 1, stream) : d == 64 ? fa::launch_bf16_x2_p16x2_d64(p, c, 1, stream) : fa::launch_bf16_x2_p16x2_d128(p, c, 1, stream);
#else
    else e = hipErrorInvalidValue;
#endif
    fa::FwdParams pc = p0;
    pc.flag_mode = 0;
    if (e == hipSuccess) e = fa::launch_combine_splits(pc, o_part, lse_part, S, d, out_f32, stream);
    return e;
}

// fp32 tensors, key-split launch of the split kernel: p0 carries the chain's flag fields (flag_mode 3: every share bounds the logit
// width of its own keys)
// exact: the exact fp32 kernel (io_in: 0 = fp32 tensors, 2 = bf16 tensors -- the partials are fp32 either way; out_f32: what the combine stores)
hipError_t launch_f32_keysplit(const fa::FwdParams& p0, int32_t d, int32_t causal, int S, char* part, hipStream_t stream, bool exact = false, int io_in = 0,
                               int out_f32 = 1)
{
    const int n_kv = keysplit_rows(p0, S, causal);
    const size_t o_bytes = (size_t)S * p0.bh * p0.n * d * 4u;
    float* o_part = (float*)part;
    float* lse_part = (float*)(part + o_bytes);
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
    hipError_t e = exact ? fa::launch_fwd_f32(p, d, causal ? 1 : 0, 0, stream, io_in) : fa::launch_f32_split(p, d, causal ? 1 : 0, 0, stream);
    fa::FwdParams pc = p0;
    pc.flag_mode = 0;
    if (e == hipSuccess) e = fa::launch_combine_splits(pc, o_part, lse_part, S, d, out_f32, stream);
    return e;
}

#if FA_ABLATION
// bf16 tensors, fp16 P (ablation library): V -> fp16 copy in scratch, fp16-P kernel, split kernel as the conditional fallback
hipError_t launch_p16_chain(const fa::FwdParams& p0, int32_t d, int32_t causal, int32_t out_f32, const Plan& pl, char* ws, const ReportRef& f,
                            hipStream_t stream)
{
    void* v16 = ws + pl.v16_off;
    hipError_t e = fa::launch_cvt_v_f16(p0.v, v16, (int64_t)p0.bh * p0.n * d, f.word, f.serial, stream);
    if (e == hipSuccess) {
        fa::FwdParams p = p0;
        p.v = v16;
        p.flag = f.word;
        p.flag_serial = f.serial;
        p.flag_mode = 1;   // skip if the copy found a value fp16 cannot hold
        if (pl.S > 1) e = launch_bf16_keysplit(p, d, causal, out_f32, pl.S, ws + pl.part_off, stream, pl.terms);
        else if (pl.terms == 1) e = fa::launch_bf16_p16(p, d, causal ? 1 : 0, out_f32, stream);
        else e = fa::launch_bf16_p16x2(p, d, causal ? 1 : 0, out_f32, stream);
    }
    if (e == hipSuccess) {
        fa::FwdParams p = p0;
        p.flag = f.word;
        p.flag_serial = f.serial;
        p.flag_mode = 2;   // run only in that case
        e = fa::launch_bf16_split(p, d, causal ? 1 : 0, out_f32, 0, stream);
    }
    return e;
}

// fp32 tensors, long non-causal rows at head dim 64: K / V split once per launch into scratch (the same pass bounds the logit width),
// then the static-slot three-product kernel; guard, range or finiteness trouble raises the flag -> exact kernel.
// The pre-pass moves 2.5 x sizeof(K + V) + sizeof(Q) through HBM (~40 us at c3).
// The experimental three-product kernel of fa_f32_t3_kernel.h (ablation library only; FA_KERNEL_SPLIT tilings 8 = guarded chain with the exact
// kernel as fallback, 9 = the kernel alone, 16 + a = timing-only ablation a of the kernel alone)
hipError_t launch_f32_t3_chain(const fa::FwdParams& p0, int32_t d, char* scratch, const ReportRef& f, hipStream_t stream, bool guarded, int abl)
{
    const int64_t count = (int64_t)p0.bh * p0.n * d;
    hipError_t e = fa::launch_t3_prepass(p0.q, p0.k, p0.v, scratch, count, p0.scale_log2e, f.stats, f.serial, stream);
    if (e == hipSuccess) {
        fa::FwdParams p = p0;
        char* s = scratch;
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
    return e;
}
#endif
}  // namespace

// One forward.  ws == nullptr && !ws_mode: a convenience entry point -- scratch, if the plan wants any, comes from the private pool
// (never while the stream is capturing: the plan is then made without scratch).  ws_mode: the caller's workspace or nothing.
int launch(const fa::FwdParams& p_in, int32_t d, int32_t causal, int32_t dtype, int32_t kernel, hipStream_t stream, void* ws, size_t ws_bytes,
           bool ws_mode)
{
    const KernelSel sel = decode_kernel(kernel);
    t_last_chain = 0;
    g_stats.forwards.fetch_add(1, std::memory_order_relaxed);
    const bool capturing = stream_is_capturing(stream);
    fa::FwdParams p = p_in;
    p.cliffs = cliff_counters();            // (the current device's pair of counter words)
    Plan pl = make_plan(p, d, causal, dtype, kernel, ws_mode ? true : !capturing);
    if (pl.status != FA_OK) return pl.status;
    char* scratch = static_cast<char*>(ws);
    bool owned = false;
    // the routes that cannot run without their scratch (ablation library); every other plan has an unsplit form
    const bool needs_scratch = pl.route == kRouteP16Chain || pl.route == kRouteF32T3;
    if (pl.total > 0 && ws_mode) {
        if (scratch == nullptr || ws_bytes == 0) {
            // a binder that skips fa_workspace_bytes(): the forward runs without scratch (the unsplit launch) instead of failing
            scratch = nullptr;
            if (needs_scratch) return fail(FA_ERR_INVALID_ARGUMENT, "this kernel choice needs a workspace of fa_workspace_bytes() = %zu bytes", pl.total);
            pl = make_plan(p, d, causal, dtype, kernel, false);
            if (pl.status != FA_OK) return pl.status;
            g_stats.scratch_replans.fetch_add(1, std::memory_order_relaxed);
        } else {
            if (ws_bytes < pl.total) return fail(FA_ERR_INVALID_ARGUMENT, "workspace of %zu bytes is too small: this call needs fa_workspace_bytes() = %zu", ws_bytes, pl.total);
            if ((reinterpret_cast<uintptr_t>(scratch) & 255u) != 0) return fail(FA_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
        }
    } else if (pl.total > 0) {
        void* ptr = nullptr;
        const hipError_t ea = scratch_alloc(&ptr, pl.total, stream);
        if (ea != hipSuccess || ptr == nullptr) {
            (void)hipGetLastError();
            // the scratch paths are optimisations: the forward falls back to the launch without scratch
            if (needs_scratch) return fail(FA_ERR_HIP, "stream-ordered allocation of %zu scratch bytes failed: %s", pl.total, hipGetErrorString(ea));
            pl = make_plan(p, d, causal, dtype, kernel, false);
            if (pl.status != FA_OK) return pl.status;
            g_stats.scratch_replans.fetch_add(1, std::memory_order_relaxed);
        } else {
            scratch = static_cast<char*>(ptr);
            owned = true;
        }
    }
    const int out_f32 = dtype == FA_DTYPE_BF16_OUT_F32 ? 1 : 0;
    const int c = causal ? 1 : 0;
    hipError_t e = hipSuccess;
    switch (pl.route) {
        case kRouteNaive: e = fa::launch_naive(p, d, c, dtype, stream); break;
        case kRouteF32Exact: {   // (bf16 tensors: the wide head dims, widened on load; partials of a key-split launch are fp32 either way)
            const bool in_bf16 = dtype != FA_DTYPE_F32;
            if (pl.S > 1) e = launch_f32_keysplit(p, d, causal, pl.S, scratch + pl.part_off, stream, true, in_bf16 ? 2 : 0, dtype == FA_DTYPE_BF16 ? 0 : 1);
            else e = fa::launch_fwd_f32(p, d, c, sel.variant, stream, !in_bf16 ? 0 : out_f32 ? 2 : 1);
            break;
        }
        case kRouteF32Split: e = fa::launch_f32_split(p, d, c, sel.variant, stream); break;
        case kRouteF32Guarded: {   // split products behind the range guard: ONE launch -- a workgroup whose operands leave what fp16 terms hold
            ReportRef f;            // (or met a NaN) redoes its own rows in fp32 arithmetic inside the kernel (flag_mode 4).  The word only
            // REPORTS that (fa_last_forward_route); a captured forward takes none: its replays would all raise the same word
            const bool have = !capturing && next_report(f);
            fa::FwdParams pg = p;
            pg.flag = have ? f.word : nullptr;
            pg.flag_serial = have ? f.serial : 0u;
            pg.flag_mode = 4;
            if (pl.S > 1) e = launch_f32_keysplit(pg, d, causal, pl.S, scratch + pl.part_off, stream);   // (every share guards its own keys)
            else e = fa::launch_f32_split(pg, d, c, 0, stream);
            if (have && e == hipSuccess) {
                t_last_report = f;
                t_last_chain = 1;
            }
            break;
        }
#if FA_ABLATION
        case kRouteF32T3: {
            ReportRef f;
            if (!next_report(f)) return fail(FA_ERR_HIP, "no report word (hipGetSymbolAddress failed)");
            // a captured chain is replayed with the same serial: clear the word first, or the verdict of an earlier replay would stand
            if (capturing && hipMemsetAsync(f.word, 0, sizeof(uint32_t), stream) != hipSuccess) return fail(FA_ERR_HIP, "memset node of the chain's word failed");
            e = launch_f32_t3_chain(p, d, scratch + pl.part_off, f, stream, sel.variant == 8, sel.variant >= 16 ? sel.variant - 16 : 0);
            if (e == hipSuccess) {
                t_last_report = f;
                t_last_chain = 1;
            }
            break;
        }
        case kRouteP16Chain: {
            ReportRef f;
            if (!next_report(f) || (capturing && hipMemsetAsync(f.word, 0, sizeof(uint32_t), stream) != hipSuccess)) {
                e = fa::launch_bf16_split(p, d, c, out_f32, 0, stream);   // the chain's always-correct kernel alone
                break;
            }
            e = launch_p16_chain(p, d, causal, out_f32, pl, scratch, f, stream);
            if (e == hipSuccess) {
                t_last_report = f;
                t_last_chain = 2;
            }
            break;
        }
#endif
        case kRouteBf16Plain: e = fa::launch_fwd_bf16(p, d, c, out_f32, sel.variant, stream); break;
        case kRouteBf16Split: e = fa::launch_bf16_split(p, d, c, out_f32, sel.variant, stream); break;
        case kRouteBf16KeySplit: e = launch_bf16_keysplit(p, d, causal, out_f32, pl.S, scratch + pl.part_off, stream); break;
        case kRouteBf16Pb2:
            if (pl.S > 1) e = launch_bf16_keysplit(p, d, causal, out_f32, pl.S, scratch + pl.part_off, stream, 3);
            else e = fa::launch_bf16_pb2(p, d, c, out_f32, sel.variant, stream);
            break;
        default: return fail(FA_ERR_UNSUPPORTED, "kernel id %d is not in this build", sel.kind);
    }
    if (owned) {
        const hipError_t ef = hipFreeAsync(scratch, stream);
        if (e == hipSuccess) e = ef;
    }
    if (e == hipErrorInvalidValue && sel.variant != 0)
        return fail(FA_ERR_UNSUPPORTED, "tiling %d is not a shipped tiling of kernel family %d for head dim %d (timing-only ablations "
                                        "are built into libflashattn_amd_ablation.so only)", sel.variant, sel.kind, d);
    if (e != hipSuccess) return fail(FA_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(e));
    return FA_OK;
}

}  // namespace fa_host
