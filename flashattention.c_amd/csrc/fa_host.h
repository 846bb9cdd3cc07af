// fa_host.h -- declarations shared by the host-side translation units behind include/flashattn_amd.h (round 5: fa_api.cpp was one
// 1400-line file):
//   fa_plan.cpp      argument validation, the parameter block, key-split arithmetic, make_plan (which launches, how much scratch)
//   fa_counters.cpp  the report ring of fp32 FA_KERNEL_AUTO forwards, the kernels' slow-path counters, private scratch pools, host counters
//   fa_launch.cpp    launch(): one forward = its plan executed on a stream (key shares + combine, the ablation library's chains)
//   fa_shard.cpp     fa_forward_sharded: one persistent host thread per shard
//   fa_timing.cpp    fa_time_forward{,_graph}: event-timed loops
//   fa_selftest.cpp  (sanitizer build only) device-free self-test of the plans and the no-device validation paths
//   fa_api.cpp       the extern "C" entry points, nothing else
// Host-side counterpart of forward() + run_flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:590-617).
#pragma once
#include "../../include/flashattn_amd.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <vector>

#include "fa_kernels.h"

namespace fa_host {

// ---- errors, validation, parameter block (fa_plan.cpp)
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);
bool aligned16(const void* p);
bool head_dim_supported(int d);    // 32, 64, 128: every kernel family
bool head_dim_exact_f32(int d);    // multiples of 32 up to 256: the exact fp32 MFMA kernel
bool head_dim_naive(int d);        // 1 .. 256: the rung-0 kernel
// kernel ids 4 / 5 (P and V in fp16, rounds 2-3): not part of the product ABI since ABI 6; the ablation library still answers them
constexpr int kKernelP16 = 4, kKernelP16x2 = 5;
// Decode the `kernel` argument: low byte = fa_kernel, bits 8.. = tiling variant (ablation driver only).
struct KernelSel {
    int kind;
    int variant;
};
inline KernelSel decode_kernel(int32_t kernel) { return KernelSel{kernel & 0xff, (kernel >> 8) & 0xff}; }
int validate_common(const void* q, const void* k, const void* v, const void* o, int64_t bh, int64_t n, int32_t d, float scale, int32_t dtype);
fa::FwdParams make_params(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int32_t d, float scale);

// ---- the plan of one forward (fa_plan.cpp)
bool f32_auto_is_exact();
bool dense_layout(const fa::FwdParams& p, int32_t d);
int keysplit_rows(const fa::FwdParams& p, int S, int32_t causal);
int keysplit_factor(const fa::FwdParams& p, int32_t d, int32_t causal, bool f32 = false, bool pb2 = false);
int keysplit_factor_exact(const fa::FwdParams& p, int32_t d, int32_t causal);
enum Route {
    kRouteNaive, kRouteF32Exact, kRouteF32Guarded, kRouteF32Split, kRouteF32T3,
    kRouteBf16Plain,      // one launch of the bf16-P dispatch (launch_fwd_bf16)
    kRouteBf16KeySplit,   // bf16-P NB = 2 kernel over key shares + combine
    kRouteBf16Split,      // hi + lo bf16 terms of P and Q' (no scratch)
    kRouteBf16Pb2,        // hi + lo bf16 terms of P in the one-wave-per-SIMD kernel (one launch, no scratch; key-split for idle grids)
    kRouteP16Chain        // (ablation library) V -> fp16 copy, fp16-P kernel (key-split for idle grids), split kernel as the conditional fallback
};
constexpr size_t kWsHeader = 0;     // (until ABI 5 the first 256 bytes of a workspace held the forward's report word)
struct Plan {
    int status = FA_OK;   // FA_OK, or the error fail() recorded
    Route route = kRouteNaive;
    int S = 1;            // key-split factor
    int terms = 1;        // fp16-P chain: fp16 terms of P (kernel id 4: one, id 5: two)
    size_t v16_off = 0, v16_bytes = 0, part_off = 0, part_bytes = 0, total = 0;   // workspace layout (total = 0: no scratch)
};
inline size_t align256(size_t x) { return (x + 255u) & ~(size_t)255u; }
// scratch_ok: scratch is available to this launch (a workspace was passed, or the stream is not capturing)
Plan make_plan(const fa::FwdParams& p, int32_t d, int32_t causal, int32_t dtype, int32_t kernel, bool scratch_ok);

// ---- report words, counters, scratch pools (fa_counters.cpp)
// process-wide host counters behind fa_get_stats()
struct Stats {
    std::atomic<uint64_t> forwards{0}, scratch_replans{0};
};
extern Stats g_stats;

// The report word of an fp32 FA_KERNEL_AUTO forward (and, ablation library, the flag of a conditional launch chain): word `serial % kReportRing`
// of a ring in the device's memory, "raised" = the word equals the call's serial.  No table, no lock, nothing to release: a word is reused
// after kReportRing further calls (a report older than that reads as "not raised"; the ablation chains, which DECIDE launches by the word,
// assume fewer than kReportRing of them in flight per device).
constexpr int kReportRing = 1024;
constexpr int kMaxDevices = 64;
struct ReportRef {
    uint32_t* word = nullptr;
    uint32_t serial = 0;
    unsigned long long* stats = nullptr;   // two 64-bit words beside it (pre-pass maxima of the ablation library's t3 chain)
};
extern thread_local ReportRef t_last_report;   // of this thread's most recent reporting forward (fa_last_forward_route)
extern thread_local int t_last_chain;          // 0 = nothing to report, 1 = fp32 guard, 2 = fp16-P chain (ablation library)

int current_device();
bool stream_is_capturing(hipStream_t stream);
bool next_report(ReportRef& r);                 // false: no device symbol (the forward then runs without a word)
// The two slow-path counters of FwdParams::cliffs (a pair of device words per GPU, bumped by the kernels with device-scope atomics);
// cliff_count() is a BLOCKING 8-byte copy per device in use (fa_read_device_counters)
unsigned long long* cliff_counters();
unsigned long long cliff_count(int which);
hipError_t scratch_alloc(void** ptr, size_t bytes, hipStream_t stream);   // from the device's PRIVATE stream-ordered pool

// ---- one forward (fa_launch.cpp).  ws == nullptr && !ws_mode: a convenience entry point -- scratch, if the plan wants any, comes from the
// private pool (never while the stream is capturing: the plan is then made without scratch).  ws_mode: the caller's workspace or nothing.
int launch(const fa::FwdParams& p, int32_t d, int32_t causal, int32_t dtype, int32_t kernel, hipStream_t stream, void* ws = nullptr,
           size_t ws_bytes = 0, bool ws_mode = false);

// ---- fa_forward_sharded (fa_shard.cpp): run work(i) for every index in `idx`, each on its own (persistent) thread, and wait for all
void run_on_shard_threads(const std::vector<int>& idx, const std::function<void(int)>& work);
int forward_sharded(int32_t n_shards, const int32_t* device_ids, const void* const* q, const void* const* k, const void* const* v,
                    void* const* o, float* const* lse, const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                    int32_t kernel, void* const* workspaces, const size_t* workspace_bytes, void* const* streams);

// ---- fa_time_forward{,_graph} (fa_timing.cpp)
int time_forward_impl(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
                      int32_t dtype, int32_t kernel, void* stream, int32_t warmup, int32_t iters, float* ms_per_forward, bool graph_replay);

}  // namespace fa_host
