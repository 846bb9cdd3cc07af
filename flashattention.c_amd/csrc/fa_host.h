// fa_host.h -- declarations shared by the host-side translation units behind include/flashattn_amd.h (round 5: fa_api.cpp was one
// 1400-line file):
//   fa_plan.cpp      argument validation, the parameter block, key-split arithmetic, make_plan (which launches, how much scratch)
//   fa_slots.cpp     the report word of an fp32 FA_KERNEL_AUTO forward: per-device slot tables, capture slots; private scratch pools; counters
//   fa_launch.cpp    launch(): one forward = its plan executed on a stream (key shares + combine, the ablation library's chains)
//   fa_shard.cpp     fa_forward_sharded: one persistent host thread per shard
//   fa_timing.cpp    fa_time_forward{,_graph}: event-timed loops
//   fa_selftest.cpp  (sanitizer build only) device-free self-test of plan + slot table
//   fa_api.cpp       the extern "C" entry points, nothing else
// Host-side counterpart of forward() + run_flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:590-617).
#pragma once
#include "../../include/flashattn_amd.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "fa_kernels.h"

namespace fa_host {

// ---- errors, validation, parameter block (fa_plan.cpp)
extern thread_local char g_err[512];
int fail(int code, const char* fmt, ...);
bool aligned16(const void* p);
bool head_dim_supported(int d);
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
constexpr size_t kWsHeader = 256;   // first bytes of a workspace: the chain's flag word (and alignment of what follows); a chained call
                                    // without other scratch asks for just these bytes
struct Plan {
    int status = FA_OK;   // FA_OK, or the error fail() recorded
    Route route = kRouteNaive;
    int S = 1;            // key-split factor
    int terms = 1;        // fp16-P chain: fp16 terms of P (1 = FA_KERNEL_P16, 2 = FA_KERNEL_P16X2 and the AUTO choice)
    size_t v16_off = 0, v16_bytes = 0, part_off = 0, part_bytes = 0, total = 0;   // workspace layout (total = 0: no scratch)
};
inline size_t align256(size_t x) { return (x + 255u) & ~(size_t)255u; }
// scratch_ok: scratch is available to this launch (a workspace was passed, or the stream is not capturing)
Plan make_plan(const fa::FwdParams& p, int32_t d, int32_t causal, int32_t dtype, int32_t kernel, bool scratch_ok);

// ---- report words, slot tables, pools, counters (fa_slots.cpp)
constexpr int kFlagSlots = 16384;          // eager slots [0, kEagerSlots), capture slots behind them
constexpr int kEagerSlots = 8192;
constexpr int kMaxDevices = 64;
// process-wide counters behind fa_get_stats()
struct Stats {
    std::atomic<uint64_t> forwards{0}, chains{0}, chains_degraded{0}, scratch_replans{0}, slot_evictions{0}, capture_slots_recycled{0};
};
extern Stats g_stats;

struct FlagRef {
    uint32_t* word = nullptr;
    uint32_t serial = 0;
    unsigned long long* stats = nullptr;   // two 64-bit words of the same slot (nullptr for a workspace word)
    hipEvent_t done = nullptr;             // eager slot: recorded behind the chain's last launch (see SlotTable)
    int dev = -1, eager_slot = -1;         // ... of this device's table
};
extern thread_local FlagRef t_last_flag;   // chain state of this thread's most recent forward (fa_last_forward_route)
extern thread_local int t_last_chain;      // 0 = no chain, 1 = fp32 guard, 2 = fp16-P (ablation library)
extern thread_local int t_last_route;      // >= 0: the route of the last chain, read before its workspace went away (fa_time_forward*)

int current_device();
bool stream_is_capturing(hipStream_t stream);
// The two slow-path counters of FwdParams::cliffs (a pair of device words per GPU, bumped by the kernels with device-scope atomics) and
// what fa_get_stats() reads from them (a blocking copy per device in use).
unsigned long long* cliff_counters();
unsigned long long cliff_count(int which);
uint32_t next_serial();

// The slots of one device.  Eager slots are keyed by stream (hipStreamPerThread is one handle for a different stream in every thread:
// those chains are keyed by a per-thread number instead).  The mutex is held from taking a slot to the chain's last launch, so two host
// threads feeding one stream cannot interleave their chains' kernels either.
struct EagerSlot {
    uint64_t key = 0;            // stream handle, or (1 << 63) | thread number for hipStreamPerThread
    hipEvent_t done = nullptr;   // created on first use; recorded behind every chain of this slot
    int state = 0;               // 0 = no chain since the slot was (re)assigned; 1 = `done` recorded behind its last chain; 2 = a chain
                                 // is (or was) in flight without an event: the slot never changes hands
    uint64_t tick = 0;           // last use (LRU)
};
struct SlotTable {
    std::mutex mu;
    std::vector<EagerSlot> eager;                  // index = slot number, grows to kEagerSlots
    std::unordered_map<uint64_t, int> by_key;
    uint64_t tick = 0;
    std::vector<int> free_capture;                 // capture slots given back by destroyed graphs
    int next_capture = 0;
};
extern SlotTable g_slots[kMaxDevices];

// ---- the table's logic, free of HIP calls (exercised under ASan / UBSan by fa_host_selftest in the sanitizer build) ----------------
// A capture slot: one given back by a destroyed graph, else a fresh one; -1 = none left.
inline int take_capture_slot(SlotTable& tb)
{
    if (!tb.free_capture.empty()) {
        const int k = tb.free_capture.back();
        tb.free_capture.pop_back();
        return k;
    }
    if (tb.next_capture < kFlagSlots - kEagerSlots) return tb.next_capture++;
    return -1;
}
// The eager slot of `key`: its own, a fresh one, or -- table full -- the least recently used slot whose last chain has completed
// (`completed(slot)`; a few candidates at most: a slot found busy is moved to the young end).  -1 = none to be had.
template <class Completed>
int take_eager_slot(SlotTable& tb, uint64_t key, Completed completed)
{
    int slot = -1;
    auto it = tb.by_key.find(key);
    if (it != tb.by_key.end()) {
        slot = it->second;
    } else if ((int)tb.eager.size() < kEagerSlots) {
        slot = (int)tb.eager.size();
        tb.eager.emplace_back();
    } else {
        for (int attempt = 0; attempt < 16 && slot < 0; ++attempt) {
            int lru = 0;
            for (int i = 1; i < (int)tb.eager.size(); ++i)
                if (tb.eager[i].tick < tb.eager[lru].tick) lru = i;
            EagerSlot& c = tb.eager[lru];
            if (c.state == 0 || (c.state == 1 && completed(lru))) {
                tb.by_key.erase(c.key);
                slot = lru;
                g_stats.slot_evictions.fetch_add(1, std::memory_order_relaxed);
            } else {
                c.tick = ++tb.tick;
            }
        }
        if (slot < 0) return -1;
    }
    EagerSlot& e = tb.eager[slot];
    if (e.key != key || tb.by_key.find(key) == tb.by_key.end()) {
        e.key = key;
        e.state = 0;
        tb.by_key[key] = slot;
    }
    e.tick = ++tb.tick;
    return slot;
}

// The flag word of a chain that has no workspace.  false = no slot to be had (or no device symbol): the caller then launches the
// always-correct kernel of the chain alone.  `hold` keeps the device's slot table locked until the chain is enqueued.
bool next_flag(FlagRef& f, hipStream_t stream, bool capturing, std::unique_lock<std::mutex>& hold);
// behind the chain's last launch, table still locked (`hold`): the event that tells when this slot may change hands
void chain_enqueued(const FlagRef& f, hipStream_t stream);
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
