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
#include <condition_variable>
#include <functional>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

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

// ---- the report word of a forward (and the ablation library's conditional launch chains) ---------------------------------------------
// Until round 4 the FA_KERNEL_AUTO path of fp32 tensors was a chain of launches in which the exact fp32 kernel, queued behind the split
// kernel, ran or skipped itself depending on a device word the split kernel raised when its logits were too wide for 16-bit operand
// terms.  Now the split kernel redoes a workgroup's rows in fp32 arithmetic itself when its operands leave the range fp16 terms hold (flag_mode 4; fa_split_kernel.h) -- ONE launch,
// and a hostile slab costs its own tiles only -- and the word merely REPORTS that some workgroup did (fa_last_forward_route).  The
// machinery below is what keeps two calls from ever sharing a word; the ablation library's chains (fp16-P kernels, the static-slot fp32
// kernel) still depend on it for correctness, the product for the truth of its report:
// "Raised" means "the word equals this call's serial number" (serials are unique per call), so a word never needs clearing between
// eager calls.  WHERE the word lives:
//   * a call that runs with a caller-owned workspace (fa_forward_ws; fa_workspace_bytes() reports at least the 256-byte header for
//     every such call) keeps its word in the first bytes of that workspace -- the caller's buffer, in use by one forward at a time
//     like every other buffer of the call;
//   * every other eager call takes the slot of its (device, stream) pair from a per-device table, and the table's mutex is held while
//     the call is enqueued: calls that share a slot are on one stream, one after the other.  An event recorded behind each call tells
//     when its slot may change hands: when the table is full the least recently used slot whose last call has COMPLETED is given to the
//     new stream (a long-running host that creates and destroys streams never runs out; round 3 handed slots out once);
//   * a call enqueued while its stream is CAPTURING takes a slot of its own and starts with a memset node that clears the word, so
//     replays of the graph report independently of each other.  The slot goes back to the table when the graph -- and every executable
//     instantiated from it -- has been destroyed (a hipUserObject retained by the capturing graph; where the runtime refuses that, the
//     slot is simply never reused).
// When no slot can be had the product launches the same kernel without a word (route 0 is reported; fa_get_stats() counts those calls);
// a chain of the ablation library launches its always-correct kernel alone.
// (Round 2 indexed a 4096-slot ring with serial % 4096: a chain whose serial was congruent -- every 4096th eager call, or a replayed
// graph -- could overwrite a raised word between the other chain's primary and its fallback kernel.)
constexpr int kFlagSlots = 16384;          // eager slots [0, kEagerSlots), capture slots behind them
constexpr int kEagerSlots = 8192;
__device__ uint32_t g_flag_ring[kFlagSlots];
__device__ unsigned long long g_stat_ring[kFlagSlots][2];   // pre-pass maxima of the t3 chain, tagged with the call's serial (experiments/fa_cvt.hip)
constexpr int kMaxDevices = 64;
std::atomic<uint32_t*> g_ring_base[kMaxDevices];
std::atomic<unsigned long long*> g_stat_base[kMaxDevices];
std::atomic<uint32_t> g_serial{1};

// process-wide counters behind fa_get_stats()
struct Stats {
    std::atomic<uint64_t> forwards{0}, chains{0}, chains_degraded{0}, scratch_replans{0}, slot_evictions{0}, capture_slots_recycled{0};
};
Stats g_stats;

struct FlagRef {
    uint32_t* word = nullptr;
    uint32_t serial = 0;
    unsigned long long* stats = nullptr;   // two 64-bit words of the same slot (nullptr for a workspace word)
    hipEvent_t done = nullptr;             // eager slot: recorded behind the chain's last launch (see SlotTable)
    int dev = -1, eager_slot = -1;         // ... of this device's table
};
thread_local FlagRef t_last_flag;   // chain state of this thread's most recent forward (fa_last_forward_route)
thread_local int t_last_chain = 0;  // 0 = no chain, 1 = fp32 guard, 2 = fp16-P (ablation library)
thread_local int t_last_route = -1; // >= 0: the route of the last chain, read before its workspace went away (fa_time_forward*)

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}

bool stream_is_capturing(hipStream_t stream)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

uint32_t next_serial()
{
    uint32_t serial = g_serial.fetch_add(1, std::memory_order_relaxed);
    if (serial == 0) serial = g_serial.fetch_add(1, std::memory_order_relaxed);   // 0 is the ring's initial content
    return serial;
}

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
SlotTable g_slots[kMaxDevices];
std::atomic<uint64_t> g_thread_numbers{1};
thread_local uint64_t t_thread_number = 0;

// capture slots come back through a hipUserObject the capturing graph retains: its destructor runs when the graph and every executable
// instantiated from it are gone.  (No HIP call is allowed in there: it only pushes a number onto a list.)
struct CaptureSlotToken {
    int dev, slot;
};
void release_capture_slot(void* ptr)
{
    CaptureSlotToken* t = static_cast<CaptureSlotToken*>(ptr);
    if (t->dev >= 0 && t->dev < kMaxDevices) {
        std::lock_guard<std::mutex> g(g_slots[t->dev].mu);
        g_slots[t->dev].free_capture.push_back(t->slot);
        g_stats.capture_slots_recycled.fetch_add(1, std::memory_order_relaxed);
    }
    delete t;
}
// true: the graph being captured on `stream` now owns `slot` (it returns it when it dies)
bool tie_capture_slot_to_graph(hipStream_t stream, int dev, int slot)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    if (hipStreamGetCaptureInfo_v2(stream, &st, &id, &graph, nullptr, nullptr) != hipSuccess || graph == nullptr) {
        (void)hipGetLastError();
        return false;
    }
    CaptureSlotToken* tok = new CaptureSlotToken{dev, slot};
    hipUserObject_t uo = nullptr;
    if (hipUserObjectCreate(&uo, tok, release_capture_slot, 1, hipUserObjectNoDestructorSync) != hipSuccess || uo == nullptr) {
        (void)hipGetLastError();
        delete tok;
        return false;
    }
    if (hipGraphRetainUserObject(graph, uo, 1, hipGraphUserObjectMove) != hipSuccess) {
        (void)hipGetLastError();
        tok->dev = -1;                       // the destructor then only frees the token
        (void)hipUserObjectRelease(uo, 1);
        return false;
    }
    return true;
}

// ---- the table's logic, free of HIP calls (exercised under ASan / UBSan by fa_host_selftest in the sanitizer build) ----------------
// A capture slot: one given back by a destroyed graph, else a fresh one; -1 = none left.
int take_capture_slot(SlotTable& tb)
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

// The flag word of a chain that has no workspace (see above).  false = no slot to be had (or no device symbol): the caller then launches
// the always-correct kernel of the chain alone.  `hold` keeps the device's slot table locked until the chain is enqueued.
bool next_flag(FlagRef& f, hipStream_t stream, bool capturing, std::unique_lock<std::mutex>& hold)
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
    SlotTable& tb = g_slots[dev];
    hold = std::unique_lock<std::mutex>(tb.mu);
    int slot = -1;
    f.done = nullptr;
    if (capturing) {
        const int k = take_capture_slot(tb);
        if (k < 0) {
            hold.unlock();
            return false;
        }
        slot = kEagerSlots + k;
        hold.unlock();                                   // (the runtime may run a user-object destructor -- which takes this mutex -- inside its calls)
        (void)tie_capture_slot_to_graph(stream, dev, k);   // not tied: the slot is never handed out again, as in round 3
        hold.lock();
    } else {
        uint64_t key = reinterpret_cast<uint64_t>(stream);
        if (stream == hipStreamPerThread) {
            if (t_thread_number == 0) t_thread_number = g_thread_numbers.fetch_add(1, std::memory_order_relaxed);
            key = (1ull << 63) | t_thread_number;
        }
        slot = take_eager_slot(tb, key, [&](int i) {
            if (hipEventQuery(tb.eager[i].done) == hipSuccess) return true;
            (void)hipGetLastError();
            return false;
        });
        if (slot < 0) {
            hold.unlock();
            return false;
        }
        EagerSlot& e = tb.eager[slot];
        if (e.done == nullptr && hipEventCreateWithFlags(&e.done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            e.done = nullptr;   // without an event the slot can never change hands safely: it simply stays with its stream
        }
        f.done = e.done;
        f.dev = dev;
        f.eager_slot = slot;
    }
    f.word = base + slot;
    f.serial = next_serial();
    f.stats = g_stat_base[dev].load(std::memory_order_acquire) + 2 * (size_t)slot;
    return true;
}
// behind the chain's last launch, table still locked (`hold`): the event that tells when this slot may change hands
void chain_enqueued(const FlagRef& f, hipStream_t stream)
{
    if (f.dev < 0 || f.eager_slot < 0) return;
    EagerSlot& e = g_slots[f.dev].eager[f.eager_slot];
    if (f.done != nullptr && hipEventRecord(f.done, stream) == hipSuccess) {
        e.state = 1;
    } else {
        (void)hipGetLastError();
        e.state = 2;
    }
}

// ---- scratch ------------------------------------------------------------------------------------------------------------------
// The C ABI proper never allocates: fa_forward_ws runs in a caller-owned workspace whose size fa_workspace_bytes reports.  The
// convenience entry points (fa_forward, fa_forward_ex, the sharded and timing entries) take the same bytes from a PRIVATE
// stream-ordered pool per device (hipMemPoolCreate; its release threshold is ours to raise -- the device's default pool, which the
// host application and torch may be using, is never touched) and return them behind the last kernel that reads them.
struct DevicePool {
    std::atomic<int> state{0};   // 0 = untried, 1 = being created, 2 = ready, 3 = unavailable (plain hipMallocAsync then)
    hipMemPool_t pool = nullptr;
};
DevicePool g_pools[kMaxDevices];

hipMemPool_t private_pool(int dev)
{
    if (dev < 0) return nullptr;
    DevicePool& dp = g_pools[dev];
    int st = dp.state.load(std::memory_order_acquire);
    if (st == 0) {
        int expect = 0;
        if (dp.state.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) {
            hipMemPoolProps props;
            memset(&props, 0, sizeof(props));
            props.allocType = hipMemAllocationTypePinned;
            props.handleTypes = hipMemHandleTypeNone;
            props.location.type = hipMemLocationTypeDevice;
            props.location.id = dev;
            hipMemPool_t pool = nullptr;
            if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool != nullptr) {
                uint64_t keep = ~0ull;   // keep what steady-state calls hand back: they then never reach the driver
                (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
                dp.pool = pool;
                dp.state.store(2, std::memory_order_release);
            } else {
                (void)hipGetLastError();
                dp.state.store(3, std::memory_order_release);
            }
        }
        st = dp.state.load(std::memory_order_acquire);
    }
    while (st == 1) st = dp.state.load(std::memory_order_acquire);
    return st == 2 ? dp.pool : nullptr;
}

hipError_t scratch_alloc(void** ptr, size_t bytes, hipStream_t stream)
{
    hipMemPool_t pool = private_pool(current_device());
    if (pool != nullptr) return hipMallocFromPoolAsync(ptr, bytes, pool, stream);
    return hipMallocAsync(ptr, bytes, stream);
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
int keysplit_factor(const fa::FwdParams& p, int32_t d, int32_t causal, bool f32 = false, bool pb2 = false)
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
size_t align256(size_t x) { return (x + 255u) & ~(size_t)255u; }

// scratch_ok: scratch is available to this launch (a workspace was passed, or the stream is not capturing)
Plan make_plan(const fa::FwdParams& p, int32_t d, int32_t causal, int32_t dtype, int32_t kernel, bool scratch_ok)
{
    Plan pl;
    const KernelSel sel = decode_kernel(kernel);
    if (sel.kind == FA_KERNEL_NAIVE) {
        if (dtype != FA_DTYPE_F32) pl.status = fail(FA_ERR_UNSUPPORTED, "the naive kernel is fp32 only");
        else if (d > 256) pl.status = fail(FA_ERR_UNSUPPORTED, "naive kernel supports head dim <= 256 (got %d)", d);
        pl.route = kRouteNaive;
        return pl;
    }
    if (sel.kind != FA_KERNEL_AUTO && sel.kind != FA_KERNEL_MFMA && sel.kind != FA_KERNEL_SPLIT && sel.kind != FA_KERNEL_P16 && sel.kind != FA_KERNEL_P16X2 &&
        sel.kind != FA_KERNEL_PB2) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "unknown kernel id %d", sel.kind);
        return pl;
    }
    if (!head_dim_supported(d)) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "head dim %d not instantiated for the MFMA kernels (32, 64, 128)", d);
        return pl;
    }
    if (dtype == FA_DTYPE_F32) {
        if (sel.kind == FA_KERNEL_P16 || sel.kind == FA_KERNEL_P16X2 || sel.kind == FA_KERNEL_PB2)
            pl.status = fail(FA_ERR_UNSUPPORTED, "FA_KERNEL_P16 / FA_KERNEL_P16X2 / FA_KERNEL_PB2 are bf16-tensor kernels");
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
            if (scratch_ok) pl.total = kWsHeader;   // the chain's verdict word (a caller-owned workspace keeps it off the slot table)
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
    const bool p16_kind = sel.kind == FA_KERNEL_P16 || sel.kind == FA_KERNEL_P16X2;
#if !FA_ABLATION
    if (p16_kind) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "FA_KERNEL_P16 / FA_KERNEL_P16X2 (P and V in fp16) were replaced by FA_KERNEL_PB2 (P as two bf16 terms: faster, one launch, "
                                             "no scratch) and are built into libflashattn_amd_ablation.so only");
        return pl;
    }
#else
    if (p16_kind && !p16_available(p, d)) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "the fp16-P kernels need dense (bh, n, d) tensors and address a slab with 32-bit byte offsets (slabs below 4 GiB; got n = %d, d = %d)", p.n, d);
        return pl;
    }
    if (p16_kind && !scratch_ok) {
        pl.status = fail(FA_ERR_UNSUPPORTED, "FA_KERNEL_P16 / FA_KERNEL_P16X2 need scratch, and stream-ordered allocations are not reliable inside a captured graph on this runtime: "
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
        pl.terms = sel.kind == FA_KERNEL_P16 ? 1 : 2;
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
hipError_t launch_f32_keysplit(const fa::FwdParams& p0, int32_t d, int32_t causal, int S, char* part, hipStream_t stream, bool exact = false)
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
    hipError_t e = exact ? fa::launch_fwd_f32(p, d, causal ? 1 : 0, 0, stream) : fa::launch_f32_split(p, d, causal ? 1 : 0, 0, stream);
    fa::FwdParams pc = p0;
    pc.flag_mode = 0;
    if (e == hipSuccess) e = fa::launch_combine_splits(pc, o_part, lse_part, S, d, 1, stream);
    return e;
}

#if FA_ABLATION
// bf16 tensors, fp16 P (ablation library): V -> fp16 copy in scratch, fp16-P kernel, split kernel as the conditional fallback
hipError_t launch_p16_chain(const fa::FwdParams& p0, int32_t d, int32_t causal, int32_t out_f32, const Plan& pl, char* ws, const FlagRef& f,
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
hipError_t launch_f32_t3_chain(const fa::FwdParams& p0, int32_t d, char* scratch, const FlagRef& f, hipStream_t stream, bool guarded, int abl)
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

// One forward.  ws == nullptr && !ws_mode: a convenience entry point -- scratch, if the plan wants any, comes from the private pool
// (never while the stream is capturing: the plan is then made without scratch).  ws_mode: the caller's workspace or nothing.
int launch(const fa::FwdParams& p, int32_t d, int32_t causal, int32_t dtype, int32_t kernel, hipStream_t stream, void* ws = nullptr,
           size_t ws_bytes = 0, bool ws_mode = false)
{
    const KernelSel sel = decode_kernel(kernel);
    t_last_chain = 0;
    t_last_route = -1;
    g_stats.forwards.fetch_add(1, std::memory_order_relaxed);
    const bool capturing = stream_is_capturing(stream);
    Plan pl = make_plan(p, d, causal, dtype, kernel, ws_mode ? true : !capturing);
    if (pl.status != FA_OK) return pl.status;
    char* scratch = static_cast<char*>(ws);
    bool owned = false;
    if (pl.total > 0 && ws_mode) {
        if (scratch == nullptr || ws_bytes == 0) {
            // a binder that skips fa_workspace_bytes(): FA_KERNEL_AUTO runs without scratch (the unsplit launch; the verdict word of a
            // chain from the slot table) instead of refusing -- an explicit kernel that cannot do without scratch still says so
            scratch = nullptr;
            if (sel.kind != FA_KERNEL_AUTO) return fail(FA_ERR_INVALID_ARGUMENT, "this kernel choice needs a workspace of fa_workspace_bytes() = %zu bytes", pl.total);
            pl = make_plan(p, d, causal, dtype, kernel, false);
            if (pl.status != FA_OK) return pl.status;
            g_stats.scratch_replans.fetch_add(1, std::memory_order_relaxed);
        } else {
            if (ws_bytes < pl.total) return fail(FA_ERR_INVALID_ARGUMENT, "workspace of %zu bytes is too small: this call needs fa_workspace_bytes() = %zu", ws_bytes, pl.total);
            if ((reinterpret_cast<uintptr_t>(scratch) & 255u) != 0) return fail(FA_ERR_INVALID_ARGUMENT, "workspace must be 256-byte aligned");
        }
    } else if (pl.total > kWsHeader) {   // (a header-only plan needs no allocation: the word of an owned chain comes from the slot table)
        void* ptr = nullptr;
        const hipError_t ea = scratch_alloc(&ptr, pl.total, stream);
        if (ea != hipSuccess || ptr == nullptr) {
            (void)hipGetLastError();
            // the scratch paths are optimisations (and the fp16-P kernels of the ablation library an explicit request): AUTO falls back to
            // the kernels without scratch
            if (sel.kind == FA_KERNEL_P16 || sel.kind == FA_KERNEL_P16X2 || pl.route == kRouteF32T3)
                return fail(FA_ERR_HIP, "stream-ordered allocation of %zu scratch bytes failed: %s", pl.total, hipGetErrorString(ea));
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
    // The chain's flag word: in the CALLER's workspace when the call has one, else the slot of (device, stream) / a capture slot.  Never
    // in scratch this call owns: that goes back to the pool behind the last kernel, and fa_last_forward_route() reads the word later
    // (round 3 put it there and read freed memory).
    std::unique_lock<std::mutex> hold;   // slot table of the device, locked from taking a slot to the chain's last launch
    auto chain_flag = [&](FlagRef& f) -> bool {
        bool ok;
        if (ws_mode && scratch != nullptr && ws_bytes >= kWsHeader) {
            f = FlagRef{};
            f.word = reinterpret_cast<uint32_t*>(scratch);
            f.serial = next_serial();
            ok = true;
        } else {
            ok = next_flag(f, stream, capturing, hold);
        }
        // a captured chain is replayed with the same serial: clear the word first, or a verdict of an earlier replay would stand
        if (ok && capturing) ok = hipMemsetAsync(f.word, 0, sizeof(uint32_t), stream) == hipSuccess;
        g_stats.chains.fetch_add(1, std::memory_order_relaxed);
        if (!ok) g_stats.chains_degraded.fetch_add(1, std::memory_order_relaxed);
        return ok;
    };
    switch (pl.route) {
        case kRouteNaive: e = fa::launch_naive_f32(p, d, c, stream); break;
        case kRouteF32Exact:
            if (pl.S > 1) e = launch_f32_keysplit(p, d, causal, pl.S, scratch + pl.part_off, stream, true);
            else e = fa::launch_fwd_f32(p, d, c, sel.variant, stream);
            break;
        case kRouteF32Split: e = fa::launch_f32_split(p, d, c, sel.variant, stream); break;
        case kRouteF32Guarded: {   // split products behind the range guard: ONE launch (round 4) -- a workgroup whose operands leave what fp16
            FlagRef f;              // terms hold (or met a NaN) redoes its own rows in fp32 arithmetic inside the kernel (flag_mode 4).  The word
            const bool have = chain_flag(f);   // only REPORTS that (fa_last_forward_route); without one the launch is the same
            fa::FwdParams pg = p;
            pg.flag = have ? f.word : nullptr;
            pg.flag_serial = have ? f.serial : 0u;
            pg.flag_mode = 4;
            if (pl.S > 1) e = launch_f32_keysplit(pg, d, causal, pl.S, scratch + pl.part_off, stream);   // (every share guards its own keys)
            else e = fa::launch_f32_split(pg, d, c, 0, stream);
            if (have) {
                chain_enqueued(f, stream);
                if (e == hipSuccess) {
                    t_last_flag = f;
                    t_last_chain = 1;
                }
            }
            break;
        }
#if FA_ABLATION
        case kRouteF32T3: {
            FlagRef f;
            if (!next_flag(f, stream, capturing, hold)) return fail(FA_ERR_HIP, "no device flag slot (hipGetSymbolAddress failed or slots exhausted)");
            e = launch_f32_t3_chain(p, d, scratch + pl.part_off, f, stream, sel.variant == 8, sel.variant >= 16 ? sel.variant - 16 : 0);
            chain_enqueued(f, stream);
            if (e == hipSuccess) {
                t_last_flag = f;
                t_last_chain = 1;
            }
            break;
        }
        case kRouteP16Chain: {
            FlagRef f;
            if (!chain_flag(f)) {
                e = fa::launch_bf16_split(p, d, c, out_f32, 0, stream);
                break;
            }
            e = launch_p16_chain(p, d, causal, out_f32, pl, scratch, f, stream);
            chain_enqueued(f, stream);
            if (e == hipSuccess) {
                t_last_flag = f;
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
    if (hold.owns_lock()) hold.unlock();
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

// ---- fa_forward_sharded: one persistent host thread per shard index -----------------------------------------------------------------
// A forward can be a chain of launches plus a pool allocation; enqueued from one thread the last device would start a whole table's
// worth of host time behind the first.  Round 3 created and joined a std::thread per shard on every call (tens of microseconds each on
// the path of a ~0.3 ms launch: ADVICE r03); now worker i is created on first use, sleeps on a condition variable between calls and is
// never destroyed (the pool is leaked on purpose: no join at process exit, where HIP may already be gone).  One sharded call at a
// time uses the pool (g_shard_call); a second caller runs meanwhile on threads of its own, as before.
struct ShardWorker {
    std::mutex mu;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = false, started = false;
    std::thread th;
    void loop()
    {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return has_job; });
            std::function<void()> j = std::move(job);
            has_job = false;
            lk.unlock();
            j();
            lk.lock();
            done = true;
            cv.notify_all();
        }
    }
};
struct ShardPool {
    std::mutex call_mu;                      // one sharded call at a time
    ShardWorker workers[kMaxDevices];
};
ShardPool* shard_pool()
{
    static ShardPool* pool = new ShardPool();   // never freed
    return pool;
}
// run work(i) for every index in `idx`, each on its own thread, and wait for all of them
void run_on_shard_threads(const std::vector<int>& idx, const std::function<void(int)>& work)
{
    ShardPool* pool = shard_pool();
    std::unique_lock<std::mutex> call(pool->call_mu, std::try_to_lock);
    if (!call.owns_lock() || (int)idx.size() > kMaxDevices) {   // the pool is busy with another caller's shards: threads of our own
        std::vector<std::thread> th;
        for (int i : idx) {
            try {
                th.emplace_back(work, i);
            } catch (const std::exception&) {   // no thread to be had (nothing may cross the extern "C" boundary): this shard from here
                work(i);
            }
        }
        for (auto& t : th) t.join();
        return;
    }
    std::vector<int> queued;
    for (size_t k = 0; k < idx.size(); ++k) {
        ShardWorker& w = pool->workers[k];
        const int i = idx[k];
        bool ok = true;
        if (!w.started) {
            try {
                w.th = std::thread([&w] { w.loop(); });
                w.th.detach();
                w.started = true;
            } catch (const std::exception&) {
                ok = false;
            }
        }
        if (!ok) {
            work(i);
            continue;
        }
        {
            std::lock_guard<std::mutex> lk(w.mu);
            w.job = [&work, i] { work(i); };
            w.has_job = true;
            w.done = false;
        }
        w.cv.notify_all();
        queued.push_back((int)k);
    }
    for (int k : queued) {
        ShardWorker& w = pool->workers[k];
        std::unique_lock<std::mutex> lk(w.mu);
        w.cv.wait(lk, [&] { return w.done; });
    }
}

}  // namespace

#if FA_HOST_TEST
// The sanitizer build's (build.py --sanitize: this file with -fsanitize=address,undefined -DFA_HOST_TEST=1) self-test of the host
// logic that needs no device: the plan over a shape grid (sizes consistent between the size query and the plan, layouts inside the
// total), the key-split arithmetic, and the slot table (allocation, reuse by key, LRU eviction with a scripted completion oracle,
// capture slots taken and given back) from several threads.  Returns 0, or the number of the first failed check.
extern "C" int fa_host_selftest(void)
{
    int check = 0;
#define FA_CHECK(cond) do { ++check; if (!(cond)) return check; } while (0)
    // ---- plans
    const int dtypes[] = {FA_DTYPE_F32, FA_DTYPE_BF16, FA_DTYPE_BF16_OUT_F32};
    const int kinds[] = {FA_KERNEL_AUTO, FA_KERNEL_MFMA, FA_KERNEL_SPLIT, FA_KERNEL_PB2, FA_KERNEL_NAIVE, FA_KERNEL_P16X2, 9};
    const int64_t bhs[] = {1, 2, 3, 8, 16, 33, 128, 1024, 70000}, ns[] = {1, 31, 300, 1024, 1100, 2048, 4095, 4096, 5000, 8192, 16384, 40000, 1 << 24};
    for (int dt : dtypes) for (int kind : kinds) for (int d : {32, 64, 128, 48}) for (int causal : {0, 1}) for (int64_t bh : bhs) for (int64_t n : ns) {
        const fa::FwdParams p = make_params(nullptr, nullptr, nullptr, nullptr, nullptr, bh, n, d, 1.0f);
        for (bool scratch_ok : {false, true}) {
            const Plan pl = make_plan(p, d, causal, dt, kind, scratch_ok);
            if (pl.status != FA_OK) continue;
            FA_CHECK(scratch_ok || pl.total == 0);                                   // no scratch, no bytes
            FA_CHECK(pl.S >= 1 && pl.S <= 8);
            FA_CHECK(pl.total % 256 == 0);
            FA_CHECK(pl.part_bytes == 0 || (pl.part_off >= kWsHeader && pl.part_off + pl.part_bytes <= pl.total));
            FA_CHECK(pl.v16_bytes == 0 || (pl.v16_off >= kWsHeader && pl.v16_off + pl.v16_bytes <= pl.total));
            FA_CHECK(pl.part_bytes == 0 || pl.v16_bytes == 0 || pl.v16_off + pl.v16_bytes <= pl.part_off);
            if (pl.S > 1) {
                const int rows = keysplit_rows(p, pl.S, causal);
                FA_CHECK((int64_t)rows * pl.S >= n && (int64_t)rows * (pl.S - 1) < n);   // every share owns a key, all keys covered
                FA_CHECK(rows % (causal ? 256 : 64) == 0);
                FA_CHECK(pl.part_bytes == (size_t)pl.S * bh * n * d * 4u + (size_t)pl.S * bh * n * 4u);
            }
            if (scratch_ok) FA_CHECK(fa_workspace_bytes(bh, n, d, causal, dt, kind) == pl.total);
        }
    }
    // ---- slot table: a private table, scripted completion
    {
        static SlotTable tb;    // (large: not on the stack)
        std::vector<char> busy(kEagerSlots, 0);
        auto completed = [&](int i) { return busy[i] == 0; };
        for (int i = 0; i < kEagerSlots; ++i) {
            const int sl = take_eager_slot(tb, 0x1000 + (uint64_t)i, completed);
            FA_CHECK(sl == i);
            tb.eager[sl].state = 1;
            busy[sl] = (i % 2) ? 1 : 0;     // odd slots: their chain is "still running"
        }
        FA_CHECK(take_eager_slot(tb, 0x1000 + 77, completed) == 77);                  // a known key keeps its slot
        int evicted = 0;
        for (int i = 0; i < 3000; ++i) {                                               // new keys: only completed slots change hands
            const int sl = take_eager_slot(tb, 0x900000 + (uint64_t)i, completed);
            FA_CHECK(sl >= 0 && sl < kEagerSlots && busy[sl] == 0);
            FA_CHECK(tb.eager[sl].key == 0x900000 + (uint64_t)i && tb.eager[sl].state == 0);
            tb.eager[sl].state = 1;
            ++evicted;
        }
        FA_CHECK((int)tb.by_key.size() == kEagerSlots);
        for (const auto& kv : tb.by_key) FA_CHECK(tb.eager[kv.second].key == kv.first);
        for (int i = 0; i < kEagerSlots; ++i) busy[i] = 1, tb.eager[i].state = 1;
        FA_CHECK(take_eager_slot(tb, 0xdead0000, completed) == -1);                    // everything in flight: no slot, the caller degrades
        tb.eager[5].state = 2;
        tb.eager[5].tick = 0;                                                          // the two oldest slots: 5 (no event) and 6
        tb.eager[6].tick = 1;
        busy[5] = 0;
        busy[6] = 0;
        FA_CHECK(take_eager_slot(tb, 0xdead0001, completed) == 6);                     // a slot without an event never changes hands
        for (int i = 0; i < kFlagSlots - kEagerSlots; ++i) FA_CHECK(take_capture_slot(tb) == i);
        FA_CHECK(take_capture_slot(tb) == -1);
        tb.free_capture.push_back(123);
        FA_CHECK(take_capture_slot(tb) == 123 && take_capture_slot(tb) == -1);
    }
    // ---- the same table logic from eight threads (the mutex the real callers hold)
    {
        static SlotTable tb;
        std::atomic<int> bad{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 8; ++t)
            th.emplace_back([&, t] {
                for (int i = 0; i < 4000; ++i) {
                    std::lock_guard<std::mutex> g(tb.mu);
                    const uint64_t key = ((uint64_t)t << 32) | (uint64_t)(i % 1500);
                    const int sl = take_eager_slot(tb, key, [](int) { return true; });
                    if (sl < 0 || tb.eager[sl].key != key) bad.fetch_add(1);
                    else tb.eager[sl].state = 1;
                    if (i % 7 == 0) {
                        const int c = take_capture_slot(tb);
                        if (c >= 0) tb.free_capture.push_back(c);
                    }
                }
            });
        for (auto& x : th) x.join();
        FA_CHECK(bad.load() == 0);
    }
    // ---- argument validation without a device
    {
        alignas(256) static char buf[4][8192];
        FA_CHECK(fa_forward(buf[0], buf[1], buf[2], buf[3], 1, 32, 48, 1.0f, 0, FA_DTYPE_F32, nullptr) == FA_ERR_UNSUPPORTED);
        FA_CHECK(fa_forward(buf[0], buf[1], buf[2], buf[0], 1, 32, 64, 1.0f, 0, FA_DTYPE_F32, nullptr) == FA_ERR_INVALID_ARGUMENT);
        FA_CHECK(fa_forward_ws(buf[0], buf[1], buf[2], buf[3], nullptr, 1, 8, 64, 1.0f, 0, FA_DTYPE_F32, FA_KERNEL_AUTO, buf[3], 4096, nullptr) == FA_ERR_INVALID_ARGUMENT);
        fa_stats st;
        FA_CHECK(fa_get_stats(&st) == FA_OK && st.eager_slots_per_device == (uint64_t)kEagerSlots);
    }
#undef FA_CHECK
    return 0;
}
#endif

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

size_t fa_workspace_bytes(int64_t bh, int64_t n, int32_t d, int32_t causal, int32_t dtype, int32_t kernel)
{
    if (bh < 1 || n < 1 || bh > 0x7fffffffLL || n > (1LL << 24) || d < 1) return 0;
    if (dtype != FA_DTYPE_F32 && dtype != FA_DTYPE_BF16 && dtype != FA_DTYPE_BF16_OUT_F32) return 0;
    char keep[sizeof(g_err)];
    memcpy(keep, g_err, sizeof(g_err));   // a size query does not disturb fa_last_error()
    const fa::FwdParams p = make_params(nullptr, nullptr, nullptr, nullptr, nullptr, bh, n, d, 1.0f);
    const Plan pl = make_plan(p, d, causal, dtype, kernel, true);
    memcpy(g_err, keep, sizeof(g_err));
    return pl.status == FA_OK ? pl.total : 0;
}

int fa_forward_ws(const void* q, const void* k, const void* v, void* o, float* lse, int64_t bh, int64_t n, int32_t d, float scale,
                  int32_t causal, int32_t dtype, int32_t kernel, void* workspace, size_t workspace_bytes, void* stream)
{
    g_err[0] = 0;
    if (int rc = validate_common(q, k, v, o, bh, n, d, scale, dtype)) return rc;
    if (workspace != nullptr && workspace_bytes > 0) {   // the workspace is written by the launch: it must not overlap a tensor of the call
        const uint64_t elems = (uint64_t)bh * (uint64_t)n * (uint64_t)d;
        const uint64_t in_bytes = elems * (dtype == FA_DTYPE_F32 ? 4u : 2u), out_bytes = elems * (dtype == FA_DTYPE_BF16 ? 2u : 4u);
        const uintptr_t wb = reinterpret_cast<uintptr_t>(workspace);
        const struct { const void* t; uint64_t bytes; } tensors[] = {{q, in_bytes}, {k, in_bytes}, {v, in_bytes}, {o, out_bytes}};
        for (const auto& t : tensors) {
            const uintptr_t tb = reinterpret_cast<uintptr_t>(t.t);
            if (wb < tb + t.bytes && tb < wb + workspace_bytes) return fail(FA_ERR_INVALID_ARGUMENT, "workspace overlaps a tensor of the call");
        }
    }
    const fa::FwdParams p = make_params(q, k, v, o, lse, bh, n, d, scale);
    return launch(p, d, causal, dtype, kernel, static_cast<hipStream_t>(stream), workspace, workspace_bytes, true);
}

int fa_forward_sharded_ex(int32_t n_shards, const int32_t* device_ids, const void* const* q, const void* const* k, const void* const* v,
                          void* const* o, float* const* lse, const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                          int32_t kernel, void* const* workspaces, const size_t* workspace_bytes, void* const* streams)
{
    g_err[0] = 0;
    if (n_shards < 1 || !device_ids || !q || !k || !v || !o || !bh)
        return fail(FA_ERR_INVALID_ARGUMENT, "fa_forward_sharded: bad shard table");
    if ((workspaces == nullptr) != (workspace_bytes == nullptr))
        return fail(FA_ERR_INVALID_ARGUMENT, "fa_forward_sharded_ex: workspaces and workspace_bytes come together (both NULL: the convenience path's private pools)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FA_ERR_NO_DEVICE, "no HIP device visible");
    // a shard table that names one device twice is almost always a bug of the caller (two shards then queue up behind each other
    // instead of running side by side); FA_ALLOW_SAME_DEVICE=1 lifts the check for single-GPU test boxes
    const char* same = getenv("FA_ALLOW_SAME_DEVICE");
    const bool allow_same = same != nullptr && same[0] == '1';
    for (int i = 0; i < n_shards; ++i) {
        if (device_ids[i] < 0 || device_ids[i] >= ndev)
            return fail(FA_ERR_INVALID_ARGUMENT, "shard %d: device %d not in [0, %d)", i, device_ids[i], ndev);
        if (bh[i] < 0) return fail(FA_ERR_INVALID_ARGUMENT, "shard %d: negative bh", i);
        if (bh[i] == 0) continue;
        if (int rc = validate_common(q[i], k[i], v[i], o[i], bh[i], n, d, scale, dtype)) return rc;
        for (int j = 0; j < i && !allow_same; ++j)
            if (bh[j] > 0 && device_ids[j] == device_ids[i])
                return fail(FA_ERR_INVALID_ARGUMENT, "shards %d and %d both name device %d (set FA_ALLOW_SAME_DEVICE=1 to allow it)", j, i, device_ids[i]);
    }
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess) return fail(FA_ERR_HIP, "hipGetDevice failed");
    t_last_chain = 0;   // the shards' chains belong to their worker threads: fa_last_forward_route() of this thread reports "no chain"
    t_last_route = -1;
    // One host thread per shard (run_on_shard_threads).  The current device is per host thread in HIP, so the workers do not disturb the
    // caller's; each worker's scratch comes from its shard's workspace, or from its own device's private pool.
    std::vector<int> rcs((size_t)n_shards, FA_OK);
    std::vector<std::string> msgs((size_t)n_shards);
    const std::function<void(int)> work = [&](int i) {
        const hipError_t e = hipSetDevice(device_ids[i]);
        if (e != hipSuccess) {
            rcs[i] = fail(FA_ERR_HIP, "hipSetDevice(%d): %s", device_ids[i], hipGetErrorString(e));
        } else {
            const fa::FwdParams p = make_params(q[i], k[i], v[i], o[i], lse ? lse[i] : nullptr, bh[i], n, d, scale);
            hipStream_t st = streams ? static_cast<hipStream_t>(streams[i]) : nullptr;
            if (workspaces != nullptr) rcs[i] = launch(p, d, causal, dtype, kernel, st, workspaces[i], workspace_bytes[i], true);
            else rcs[i] = launch(p, d, causal, dtype, kernel, st);
        }
        if (rcs[i] != FA_OK) msgs[i] = g_err;
    };
    std::vector<int> active;
    for (int i = 0; i < n_shards; ++i)
        if (bh[i] > 0) active.push_back(i);   // bh[i] == 0: more devices than slabs, this shard is empty
    if (active.size() == 1) work(active[0]);
    else if (active.size() > 1) run_on_shard_threads(active, work);
    t_last_chain = 0;
    (void)hipSetDevice(prev);
    for (int i = 0; i < n_shards; ++i)
        if (rcs[i] != FA_OK) return fail(rcs[i], "shard %d (device %d): %s", i, device_ids[i], msgs[i].c_str());
    return FA_OK;
}

int fa_forward_sharded(int32_t n_shards, const int32_t* device_ids, const void* const* q, const void* const* k, const void* const* v,
                       void* const* o, const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                       void* const* streams)
{
    return fa_forward_sharded_ex(n_shards, device_ids, q, k, v, o, nullptr, bh, n, d, scale, causal, dtype, FA_KERNEL_AUTO, nullptr, nullptr, streams);
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

// The blocking measurement entry points own a workspace for the duration of the measurement (hipMalloc / hipFree around the timed
// region, never inside it) and launch through the fa_forward_ws path: what is timed is the C ABI proper, and the launch chains that
// need scratch are legal inside the captured graph of fa_time_forward_graph.
static int time_forward_impl(const void* q, const void* k, const void* v, void* o, int64_t bh, int64_t n, int32_t d, float scale,
                             int32_t causal, int32_t dtype, int32_t kernel, void* stream, int32_t warmup, int32_t iters,
                             float* ms_per_forward, bool graph_replay)
{
    g_err[0] = 0;
    if (!ms_per_forward || iters < 1 || warmup < 0) return fail(FA_ERR_INVALID_ARGUMENT, "bad timing arguments");
    if (int rc = validate_common(q, k, v, o, bh, n, d, scale, dtype)) return rc;
    const fa::FwdParams p = make_params(q, k, v, o, nullptr, bh, n, d, scale);
    const Plan pl = make_plan(p, d, causal, dtype, kernel, true);
    if (pl.status != FA_OK) return pl.status;
    void* ws = nullptr;
    if (pl.total > 0 && hipMalloc(&ws, pl.total) != hipSuccess) return fail(FA_ERR_HIP, "hipMalloc(%zu) for the measurement's workspace failed", pl.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
        if (e0) (void)hipEventDestroy(e0);
        if (ws) (void)hipFree(ws);
        return fail(FA_ERR_HIP, "hipEventCreate failed");
    }
    int rc = FA_OK;
    for (int i = 0; i < warmup && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, s, ws, pl.total, true);
    if (rc == FA_OK && graph_replay) {  // the `iters` launches captured into one hipGraph; three replays timed one by one, the median reported
        hipStream_t cs = nullptr;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        if (hipStreamSynchronize(s) != hipSuccess) rc = fail(FA_ERR_HIP, "hipStreamSynchronize failed");
        if (rc == FA_OK && hipStreamCreate(&cs) != hipSuccess) rc = fail(FA_ERR_HIP, "hipStreamCreate failed");
        if (rc == FA_OK && hipStreamBeginCapture(cs, hipStreamCaptureModeGlobal) != hipSuccess) rc = fail(FA_ERR_HIP, "begin capture failed");
        for (int i = 0; i < iters && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, cs, ws, pl.total, true);
        if (rc == FA_OK && hipStreamEndCapture(cs, &graph) != hipSuccess) rc = fail(FA_ERR_HIP, "end capture failed");
        if (rc == FA_OK && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(FA_ERR_HIP, "graph instantiate failed");
        if (rc == FA_OK) {
            (void)hipGraphLaunch(exec, cs);   // warm replay
            (void)hipStreamSynchronize(cs);
            float t[3] = {0.0f, 0.0f, 0.0f};
            for (int r = 0; r < 3 && rc == FA_OK; ++r) {
                (void)hipEventRecord(e0, cs);
                (void)hipGraphLaunch(exec, cs);
                (void)hipEventRecord(e1, cs);
                const hipError_t e = hipEventSynchronize(e1);
                if (e != hipSuccess) rc = fail(FA_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
                else (void)hipEventElapsedTime(&t[r], e0, e1);
            }
            if (rc == FA_OK) {
                const float lo = fminf(fminf(t[0], t[1]), t[2]), hi = fmaxf(fmaxf(t[0], t[1]), t[2]);
                *ms_per_forward = (t[0] + t[1] + t[2] - lo - hi) / (float)iters;
            }
        }
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        if (cs) (void)hipStreamDestroy(cs);
    } else if (rc == FA_OK) {
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < iters && rc == FA_OK; ++i) rc = launch(p, d, causal, dtype, kernel, s, ws, pl.total, true);
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
    if (ws) {
        (void)hipStreamSynchronize(s);
        // the chain's verdict word lives in the workspace: read it for fa_last_forward_route before the buffer goes away
        if (rc == FA_OK && t_last_chain != 0 && t_last_flag.word != nullptr) {
            uint32_t word = 0;
            if (hipMemcpy(&word, t_last_flag.word, sizeof(word), hipMemcpyDeviceToHost) == hipSuccess) t_last_route = word == t_last_flag.serial ? 2 : 1;
        }
        (void)hipFree(ws);
    }
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
    if (t_last_route >= 0) {
        *route = t_last_route;
        return FA_OK;
    }
    uint32_t word = 0;
    hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = hipMemcpy(&word, t_last_flag.word, sizeof(word), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(FA_ERR_HIP, "reading the chain's flag word failed: %s", hipGetErrorString(e));
    *route = word == t_last_flag.serial ? 2 : 1;
    return FA_OK;
}

int fa_get_stats(fa_stats* out)
{
    if (!out) return fail(FA_ERR_INVALID_ARGUMENT, "null stats pointer");
    memset(out, 0, sizeof(*out));
    out->forwards = g_stats.forwards.load(std::memory_order_relaxed);
    out->chains = g_stats.chains.load(std::memory_order_relaxed);
    out->chains_degraded = g_stats.chains_degraded.load(std::memory_order_relaxed);
    out->scratch_replans = g_stats.scratch_replans.load(std::memory_order_relaxed);
    out->slot_evictions = g_stats.slot_evictions.load(std::memory_order_relaxed);
    out->capture_slots_recycled = g_stats.capture_slots_recycled.load(std::memory_order_relaxed);
    for (int dev = 0; dev < kMaxDevices; ++dev) {
        std::lock_guard<std::mutex> g(g_slots[dev].mu);
        out->eager_slots_in_use += (uint64_t)g_slots[dev].eager.size();
        out->capture_slots_in_use += (uint64_t)(g_slots[dev].next_capture - (int)g_slots[dev].free_capture.size());
    }
    out->eager_slots_per_device = kEagerSlots;
    out->capture_slots_per_device = kFlagSlots - kEagerSlots;
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
    return "flashattn_amd abi 4 gfx950 (hip, mfma f32 32x32x2 / bf16 32x32x16, lds-dma) +ablation";
#else
    return "flashattn_amd abi 4 gfx950 (hip, mfma f32 32x32x2 / bf16 32x32x16, lds-dma)";
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
        const fa::FwdParams pk = make_params(nullptr, nullptr, nullptr, nullptr, nullptr, bh, n, d, 1.0f);
        if (keysplit_factor(pk, d, causal) > 1) return "fa_fwd_bf16_x2_kernel";       // small grids: key-split launch of the NB = 2 kernel
        return fa::bf16_kernel_name(bh, n, d, causal);
    }
    if (dtype == FA_DTYPE_BF16_OUT_F32) {   // the accurate P (see fa_dtype): hi + lo bf16 terms in the one-wave-per-SIMD kernel (slabs below 4 GiB)
        if (((n - 1) * d + d) * 2 >= 0xffffffffLL) return "fa_fwd_f32_split_kernel";
        const fa::FwdParams pk = make_params(nullptr, nullptr, nullptr, nullptr, nullptr, bh, n, d, 1.0f);
        if (keysplit_factor(pk, d, causal, false, true) > 1) return "fa_fwd_bf16_x2_pb2_kernel";   // small grids: key-split launch of the NB = 2 kernel
        return d == 64 && fa::bf16_pb2_uses_x4(bh, n, causal) ? "fa_fwd_bf16_x4_pb2_kernel" : "fa_fwd_bf16_x2_pb2_kernel";
    }
    return nullptr;
}

const char* fa_kernel_name(int32_t dtype, int32_t d, int32_t causal) { return fa_kernel_name_for(dtype, d, causal, 16, 8192); }

}  // extern "C"
