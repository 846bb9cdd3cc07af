// fa_selftest.cpp -- device-free self-test of the host logic, compiled into libflashattn_amd_asan.so only (build.py --sanitize: every
// host translation unit under -fsanitize=address,undefined, -DFA_HOST_TEST=1) (fa_host.h).
#include "fa_host.h"

#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

using namespace fa_host;

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

