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
// total; head dims inside and outside the instantiated sets), the key-split arithmetic and the no-device validation paths.
// Returns 0, or the number of the first failed check.
extern "C" int fa_host_selftest(void)
{
    int check = 0;
#define FA_CHECK(cond) do { ++check; if (!(cond)) { fprintf(stderr, "fa_host_selftest: line %d: %s\n", __LINE__, #cond); return check; } } while (0)
    // ---- plans
    const int dtypes[] = {FA_DTYPE_F32, FA_DTYPE_BF16, FA_DTYPE_BF16_OUT_F32};
    const int kinds[] = {FA_KERNEL_AUTO, FA_KERNEL_MFMA, FA_KERNEL_SPLIT, FA_KERNEL_PB2, FA_KERNEL_NAIVE, kKernelP16x2, 9};
    const int64_t bhs[] = {1, 2, 3, 8, 16, 33, 128, 1024, 70000}, ns[] = {1, 31, 300, 1024, 1100, 2048, 4095, 4096, 5000, 8192, 16384, 40000, 1 << 24};
    for (int dt : dtypes) for (int kind : kinds) for (int d : {32, 64, 128, 48, 96, 256, 300}) for (int causal : {0, 1}) for (int64_t bh : bhs) for (int64_t n : ns) {
        const fa::FwdParams p = make_params(nullptr, nullptr, nullptr, nullptr, nullptr, bh, n, d, 1.0f);
        for (bool scratch_ok : {false, true}) {
            const Plan pl = make_plan(p, d, causal, dt, kind, scratch_ok);
            if (kind == FA_KERNEL_AUTO) FA_CHECK((pl.status == FA_OK) == (d <= 256));          // AUTO takes every head dim up to 256
            if (kind == FA_KERNEL_AUTO && d == 48) FA_CHECK(pl.route == kRouteNaive);
            if (kind == FA_KERNEL_AUTO && (d == 96 || d == 256)) FA_CHECK(pl.route == kRouteF32Exact);
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
    // ---- argument validation without a device
    {
        alignas(256) static char buf[4][8192];
        FA_CHECK(fa_forward(buf[0], buf[1], buf[2], buf[3], 1, 4, 300, 1.0f, 0, FA_DTYPE_F32, nullptr) == FA_ERR_UNSUPPORTED);   // head dim > 256
        FA_CHECK(fa_forward_ex(buf[0], buf[1], buf[2], buf[3], nullptr, 1, 32, 48, 1.0f, 0, FA_DTYPE_F32, FA_KERNEL_SPLIT, nullptr) == FA_ERR_UNSUPPORTED);
        FA_CHECK(fa_forward(buf[0], buf[1], buf[2], buf[0], 1, 32, 64, 1.0f, 0, FA_DTYPE_F32, nullptr) == FA_ERR_INVALID_ARGUMENT);
        FA_CHECK(fa_forward_ws(buf[0], buf[1], buf[2], buf[3], nullptr, 1, 8, 64, 1.0f, 0, FA_DTYPE_F32, FA_KERNEL_AUTO, buf[3], 4096, nullptr) == FA_ERR_INVALID_ARGUMENT);
        fa_stats st;
        FA_CHECK(fa_get_stats(&st, sizeof(st)) == FA_OK && st.struct_bytes == sizeof(st));
        uint64_t first_field = 0;
        FA_CHECK(fa_get_stats(reinterpret_cast<fa_stats*>(&first_field), sizeof(first_field)) == FA_OK && first_field == sizeof(fa_stats));   // a shorter caller gets its prefix
        FA_CHECK(fa_get_stats(nullptr, sizeof(st)) == FA_ERR_INVALID_ARGUMENT && fa_get_stats(&st, 4) == FA_ERR_INVALID_ARGUMENT);
    }
#undef FA_CHECK
    return 0;
}
#endif

