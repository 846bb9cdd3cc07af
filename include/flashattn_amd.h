/*
 * include/flashattn_amd.h -- C ABI of the MI355X-native fused flash-attention forward.
 *
 * This is the drop-in boundary for the one hot path of kilianhae/FlashAttention.C:
 *
 *     torch::Tensor forward(torch::Tensor Q, torch::Tensor K, torch::Tensor V, bool causal)
 *         declared  /root/reference/src/main.cpp:3, bound to Python at :5-6
 *         defined   /root/reference/src/flashattention.cu:603-617
 *         launchers /root/reference/src/flashattention.cu:590-602 (run_flash_tiled_coarse{,_causal})
 *
 * Everything here is plain C: raw device pointers, sizes, an opaque HIP stream.  No torch types.
 * The reference-side bindings a maintainer would add (pybind TU for main.cpp, ctypes for
 * bench_flashattention.py, a direct call for test.cu) are shown in INTEGRATION.md.
 *
 * Semantics shared by every entry point
 *   tensors     (BH, N, d) row-major contiguous, batch and head pre-flattened, exactly as the reference
 *               addresses them: element (b, r, c) at b*N*d + r*d + c  (flashattention.cu:144,198,224,350)
 *   math        O = softmax(scale * Q K^T  [causal: key index <= query index]) V, per (batch*head)
 *   scale       explicit; the reference hard-wires 1.0 (flashattention.cu:593,600) -- pass 1.0f for parity
 *   ragged N    any N >= 1 is handled exactly (tail keys are masked, not zero-filled as at
 *               flashattention.cu:224-231)
 *   ownership   the caller owns every buffer; nothing is allocated, freed or zero-filled by fa_forward_ws, the entry point of this
 *               boundary proper (the reference allocates O and a dead O_l inside forward(), :608-609).  Scratch is needed only for
 *               key-split launches (rows of 4096 keys and more -- 2048 for FA_KERNEL_MFMA on fp32 tensors -- on grids that leave the chip idle: the partial outputs of the key
 *               shares) and, 256 bytes of it, for the report word of an fp32 FA_KERNEL_AUTO forward (fa_last_forward_route):
 *               fa_workspace_bytes() sizes it, fa_forward_ws() takes it.  The convenience entry points (fa_forward, fa_forward_ex, fa_forward_sharded,
 *               fa_forward_packed_qkv) draw the key-split scratch from a PRIVATE stream-ordered pool of the device
 *               (hipMemPoolCreate; hipMallocFromPoolAsync / hipFreeAsync on `stream`; the device's default pool is never touched) --
 *               except while `stream` is capturing (graph allocations proved unreliable on ROCm 7.2): the launch then runs unsplit.
 *               A failed pool allocation has the same effect.  Their report words live in a per-device slot table (see fa_get_stats)
 *   aliasing    o must not overlap q, k or v (a tile that fails its verification is recomputed from q, k, v after o was
 *               written): overlapping ranges are rejected with FA_ERR_INVALID_ARGUMENT
 *   ordering    the kernel is enqueued on `stream` and the call returns without synchronising
 *               (the reference launches on the legacy stream and calls cudaDeviceSynchronize, :593-594)
 *   errors      every entry point returns FA_OK (0) or an fa_status code; fa_last_error() returns a
 *               thread-local message.  Nothing asserts or exits (the reference asserts on d, :606).
 *   threads     stateless and re-entrant; safe from any host thread for any (device, stream)
 */
#ifndef FLASHATTN_AMD_H
#define FLASHATTN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLASHATTN_AMD_ABI_VERSION 5   /* 5: fa_stats gained tiles_redone, workgroups_fp32 (appended: a caller built against 4 must re-build) */

typedef enum fa_status {
    FA_OK = 0,
    FA_ERR_INVALID_ARGUMENT = 1, /* null pointer, non-positive size, misaligned buffer              */
    FA_ERR_UNSUPPORTED = 2,      /* head dim / dtype / kernel id not instantiated                    */
    FA_ERR_HIP = 3,              /* a HIP runtime call or the launch itself failed                   */
    FA_ERR_NO_DEVICE = 4         /* no gfx950 device visible                                         */
} fa_status;

typedef enum fa_dtype {
    FA_DTYPE_F32 = 0, /* fp32 in, fp32 out -- the reference's dtype.  FA_KERNEL_AUTO (round 5): Q.K^T as three matrix products of two-term
                         FP16 splits of the fp32 operands (hi = f16(x), lo = f16(x - hi): 22 significant bits; v_mfma_f32_32x32x16_f16,
                         fp32 accumulate, the hi.hi products of all k-steps first so that the cross terms are added where the partial sum
                         is small), on keys CENTRED on a reference key (k_j - kbar, kbar = the coordinate-wise median of three keys of
                         the share: softmax only needs differences, and a magnitude all keys share then never enters a rounded sum; the
                         row constant q.kbar goes back into the LSE); P.V as three products of two-term BF16 splits (P needs fp32's
                         exponent range) of values centred the same way (v_j - vbar, vbar added back to O: the 16 bits cover the spread
                         of V, not an offset all values share); 3x faster than fp32 arithmetic.  GUARANTEED: the logit error of the operand terms is
                         <= 3 * 2^-22 * sum |q_i (k_i - kbar_i)| * scale -- below the rounding bound d * 2^-24 * sum |q_i k_i| of the
                         reference's own fp32 FMA chain for every d >= 12 -- and the P.V terms add <= 3 * 2^-17 * max|v - vbar|.  CONTRACT
                         (tests/test_gpu_adversarial.py): |O - O_fp64| and |LSE - LSE_fp64| are <= max(1e-3, E_ref) on every input, E_ref =
                         what the reference's own arithmetic (a k-ordered chain of rounding fp32 FMAs, flashattention.cu:236-252) leaves
                         on that input.  OBSERVED: <= 1e-4 on unit-variance data at scale 1 (c2, c3; FA_KERNEL_MFMA reads 2e-5 there),
                         <= 1.5e-5 at 1/sqrt(d); <= 2.7e-4 (O and LSE) on coherent inputs (constant-component rows, v = +-5, logits ~1500)
                         where FA_KERNEL_MFMA reads 8e-4 .. 5.9e-3.
                         RANGE GUARD: fp16 terms hold |x| < 65520 and lose elements below 2^-3 to subnormal lo terms (absolute error 2^-25
                         each); a workgroup whose first attempt produced a NaN, or whose D * max|k - kbar| + sqrt(D) * max|q * scale * log2 e|_2
                         exceeds 8192 (the subnormal terms could then add more than 2^-12 to a logit in the worst case; unit-variance data: ~600 at d = 64,
                         ~1000 at d = 128; key elements up to 60 at d = 128 still pass), redoes its rows in exact fp32 arithmetic before it exits -- ONE launch, no host round trip;
                         fa_last_forward_route() tells whether any workgroup did.  (Rounds 1-4 used two BF16 terms, 16 bits, behind a
                         logit-width guard that bounded an RMS error model: coherent inputs under its limit read up to 6e-2.)
                         FA_KERNEL_SPLIT: the same products without the guard.  FA_KERNEL_MFMA: fp32 arithmetic (v_mfma_f32_32x32x2_f32),
                         bit-for-bit an fmaf chain -- the reference's arithmetic, NOT an oracle (see E_ref above); causal launches pair a
                         heavy with a light tile per workgroup, idle grids run over key shares.  FA_F32_AUTO=exact in the environment makes
                         that the FA_KERNEL_AUTO choice for the whole process */
    FA_DTYPE_BF16 = 1,        /* bf16 in, bf16 MFMA with fp32 accumulate and fp32 softmax, bf16 out     */
    FA_DTYPE_BF16_OUT_F32 = 2 /* bf16 in, O written as fp32 (the accumulator precision).  Under FA_KERNEL_AUTO this also selects
                                 the ACCURATE P: a caller who wants the fp32 accumulator gets P as two bf16 terms, hi + lo
                                 (FA_KERNEL_PB2: ~17 significant bits, Q.K^T in the fp32 accumulator: asserted at 2e-4 of fp64 on every data
                                 family of the soak -- observed 1.7e-5 on B=2 H=8 d=64 N=8192, <= 4e-5 on the BASELINE configs, 1.2e-4 on
                                 coherent wide-logit inputs, where the accumulator itself rounds at the logit's magnitude -- where bf16 P reads
                                 ~8e-3 in the accumulator and ~1.5e-2 after the bf16 output's own rounding) in ONE launch without
                                 scratch, at every launch size and layout.  Only slabs beyond 4 GiB take P and the scaled Q as hi + lo
                                 bf16 terms in the split kernel instead (FA_KERNEL_SPLIT: 1 .. 2e-4 on unit-variance data, growing with
                                 the logit width).  A bf16 output rounds at 2^-9 |O| by itself and keeps the fastest kernels (bf16 P). */
} fa_dtype;

typedef enum fa_kernel {
    FA_KERNEL_AUTO = 0,  /* the documented choice per dtype (see fa_dtype)                              */
    FA_KERNEL_NAIVE = 1, /* rung-0 scalar kernel: fp32 only, any d <= 256; on-device cross-check        */
    FA_KERNEL_MFMA = 2,  /* the tiled MFMA kernel in the arithmetic of `dtype`: exact fp32 for fp32 tensors, bf16 P for bf16
                            tensors (whatever the output type); d in {32, 64, 128}                        */
    FA_KERNEL_SPLIT = 3, /* split products on the 16-bit matrix pipe.  fp32 tensors: see FA_DTYPE_F32 (the same products, unguarded).
                            bf16 tensors: K, V exact in one term, Q*scale*log2e and P carried as hi + lo (two products per
                            contraction): max-abs error ~1e-4 against fp64 at scale 1 with FA_DTYPE_BF16_OUT_F32, at ~2x the
                            time of the bf16-P kernels */
    FA_KERNEL_P16 = 4,   /* (libflashattn_amd_ablation.so only since ABI 4; FA_ERR_UNSUPPORTED in the product library)  P and V in fp16,
                            ONE fp16 term of P (11 significant bits): 8e-4 .. 1.2e-3 at scale 1 -- AT the 1e-3 bar, not inside it. */
    FA_KERNEL_P16X2 = 5, /* (ablation library only since ABI 4)  P as fp16 hi + fp16 lo, V copied to fp16 in scratch, split kernel as the
                            device-side fallback when some |v| >= 2^16: round 3's accurate path, a chain of three launches. */
    FA_KERNEL_PB2 = 6    /* bf16 tensors: P as bf16 hi + bf16 lo (lo = bf16(p - hi), the exact difference from one v_dot2c_f32_bf16 per
                            element; twice the P.V and row-sum MFMAs; P to ~2^-17), V as it is: ONE launch, no copy of V, no scratch
                            except for key-split launches of idle grids, any |v|, any layout with slabs below 4 GiB; the optimistic
                            softmax of the bf16-P kernels with its rescaled redo.  ~1.5x the time of the bf16-P kernels.  The
                            FA_KERNEL_AUTO choice for FA_DTYPE_BF16_OUT_F32. */
} fa_kernel;
/* `kernel` arguments: bits 0..7 = fa_kernel; bits 8..15 = 0, or the number of one of the co-compiled tilings of that family
 * (every one of them computes the same function; csrc/fa_fwd_bf16.hip and csrc/fa_split_kernel.h list them, tests/ run them
 * all).  Numbers that are not shipped tilings are rejected with FA_ERR_UNSUPPORTED. */

/*
 * fa_forward -- replaces forward() / run_flash_tiled_coarse{,_causal}
 *               (/root/reference/src/flashattention.cu:590-617).
 *   q, k, v   device pointers, (bh, n, d) elements of `dtype`, 16-byte aligned
 *   o         device pointer, (bh, n, d) elements of `dtype` (fp32 for FA_DTYPE_BF16_OUT_F32); every element is written
 *   d         head dim: 32, 64 or 128 (the reference compiles exactly one, `#define d 64`, :15)
 *   causal    0 / non-zero: the `bool causal` of the reference signature
 *   stream    hipStream_t (NULL = the null stream)
 */
int fa_forward(const void* q, const void* k, const void* v, void* o,
               int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
               int32_t dtype, void* stream);

/*
 * fa_forward_ex -- fa_forward plus the row log-sum-exp the reference reserves `O_l` for
 *                  (/root/reference/src/flashattention.cu:609; filled only by the "lightning" kernels,
 *                  flashattention_lightning.cu:124,234) and an explicit kernel choice.
 *   lse       NULL, or device pointer to (bh, n) fp32: lse[b, r] = log(sum_c exp(scale * q_r . k_c))
 *   kernel    an fa_kernel value
 */
int fa_forward_ex(const void* q, const void* k, const void* v, void* o, float* lse,
                  int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
                  int32_t dtype, int32_t kernel, void* stream);

/*
 * fa_workspace_bytes -- bytes of caller-owned scratch THIS call needs (0 = none; also 0 for arguments fa_forward_ws would reject).
 *                       Same (bh, n, d, causal, dtype, kernel) as the forward it sizes; an upper bound of what is touched.
 * fa_forward_ws      -- fa_forward_ex that never allocates: the non-allocating form of the boundary
 *                       (ownership as in /root/reference/src/flashattention.cu:608-609 inverted: caller owns all buffers).
 *   workspace        device pointer, 256-byte aligned, at least fa_workspace_bytes() bytes, not overlapping q, k, v, o; NULL is fine
 *                    when the call needs none.  In use until the forward has completed on `stream`; one forward at a time per
 *                    workspace (its first bytes hold the forward's report word).  Contents need no initialisation.
 *                    NULL with FA_KERNEL_AUTO when the call would use one: the forward runs without scratch (unsplit launch; the
 *                    report word from the slot table) instead of failing -- a binder that skips fa_workspace_bytes() works.
 *   capture          legal while `stream` is capturing, with every kernel family (an fp32 FA_KERNEL_AUTO forward clears its report
 *                    word with a memset node, so replays of the graph report independently -- with or without a workspace).
 */
size_t fa_workspace_bytes(int64_t bh, int64_t n, int32_t d, int32_t causal, int32_t dtype, int32_t kernel);
int fa_forward_ws(const void* q, const void* k, const void* v, void* o, float* lse,
                  int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
                  int32_t dtype, int32_t kernel, void* workspace, size_t workspace_bytes, void* stream);

/*
 * fa_forward_sharded -- the batch*head axis split across several devices of one node, no collective
 *                       (every blockIdx.x of the reference grid is independent: flashattention.cu:144).
 *   n_shards        number of shards
 *   device_ids[i]   HIP device ordinal of shard i (buffers of shard i live there); the devices of non-empty shards must be distinct
 *                   (FA_ALLOW_SAME_DEVICE=1 in the environment lifts the check: single-GPU test boxes)
 *   q/k/v/o[i]      device pointers of shard i, (bh[i], n, d)
 *   streams[i]      hipStream_t on device_ids[i] (NULL entries / NULL array = null stream)
 * Each shard is enqueued by its own (persistent) host thread (a forward can be several launches and a pool allocation: one thread would start the last
 * device a whole table's worth of host time behind the first) without synchronising; the caller's current device is untouched.
 */
int fa_forward_sharded(int32_t n_shards, const int32_t* device_ids,
                       const void* const* q, const void* const* k, const void* const* v, void* const* o,
                       const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal,
                       int32_t dtype, void* const* streams);
/*
 * fa_forward_sharded_ex -- the same with what fa_forward_ws has per shard: lse[i] (NULL array or NULL entries: none), an explicit
 *                          `kernel`, and caller-owned scratch -- workspaces[i] / workspace_bytes[i], each at least
 *                          fa_workspace_bytes(bh[i], n, d, causal, dtype, kernel) (both arrays NULL: scratch from each device's private
 *                          pool, as fa_forward_sharded).  With workspaces nothing is allocated and the call is legal while the shards'
 *                          streams are capturing.  The worker threads are persistent (created on first use, one per shard index).
 */
int fa_forward_sharded_ex(int32_t n_shards, const int32_t* device_ids,
                          const void* const* q, const void* const* k, const void* const* v, void* const* o, float* const* lse,
                          const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal,
                          int32_t dtype, int32_t kernel, void* const* workspaces, const size_t* workspace_bytes,
                          void* const* streams);

/*
 * fa_forward_packed_qkv -- llm.c layout entry, replaces attention_forward6
 *                          (/root/reference/src/llm.c/attention_forward.cu:1106-1179): causal, scale
 *                          1/sqrt(C/NH), fp32.  Reads the packed (B, T, 3C) activations directly and writes
 *                          (B, T, C) -- the reference's permute_kernel / unpermute_kernel (:519-565) and
 *                          their three temporaries are fused away.
 */
int fa_forward_packed_qkv(const float* inp, float* out, int32_t B, int32_t T, int32_t C, int32_t NH,
                          void* stream);

/*
 * fa_time_forward -- enqueue `warmup` + `iters` forwards on `stream`, bracket the `iters` timed ones with
 *                    HIP events recorded on that same stream, and return the mean milliseconds per forward.
 *                    Blocking (the counterpart of benchmark_kernel, /root/reference/src/llm.c/common.h:108-124; used by the C
 *                    driver and bench.py's roofline leg).  The launches go through the fa_forward_ws path with a workspace the
 *                    measurement owns (hipMalloc / hipFree outside the timed region).
 */
int fa_time_forward(const void* q, const void* k, const void* v, void* o,
                    int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
                    int32_t dtype, int32_t kernel, void* stream,
                    int32_t warmup, int32_t iters, float* ms_per_forward);

/*
 * fa_time_forward_graph -- the same measurement with the `iters` launches captured into one hipGraph on a private
 *                          stream; after a warm replay three replays are timed one by one and the median is reported.
 *                          Reported beside the stream-launch figure, never instead of it.  (On ROCm 7.2 a graph replay of
 *                          back-to-back forwards is NOT faster than the same launches on a stream: see DESIGN.md section 7.)
 */
int fa_time_forward_graph(const void* q, const void* k, const void* v, void* o,
                          int64_t bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                          int32_t kernel, int32_t warmup, int32_t iters, float* ms_per_forward);

/*
 * fa_last_forward_route -- which arithmetic produced the output of this thread's most recent forward.  Blocking (waits for
 *                          `stream`, reads one word back): diagnostics and benchmarks only.
 *   *route  0 = nothing to report (every bf16 path; explicit kernels; an fp32 forward that found no report word);  1 = fp32 tensors under
 *           FA_KERNEL_AUTO: split products throughout;  2 = the range guard fired (operands outside what fp16 terms hold, or a NaN): at
 *           least one workgroup redid its rows in exact fp32 arithmetic (inside the same launch)
 */
int fa_last_forward_route(void* stream, int32_t* route);

/*
 * fa_get_stats -- process-wide counters of the launch machinery (never fails for a non-NULL pointer; cheap; no device access).
 * The report word of an fp32 FA_KERNEL_AUTO forward that has no caller-owned workspace comes from a per-device table: one slot per
 * (device, stream) for eager calls -- when all eager_slots_per_device are taken, the least recently used slot whose last forward has
 * completed changes hands (slot_evictions) --, one slot per captured forward, returned when the graph and its executables are destroyed
 * (capture_slots_recycled; on a runtime that refuses the user-object hook a capture slot is used once, and capture_slots_per_device
 * captures without a workspace exhaust them).  chains_degraded counts the forwards that found no slot: since round 4 they run exactly
 * as fast and as accurately as the others (the fallback is inside the kernel) and only report route 0; in the ablation library, whose
 * chains of launches still depend on the word, such a call runs the chain's always-correct kernel alone.
 */
typedef struct fa_stats {
    uint64_t forwards;                 /* forwards enqueued through any entry point                          */
    uint64_t chains;                   /* ... of which forwards with a report word (fp32 FA_KERNEL_AUTO)     */
    uint64_t chains_degraded;          /* ... that found no slot for the word (see above)                    */
    uint64_t scratch_replans;          /* forwards re-planned without scratch (NULL workspace / pool failure) */
    uint64_t slot_evictions;           /* eager slots that changed hands                                     */
    uint64_t capture_slots_recycled;   /* capture slots returned by destroyed graphs                         */
    uint64_t eager_slots_in_use, capture_slots_in_use, eager_slots_per_device, capture_slots_per_device;
    /* ABI 5: the two performance cliffs of correct-but-slower paths, counted by the kernels themselves (device-scope atomics into two words
       of the GPU's memory, on the rare path only; fa_get_stats() sums them over the devices this process has launched on with one blocking
       16-byte copy per device -- it synchronises with the device like fa_last_forward_route()):
       tiles_redone     workgroup tiles whose optimistic attempt failed its range check and were recomputed with the rescaled / textbook
                        softmax (about 2x the tile's time): exponent references outgrown by 2^100, values below ~2^-30 (an all-zero V --
                        fp32 tensors: any V that is exactly constant over the share -- is recognised by a look at V and costs no redo);
       workgroups_fp32  workgroups of an fp32 FA_KERNEL_AUTO forward that redid their rows in fp32 arithmetic (about 3x): operands outside
                        what fp16 terms hold -- the events fa_last_forward_route() == 2 reports per forward.
       Read them before and after a call (synchronise the stream in between) to see whether it ran into either. */
    uint64_t tiles_redone, workgroups_fp32;
} fa_stats;
int fa_get_stats(fa_stats* out);

/* Thread-local description of the last failure on this thread ("" if none). */
const char* fa_last_error(void);

/* Number of visible HIP devices (0 on a machine without one; never fails). */
int fa_device_count(void);

/* "flashattn_amd <abi> gfx950 ..." build string. */
const char* fa_version(void);

/* Name of the kernel FA_KERNEL_AUTO resolves to for (dtype, d), or NULL if unsupported. */
const char* fa_kernel_name(int32_t dtype, int32_t d, int32_t causal);   /* at the headline shape bh = 16, n = 8192 */
/* the dispatch is shape dependent (tile sizes follow the grid): the kernel fa_forward picks for this very call */
const char* fa_kernel_name_for(int32_t dtype, int32_t d, int32_t causal, int64_t bh, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* FLASHATTN_AMD_H */
