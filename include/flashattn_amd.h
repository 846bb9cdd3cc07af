/*
 * include/flashattn_amd.h -- C ABI of the MI355X-native fused flash-attention forward (ABI 6, frozen).
 *
 * The drop-in boundary for the one hot path of kilianhae/FlashAttention.C:
 *
 *     torch::Tensor forward(torch::Tensor Q, torch::Tensor K, torch::Tensor V, bool causal)
 *         declared  /root/reference/src/main.cpp:3, bound to Python at :5-6
 *         defined   /root/reference/src/flashattention.cu:603-617
 *         launchers /root/reference/src/flashattention.cu:590-602 (run_flash_tiled_coarse{,_causal})
 *
 * Plain C: raw device pointers, sizes, an opaque HIP stream; no torch types.  The reference-side bindings
 * (pybind TU for main.cpp, ctypes for bench_flashattention.py, a direct call for test.cu) are in INTEGRATION.md;
 * design notes and measurements in DESIGN.md.
 *
 * 1. Semantics shared by every entry point
 *   tensors    (BH, N, d) row-major contiguous, batch and head pre-flattened, as the reference addresses them:
 *              element (b, r, c) at b*N*d + r*d + c  (flashattention.cu:144,198,224,350)
 *   math       O = softmax(scale * Q K^T  [causal: key index <= query index]) V, per (batch*head)
 *   scale      explicit; the reference hard-wires 1.0 (flashattention.cu:593,600) -- pass 1.0f for parity
 *   N          any N >= 1, exactly (tail keys are masked; the reference zero-fills, flashattention.cu:224-231)
 *   d          32, 64, 128: every kernel family.  The other multiples of 32 up to 256 (the head dims the reference
 *              compiles by editing `#define d`, flashattention.cu:15,164): the exact fp32 MFMA kernel, for fp32
 *              tensors and (widened on load, fp32 arithmetic) for bf16 tensors.  Any other d <= 256:
 *              FA_KERNEL_AUTO runs the rung-0 kernel (fp32 arithmetic; correct, slow).  fa_kernel_name_for()
 *              names what runs.
 *   ownership  the caller owns every buffer.  fa_forward_ws, the boundary proper, allocates nothing: scratch is
 *              needed only by key-split launches (long rows on grids that leave the chip idle);
 *              fa_workspace_bytes() sizes it (0 for every other call).  The convenience entries (fa_forward,
 *              fa_forward_ex, fa_forward_sharded, fa_forward_packed_qkv) take the same bytes from a PRIVATE
 *              stream-ordered pool per device (hipMemPoolCreate; the device's default pool is never touched) --
 *              except while `stream` is capturing, or when the pool fails: the launch then runs unsplit.
 *              (The reference allocates O and a dead O_l inside forward(), flashattention.cu:608-609.)
 *   aliasing   o must not overlap q, k or v (a tile that fails its range check is recomputed from q, k, v after o
 *              was written); overlap is rejected with FA_ERR_INVALID_ARGUMENT
 *   ordering   kernels are enqueued on `stream`; the call returns without synchronising
 *              (the reference launches on the legacy stream and calls cudaDeviceSynchronize, :593-594)
 *   capture    every forward entry is legal while `stream` is capturing
 *   errors     every entry point returns FA_OK (0) or an fa_status; fa_last_error() holds a thread-local message.
 *              Nothing asserts or exits (the reference asserts on d, :606)
 *   threads    thread-safe from any host thread for any (device, stream).  NOT stateless: the library keeps
 *              process-wide private memory pools, persistent worker threads (fa_forward_sharded) and counters
 *
 * 2. What FA_KERNEL_AUTO guarantees, per dtype (bounds against an fp64 evaluation of the same inputs)
 *   FA_DTYPE_F32           |O - O64| and |LSE - LSE64| <= max(1e-3, E_ref) + 3 * 2^-17 * max_j |v_j - vbar|
 *                          on every input; E_ref = the error the reference's own arithmetic (a k-ordered chain of
 *                          rounding fp32 FMAs, flashattention.cu:236-252) leaves on that input, vbar = a row of V
 *                          among the keys the workgroup reads.  Unit-variance data: <= 1e-4.  Operands outside
 *                          what the 16-bit terms hold are recomputed in fp32 arithmetic inside the same launch
 *                          (fa_last_forward_route() == 2 reports it).  ~3x faster than fp32 arithmetic.
 *   FA_DTYPE_BF16          every softmax weight off by <= 2^-8 + 2^-10 relative:
 *                          |dO| <= (2^-8 + 2^-10) * max_rows sum_j w_j |v_j - O|  +  2^-8 |O| (the bf16 output);
 *                          1.5e-2 on unit-variance data at scale 1 (B=2 H=8 d=64 N=8192), 3e-4 at 1/sqrt(d)
 *   FA_DTYPE_BF16_OUT_F32  weights to 2^-17, fp32 output: <= 2e-4 on every data family tested, 2e-5 on
 *                          unit-variance data at scale 1 -- the path inside the 1e-3 of the north star; ~1.5x the
 *                          time of FA_DTYPE_BF16
 *   other head dims        exact fp32 MFMA kernel / rung-0 kernel: fp32 arithmetic, E_ref-class error (<= 1e-4);
 *                          bf16 tensors: plus the one rounding of a bf16 output, 2^-9 |O|
 */
#ifndef FLASHATTN_AMD_H
#define FLASHATTN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLASHATTN_AMD_ABI_VERSION 6 /* frozen: fa_stats carries its size, so counters are appended without a bump */

typedef enum fa_status {
    FA_OK = 0,
    FA_ERR_INVALID_ARGUMENT = 1, /* null pointer, non-positive size, misaligned or overlapping buffer      */
    FA_ERR_UNSUPPORTED = 2,      /* head dim / dtype / kernel id not available                             */
    FA_ERR_HIP = 3,              /* a HIP runtime call or the launch itself failed                         */
    FA_ERR_NO_DEVICE = 4         /* no gfx950 device visible                                               */
} fa_status;

typedef enum fa_dtype {
    FA_DTYPE_F32 = 0,         /* fp32 in, fp32 out: the reference's dtype                                  */
    FA_DTYPE_BF16 = 1,        /* bf16 in, bf16 out: the fastest kernels (bf16 P)                           */
    FA_DTYPE_BF16_OUT_F32 = 2 /* bf16 in, fp32 out; under FA_KERNEL_AUTO also the accurate P (section 2)   */
} fa_dtype;

typedef enum fa_kernel {
    FA_KERNEL_AUTO = 0,  /* the documented choice per dtype and head dim (sections 1, 2)                    */
    FA_KERNEL_NAIVE = 1, /* rung-0 kernel: one wave per query row, fp32 arithmetic, any dtype, d <= 256;
                            the on-device cross-check of the other families                                */
    FA_KERNEL_MFMA = 2,  /* the arithmetic of the tensors' dtype.  fp32 tensors: v_mfma_f32_32x32x2_f32 for both
                            contractions, bit for bit a k-ordered fmaf chain -- the reference's arithmetic, not an
                            oracle (its error IS E_ref); d a multiple of 32 up to 256.  bf16 tensors: bf16 P
                            whatever the output type; d in {32, 64, 128}                                   */
    FA_KERNEL_SPLIT = 3, /* split products on the 16-bit matrix pipe without the range guard of AUTO.  fp32 tensors:
                            the AUTO arithmetic.  bf16 tensors: Q*scale*log2e and P as hi + lo (the path for slabs
                            beyond 4 GiB); d in {32, 64, 128}                                              */
    /* 4, 5: retired (P and V in fp16, rounds 2-3); FA_ERR_UNSUPPORTED                                     */
    FA_KERNEL_PB2 = 6    /* bf16 tensors: P as bf16 hi + bf16 lo, V as it is, one launch, any |v|, slabs below
                            4 GiB; the FA_KERNEL_AUTO choice for FA_DTYPE_BF16_OUT_F32; d in {32, 64, 128}  */
} fa_kernel;
/* `kernel` arguments: bits 0..7 = fa_kernel; bits 8..15 = 0, or the number of one of the co-compiled tilings of
 * that family (all compute the same function; tests/ run them all).  Other numbers: FA_ERR_UNSUPPORTED. */

/*
 * fa_forward -- replaces forward() / run_flash_tiled_coarse{,_causal} (/root/reference/src/flashattention.cu:590-617).
 *   q, k, v   device pointers, (bh, n, d) elements of `dtype`, 16-byte aligned
 *   o         device pointer, (bh, n, d) elements (fp32 for FA_DTYPE_BF16_OUT_F32); every element is written
 *   d         head dim, 1 .. 256 (section 1; the reference compiles exactly one, `#define d 64`, :15)
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
 * fa_workspace_bytes -- bytes of caller-owned scratch THIS call can use: the partial outputs of a key-split launch,
 *                       S * bh * n * (d + 1) * 4, and 0 for every call that does not split (also 0 for arguments
 *                       fa_forward_ws would reject).  Same (bh, n, d, causal, dtype, kernel) as the forward.
 * fa_forward_ws      -- fa_forward_ex that never allocates: the non-allocating form of the boundary (ownership as
 *                       in /root/reference/src/flashattention.cu:608-609 inverted: the caller owns all buffers).
 *   workspace        device pointer, 256-byte aligned, at least fa_workspace_bytes() bytes, not overlapping q, k,
 *                    v, o; in use until the forward has completed on `stream`; contents need no initialisation.
 *                    NULL (any kernel choice): the forward runs without scratch -- the unsplit launch -- instead
 *                    of failing, so a binder that skips fa_workspace_bytes() works.
 */
size_t fa_workspace_bytes(int64_t bh, int64_t n, int32_t d, int32_t causal, int32_t dtype, int32_t kernel);
int fa_forward_ws(const void* q, const void* k, const void* v, void* o, float* lse,
                  int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
                  int32_t dtype, int32_t kernel, void* workspace, size_t workspace_bytes, void* stream);

/*
 * fa_forward_sharded -- the batch*head axis split across several devices of one node, no collective
 *                       (every blockIdx.x of the reference grid is independent: flashattention.cu:144).
 *   n_shards        number of shards
 *   device_ids[i]   HIP device ordinal of shard i (its buffers live there); devices of non-empty shards must be
 *                   distinct (FA_ALLOW_SAME_DEVICE=1 in the environment lifts the check: single-GPU test boxes)
 *   q/k/v/o[i]      device pointers of shard i, (bh[i], n, d); bh[i] == 0: an empty shard
 *   streams[i]      hipStream_t on device_ids[i] (NULL entries / NULL array = null stream)
 * Each shard is enqueued by its own persistent host thread, without synchronising; the caller's current device is
 * untouched.
 */
int fa_forward_sharded(int32_t n_shards, const int32_t* device_ids,
                       const void* const* q, const void* const* k, const void* const* v, void* const* o,
                       const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal,
                       int32_t dtype, void* const* streams);
/*
 * fa_forward_sharded_ex -- the same with what fa_forward_ws has, per shard: lse[i] (NULL array or NULL entries:
 *                          none), an explicit `kernel`, caller-owned scratch workspaces[i] / workspace_bytes[i],
 *                          each at least fa_workspace_bytes(bh[i], n, d, causal, dtype, kernel) (both arrays NULL:
 *                          each device's private pool).  With workspaces nothing is allocated.
 */
int fa_forward_sharded_ex(int32_t n_shards, const int32_t* device_ids,
                          const void* const* q, const void* const* k, const void* const* v, void* const* o,
                          float* const* lse, const int64_t* bh, int64_t n, int32_t d, float scale, int32_t causal,
                          int32_t dtype, int32_t kernel, void* const* workspaces, const size_t* workspace_bytes,
                          void* const* streams);

/*
 * fa_forward_packed_qkv -- llm.c layout entry, replaces attention_forward6
 *                          (/root/reference/src/llm.c/attention_forward.cu:1106-1179): causal, scale
 *                          1/sqrt(C/NH), fp32.  Reads the packed (B, T, 3C) activations in place and writes
 *                          (B, T, C): the reference's permute_kernel / unpermute_kernel (:519-565) and their
 *                          three temporaries are fused away.  Head size C/NH: as `d` of fa_forward.
 */
int fa_forward_packed_qkv(const float* inp, float* out, int32_t B, int32_t T, int32_t C, int32_t NH,
                          void* stream);

/*
 * fa_time_forward -- enqueue `warmup` + `iters` forwards on `stream`, bracket the timed ones with HIP events
 *                    recorded on that stream, return the mean milliseconds per forward.  Blocking (the counterpart
 *                    of benchmark_kernel, /root/reference/src/llm.c/common.h:108-124; used by the C driver and
 *                    bench.py's roofline leg).  Launches through fa_forward_ws with a workspace the measurement
 *                    owns (hipMalloc / hipFree outside the timed region).
 * fa_time_forward_graph -- the same with the `iters` launches captured into one hipGraph on a private stream; three
 *                    replays are timed, the median is reported.
 */
int fa_time_forward(const void* q, const void* k, const void* v, void* o,
                    int64_t bh, int64_t n, int32_t d, float scale, int32_t causal,
                    int32_t dtype, int32_t kernel, void* stream,
                    int32_t warmup, int32_t iters, float* ms_per_forward);
int fa_time_forward_graph(const void* q, const void* k, const void* v, void* o,
                          int64_t bh, int64_t n, int32_t d, float scale, int32_t causal, int32_t dtype,
                          int32_t kernel, int32_t warmup, int32_t iters, float* ms_per_forward);

/*
 * fa_last_forward_route -- which arithmetic produced this thread's most recent forward.  Blocking (waits for
 *                          `stream`, copies one word back): diagnostics and benchmarks only.
 *   *route  0 = nothing to report: bf16 tensors, explicit kernels, head dims outside {32, 64, 128}, a forward
 *               enqueued while its stream was capturing, fa_forward_sharded (its shards run on worker threads)
 *           1 = fp32 tensors under FA_KERNEL_AUTO: split products on the 16-bit pipes throughout
 *           2 = ... and at least one workgroup left what the 16-bit terms hold (or met a NaN) and redid its rows
 *               in fp32 arithmetic inside the same launch
 *   The word is one of a ring of 1024 per device (word = the forward's serial mod 1024; raised = equal to the serial; a
 *   forward that did not fall back writes nothing): 2 is never wrong; 1 can be stale for a forward more than 1024 fp32
 *   FA_KERNEL_AUTO forwards back.  Use fa_read_device_counters() around a captured graph or a sharded call.
 */
int fa_last_forward_route(void* stream, int32_t* route);

/*
 * fa_get_stats -- process-wide host counters.  Never blocks, touches no device.  `struct_bytes` = sizeof(fa_stats)
 *                 as the caller was compiled: at most that many bytes are written, and fields appended by a later
 *                 library never break an older caller (st.struct_bytes tells what the library knows).
 */
typedef struct fa_stats {
    uint64_t struct_bytes;    /* sizeof(fa_stats) of the library that filled this in                       */
    uint64_t forwards;        /* forwards enqueued through any entry point                                 */
    uint64_t scratch_replans; /* forwards that ran without their scratch (NULL workspace, capture, pool failure) */
} fa_stats;
int fa_get_stats(fa_stats* out, size_t struct_bytes);

/*
 * fa_read_device_counters -- the two performance cliffs of correct-but-slower paths, counted by the kernels themselves
 *                            (device-scope atomics into two words of each GPU's memory, on the rare path only),
 *                            summed over the devices this process has launched on.  BLOCKING: one 8-byte hipMemcpy
 *                            per counter and device (it synchronises with the device; do not call it while a
 *                            stream of the process is capturing in global mode).  Either pointer may be NULL.
 *   tiles_redone     workgroup tiles whose optimistic attempt failed its range check and were recomputed with the
 *                    rescaled / textbook softmax (~2x that tile): exponent references outgrown by 2^100, values
 *                    below ~2^-30 (an all-zero V -- fp32 tensors: any V constant over the share -- costs no redo)
 *   workgroups_fp32  workgroups of an fp32 FA_KERNEL_AUTO forward that redid their rows in fp32 arithmetic (~3x)
 * Read them before and after a call (synchronise in between) to see whether it ran into either.
 */
int fa_read_device_counters(uint64_t* tiles_redone, uint64_t* workgroups_fp32);

/* Thread-local description of the last failure on this thread ("" if none). */
const char* fa_last_error(void);

/* Number of visible HIP devices (0 on a machine without one; never fails). */
int fa_device_count(void);

/* "flashattn_amd abi <n> gfx950 ..." build string. */
const char* fa_version(void);

/* Name of the kernel FA_KERNEL_AUTO launches for this call (the dispatch is shape dependent: tile sizes follow the
 * grid), or NULL for arguments fa_forward would reject.  fa_kernel_name: at the headline shape bh = 16, n = 8192. */
const char* fa_kernel_name_for(int32_t dtype, int32_t d, int32_t causal, int64_t bh, int64_t n);
const char* fa_kernel_name(int32_t dtype, int32_t d, int32_t causal);

#ifdef __cplusplus
}
#endif
#endif /* FLASHATTN_AMD_H */
