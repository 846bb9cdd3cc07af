// fa_kernels.h -- launcher declarations shared by the C-ABI translation unit and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include "fa_common.h"

namespace fa {

// Each launcher enqueues one forward on `stream` and returns hipGetLastError().
// `variant` selects among co-compiled tilings of the same kernel (0 = default); used by the ablation driver.
hipError_t launch_naive(const FwdParams& p, int d, int causal, int dtype, hipStream_t stream);   // rung 0: fp32 arithmetic, fp32 or bf16 tensors, any d <= 256
hipError_t launch_fwd_f32(const FwdParams& p, int d, int causal, int variant, hipStream_t stream, int io = 0);        // exact fp32: d in {32, 64, 128} + the wide set
hipError_t launch_fwd_f32_wide(const FwdParams& p, int d, int causal, int variant, int io, hipStream_t stream);   // d in {96, 160, 192, 224, 256}
// ... on bf16 tensors (io = 1: bf16 output, 2: fp32 output / key-share partials): fa_fwd_f32_wide_bf16.hip
hipError_t launch_fwd_f32_wide_bf16(const FwdParams& p, int d, int causal, int variant, int out_bf16, hipStream_t stream);
hipError_t launch_fwd_bf16(const FwdParams& p, int d, int causal, int out_f32, int variant, hipStream_t stream);
// fp32 tensors on the bf16 matrix pipe (three products of two-term bf16 splits); called by launch_fwd_f32
hipError_t launch_f32_split(const FwdParams& p, int d, int causal, int mode, hipStream_t stream);
// bf16 tensors through the same split machinery (K, V exact in one term; Q' and P split): the accurate bf16 mode
hipError_t launch_bf16_split(const FwdParams& p, int d, int causal, int out_f32, int mode, hipStream_t stream);

// bf16 kernels living in their own translation units (called by launch_fwd_bf16)
hipError_t launch_bf16_pipelined(const FwdParams& p, int d, int nwaves, int causal, int out_f32, int mode, hipStream_t stream);
const char* bf16_kernel_name(int64_t bh, int64_t n, int d, int causal);  // the kernel the product dispatch picks
bool bf16_pipelined_supported(const FwdParams& p, int d);  // d in {32, 64} and the slab addressable with 32-bit byte offsets
hipError_t launch_bf16_x4(const FwdParams& p, int causal, int out_f32, int mode, hipStream_t stream);  // D = 64, 128 rows/wave
hipError_t launch_bf16_x2(const FwdParams& p, int d, int causal, int out_f32, int mode, hipStream_t stream);  // D = 64 / 128, 64 rows/wave, one wave per SIMD
// fp16-P ("accurate") kernels; p.v = the fp16 copy of V made by launch_cvt_v_f16.  launch_bf16_p16 picks the tiling (NB = 4 / NB = 2)
hipError_t launch_bf16_p16(const FwdParams& p, int d, int causal, int out_f32, hipStream_t stream);
bool bf16_p16_supported(const FwdParams& p, int d);   // d in {32, 64, 128} and the slab addressable with 32-bit byte offsets
hipError_t launch_bf16_x4_p16(const FwdParams& p, int causal, int out_f32, hipStream_t stream);
bool bf16_p16_uses_x4(int64_t bh, int64_t n, int causal);
hipError_t launch_bf16_x2_p16_d32(const FwdParams& p, int causal, int out_f32, hipStream_t stream);
hipError_t launch_bf16_x2_p16_d64(const FwdParams& p, int causal, int out_f32, hipStream_t stream);
hipError_t launch_bf16_x2_p16_d128(const FwdParams& p, int causal, int out_f32, hipStream_t stream);
// two-term fp16 P (hi + lo): the same chain, kernel fa_fwd_bf16_x2_p16x2_kernel (NB = 2 at every grid size)
hipError_t launch_bf16_p16x2(const FwdParams& p, int d, int causal, int out_f32, hipStream_t stream);
hipError_t launch_bf16_x2_p16x2_d32(const FwdParams& p, int causal, int out_f32, hipStream_t stream);
hipError_t launch_bf16_x2_p16x2_d64(const FwdParams& p, int causal, int out_f32, hipStream_t stream);
hipError_t launch_bf16_x2_p16x2_d128(const FwdParams& p, int causal, int out_f32, hipStream_t stream);
// P as bf16 hi + bf16 lo (fa_fwd_bf16_x{4,2}_pb2_kernel): one launch, p.v = the caller's bf16 V, any layout the bf16-P kernels take
hipError_t launch_bf16_pb2(const FwdParams& p, int d, int causal, int out_f32, int variant, hipStream_t stream);
bool bf16_pb2_uses_x4(int64_t bh, int64_t n, int causal);
#define FA_PB2_DECL(name) hipError_t name##_f32out(const FwdParams& p, int causal, hipStream_t stream); hipError_t name##_bf16out(const FwdParams& p, int causal, hipStream_t stream)
FA_PB2_DECL(launch_bf16_x4_pb2);
FA_PB2_DECL(launch_bf16_x2_pb2_d32);
FA_PB2_DECL(launch_bf16_x2_pb2_d64);
FA_PB2_DECL(launch_bf16_x2_pb2_d128);
#undef FA_PB2_DECL
hipError_t launch_cvt_v_f16(const void* src, void* dst, int64_t count, uint32_t* flag, uint32_t serial, hipStream_t stream);
// combine of a key-split launch: partial outputs fp32 [S][bh][n][d], partial log-sum-exps [bh][S][n] -> p.o (and p.lse)
hipError_t launch_combine_splits(const FwdParams& p, const float* o_part, const float* lse_part, int S, int d, int out_f32, hipStream_t stream);
// fp32 tensors, head dim 64, long non-causal rows: pre-split K / V (launch_t3_prepass) + the static-slot three-product kernel
bool f32_t3_supported(const FwdParams& p, int d, int causal);
hipError_t launch_f32_t3(const FwdParams& p, int abl, hipStream_t stream);   // ablation library only
hipError_t launch_t3_prepass(const void* q, const void* k, const void* v, void* scratch, int64_t count, float scale_log2e, unsigned long long* stats,
                             uint32_t serial, hipStream_t stream);
hipError_t launch_bf16_pp2(const FwdParams& p, int causal, int out_f32, int variant, hipStream_t stream);

}  // namespace fa
