// fa_torch_binding.cpp -- the reference's pybind translation unit over the C ABI: what src/main.cpp becomes.
//
// Replaces /root/reference/src/main.cpp:1-7 (declaration + PYBIND11_MODULE) and the body of forward() at
// /root/reference/src/flashattention.cu:603-617.  Same Python surface as the reference's JIT-built module --
//     flash = load(name='flash', ...);  out = flash.forward(q, k, v, masking)          (bench_flashattention.py:10,70)
// -- with the hot path in libflashattn_amd.so.  This file contains no device code: it is host C++ against torch-ROCm
// (device memory, current stream) and include/flashattn_amd.h.  Built by `python flashattention.c_amd/build.py --torch-binding`
// (hipcc as the host compiler, no JIT cache); INTEGRATION.md section 1 shows the same text for the reference tree.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include "flashattn_amd.h"

torch::Tensor forward(torch::Tensor Q, torch::Tensor K, torch::Tensor V, bool causal)
{
    TORCH_CHECK(Q.dim() == 3 && Q.sizes() == K.sizes() && Q.sizes() == V.sizes(), "q, k, v: (batch*heads, seq, head_dim), identical shapes");
    TORCH_CHECK(Q.is_cuda() && K.is_cuda() && V.is_cuda(), "q, k, v must live on the GPU");
    TORCH_CHECK(Q.scalar_type() == K.scalar_type() && Q.scalar_type() == V.scalar_type(), "q, k, v must share one dtype");
    TORCH_CHECK(Q.scalar_type() == torch::kFloat32 || Q.scalar_type() == torch::kBFloat16, "float32 (the reference's dtype) or bfloat16");
    Q = Q.contiguous();
    K = K.contiguous();
    V = V.contiguous();
    const int dtype = Q.scalar_type() == torch::kBFloat16 ? FA_DTYPE_BF16 : FA_DTYPE_F32;   // reference: fp32 only (data_ptr<float>())
    torch::Tensor O = torch::empty_like(Q);   // reference: torch::zeros + a dead O_l (flashattention.cu:608-609); every element is written
    const c10::hip::HIPStream stream = c10::hip::getCurrentHIPStream(Q.device().index());
    // the non-allocating form of the boundary: scratch, when the chosen kernels want any (small bf16 grids), is a torch tensor
    const size_t ws_bytes = fa_workspace_bytes(Q.size(0), Q.size(1), (int32_t)Q.size(2), causal ? 1 : 0, dtype, FA_KERNEL_AUTO);
    torch::Tensor ws;
    if (ws_bytes > 0) ws = torch::empty({(int64_t)ws_bytes}, Q.options().dtype(torch::kUInt8));
    const int rc = fa_forward_ws(Q.data_ptr(), K.data_ptr(), V.data_ptr(), O.data_ptr(), /*lse=*/nullptr, Q.size(0), Q.size(1),
                                 (int32_t)Q.size(2), /*scale=*/1.0f,   // the reference's hard-wired scale (flashattention.cu:593,600)
                                 causal ? 1 : 0, dtype, FA_KERNEL_AUTO, ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream.stream());
    TORCH_CHECK(rc == FA_OK, "flashattn_amd: ", fa_last_error());
    return O;   // asynchronous on the current stream (the reference ends with cudaDeviceSynchronize, :594)
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("forward", torch::wrap_pybind_function(forward), "forward");   // src/main.cpp:5-6
}
