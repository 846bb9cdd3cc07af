"""r06 experiment 4 (the arithmetic kill test of a ONE-term fp16 P): the derived bound of a kernel that rounds every softmax
weight to `bits` significant bits (round to nearest; row sum from the same rounded P) is  rel * max over (row, column) of sum_j w_j |v_jc - O_c|,
rel = 2^-bits (tests/adversarial.py: p_rounding_bound).  Evaluated here over EVERY row of the c4 tensors (B=2 H=8 d=64 N=8192, bf16-rounded
randn, scale 1) for three seeds, on the GPU in fp64 (torch only: no library call)."""
import torch

dev = torch.device("cuda", 0)
for seed in (0, 1, 2):
    g = torch.Generator(device=dev).manual_seed(seed)
    q, k, v = (torch.randn(16, 8192, 64, device=dev, generator=g).to(torch.bfloat16).double() for _ in range(3))
    mad = 0.0
    for b in range(16):
        for r0 in range(0, 8192, 256):
            s = q[b, r0:r0 + 256] @ k[b].T
            w = torch.softmax(s, dim=-1)
            o = w @ v[b]
            for c0 in range(0, 64, 8):   # (rows, keys, 8 columns) at a time
                dev_abs = (v[b][None, :, c0:c0 + 8] - o[:, None, c0:c0 + 8]).abs()
                mad = max(mad, float(torch.einsum("rj,rjc->rc", w, dev_abs).max()))
    print(f"seed {seed}: max sum_j w_j |v_j - O| = {mad:.4f}   bound bf16 P (2^-8+2^-10) {mad * (2**-8 + 2**-10):.3e}   one fp16 term (2^-11) {mad * 2**-11:.3e}"
          f"   (2^-12 if half-ulp were relative to the binade top) {mad * 2**-12:.3e}   two bf16 terms (2^-17) {mad * 2**-17:.3e}", flush=True)
