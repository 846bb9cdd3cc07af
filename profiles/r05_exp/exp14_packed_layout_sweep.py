"""The llm.c entry (packed (B, T, 3C) fp32, causal, 1/sqrt(hs)): every T in 1 .. 320 and a few long ones, NH in {1, 2, 3, 4, 12}, hs in
{32, 64, 128}, against rung 0 on the unpacked tensors."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import flashattention_c_amd as fa  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(11)
worst, n_bad, n_all = {}, 0, 0
for hs in (32, 64, 128):
    for nh in (1, 2, 3, 4, 12):
        for T in list(range(1, 321)) + [511, 512, 513, 1023, 1024, 1025, 2048, 4096]:
            B = 2 if T <= 1025 else 1
            inp = torch.randn(B, T, 3 * nh * hs, generator=g).to(dev)
            got = fa.forward_packed_qkv(inp, nh)
            qq, kk, vv = (inp[:, :, i * nh * hs:(i + 1) * nh * hs].reshape(B, T, nh, hs).permute(0, 2, 1, 3).reshape(B * nh, T, hs).contiguous() for i in range(3))
            want = fa.forward(qq, kk, vv, True, scale=hs ** -0.5, kernel="naive").reshape(B, nh, T, hs).permute(0, 2, 1, 3).reshape(B, T, nh * hs)
            e = float((got - want).abs().max())
            n_all += 1
            if not e < 1e-4:
                n_bad += 1
                print("outside 1e-4:", hs, nh, T, e)
            if e != e or e > worst.get((hs, nh), (0.0, 0))[0]:
                worst[(hs, nh)] = (e, T)
for key in sorted(worst):
    print(f"hs={key[0]:3d} NH={key[1]:2d}: worst {worst[key][0]:.2e} at T={worst[key][1]}")
print(f"{n_all} launches, {n_bad} outside the reference's 1e-4")
