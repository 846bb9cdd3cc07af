"""How often does the optimistic attempt of a tile fail, per data family of the soak?  (fa_get_stats: tiles_redone, counted by the kernels.)
One case per family at BH = 8, N = 4096, d in {64, 128}, scale 1; tiles = what the launch has (256- or 128-row workgroups)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import flashattention_c_amd as fa  # noqa: E402
from tests import soak_fuzz as sf  # noqa: E402

NAMES = {0: "N(0,1)", 1: "wide logits (q x 3)", 2: "planted dominant keys", 3: "constant V", 4: "zero Q", 5: "large V", sf.V_OFFSET: "V offset"}
for i, nm in enumerate(sf.COHERENT):
    NAMES[6 + i] = "coherent: " + nm


def moved(fn):
    torch.cuda.synchronize()
    a = fa.stats()
    fn()
    torch.cuda.synchronize()
    b = fa.stats()
    return b["tiles_redone"] - a["tiles_redone"], b["workgroups_fp32"] - a["workgroups_fp32"]


dev = torch.device("cuda:0")
fa.forward(*(torch.randn(1, 256, 64, device=dev) for _ in range(3)), False)
print("family: tiles redone (of workgroups launched ~ BH * N / 256) for fp32 default | bf16 -> bf16 | bf16 -> fp32;  causal in brackets")
for d in (64, 128):
    for fam in range(sf.N_FAMILIES):
        rng = np.random.default_rng(fam)
        g = torch.Generator(device="cpu").manual_seed(1000 + fam)
        bh, n = 8, 4096
        q, k, v, _ = sf.make_data(rng, g, fam, bh, n, d, case_seed=fam)
        qd, kd, vd = (t.to(dev) for t in (q, k, v))
        qb, kb, vb = (t.to(torch.bfloat16) for t in (qd, kd, vd))
        out = []
        for causal in (False, True):
            a = moved(lambda: fa.forward(qd, kd, vd, causal))
            b = moved(lambda: fa.forward(qb, kb, vb, causal))
            c = moved(lambda: fa.forward(qb, kb, vb, causal, out_dtype=torch.float32))
            out.append((a, b, c))
        (a0, b0, c0), (a1, b1, c1) = out
        print(f"d={d:3d} {NAMES[fam]:32s} fp32 {a0[0]:4d} [{a1[0]:4d}] (fp32-arith wgs {a0[1]} [{a1[1]}])   bf16 {b0[0]:4d} [{b1[0]:4d}]   bf16->fp32 {c0[0]:4d} [{c1[0]:4d}]")
