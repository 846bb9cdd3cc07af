"""Round 5, experiment 5 (GPU): V with a common offset (V = off + N(0, 1)).  The split kernel carries V as bf16 hi + lo (16 bits): its P.V error
is relative to max|v|, fp32 arithmetic's too but with 24 bits.  Columns: the reference kernel's own recurrence in fp32 (oracle/: flash_tiled_f32,
flashattention.cu:214-354) / kernel="exact" / the default, each against the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import flashattention_c_amd as fa
import adversarial as adv
from oracle import oracle as orc
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
for d in (64, 128):
    q, k, x = (rng.standard_normal((2, 512, d)).astype(np.float32) for _ in range(3))
    for scale in (1.0, 0.02):
        for off in (0.0, 10.0, 100.0, 1000.0, 1e4):
            v = (x + np.float32(off)).astype(np.float32)
            o_ref, _ = adv.attention_f64(q, k, v, scale=scale)
            e_cpu = float(np.abs(orc.flash_tiled_f32(q, k, v, scale=scale).astype(np.float64) - o_ref).max())
            out = []
            for kern in ("exact", "auto"):
                o = fa.forward(*(torch.from_numpy(t).to(dev) for t in (q, k, v)), False, scale=scale, kernel=kern).cpu().numpy().astype(np.float64)
                out.append(float(np.abs(o - o_ref).max()))
            print(f"d={d:3d} scale={scale:<4} V offset {off:7.0f}: reference recurrence (fp32, CPU) {e_cpu:.1e}  exact {out[0]:.1e}  default {out[1]:.1e}   (3 * 2^-17 * max|v| = {3 * 2.0 ** -17 * np.abs(v).max():.1e})", flush=True)
