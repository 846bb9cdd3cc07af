"""Magnitude sweep through every dispatched kernel family: V x 2^e (large) and Q x 2^e (tiny and large logits), against the fp64 oracle on
sampled rows of slab 0 (rung 0 on the device is itself fp32 arithmetic).  Errors relative to max |v|."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import flashattention_c_amd as fa  # noqa: E402
from flashattention_c_amd import _cabi  # noqa: E402
from tests import adversarial as adv  # noqa: E402

dev = torch.device("cuda:0")
L = _cabi.lib()
SHAPES = [(512, 256, 64), (128, 1024, 64), (16, 8192, 64), (1, 8192, 64), (128, 1024, 128), (16, 4096, 128), (128, 1024, 32), (16, 8192, 32), (40, 700, 128)]
rng = np.random.default_rng(5)
for dt, dtid in ((torch.float32, _cabi.FA_DTYPE_F32), (torch.bfloat16, _cabi.FA_DTYPE_BF16)):
    for bh, n, d in SHAPES:
        for causal in (False, True):
            q, k, v = (torch.randn(bh, n, d, device=dev, dtype=dt) for _ in range(3))
            rows = np.unique(np.concatenate([[0, 1, n - 1], rng.integers(0, n, 96)]))
            name = L.fa_kernel_name_for(dtid, d, int(causal), bh, n).decode().replace("fa_fwd_", "").replace("_kernel", "")
            row = f"{str(dt)[6:]:8s} {bh:4d}x{n:<5d} d={d:<3d} c={int(causal)} {name:10s}"
            for what, e in (("v", 40), ("v", 100), ("q", -20), ("q", -6), ("q", 2), ("q", 4)):
                qq = (q.float() * 2.0 ** e).to(dt) if what == "q" else q
                vv = (v.float() * 2.0 ** e).to(dt) if what == "v" else v
                o, lse = fa.forward(qq, k, vv, causal, return_lse=True, out_dtype=torch.float32 if dt == torch.bfloat16 else None)
                r = fa.last_forward_route()
                o64, l64 = adv.rows_f64(qq[0].float().cpu().numpy(), k[0].float().cpu().numpy(), vv[0].float().cpu().numpy(), rows, causal)
                vm = float(vv.float().abs().max())
                eo = float(np.abs(o[0].float().cpu().numpy()[rows] - o64).max()) / vm
                el = float(np.abs(lse[0].cpu().numpy()[rows] - l64).max())
                row += f"  {what}x2^{e}: {eo:.0e}/{el:.0e}" + (f"[r{r}]" if r == 2 else "")
            print(row, flush=True)
