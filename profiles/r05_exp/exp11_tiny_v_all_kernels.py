"""Relative accuracy on tiny values, every kernel family the dispatch reaches: max |O - ref| / 2^e for V = N(0,1) * 2^e against rung 0 (fp32
arithmetic on the device, itself checked against the fp64 oracle elsewhere)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import flashattention_c_amd as fa  # noqa: E402
from flashattention_c_amd import _cabi  # noqa: E402

dev = torch.device("cuda:0")
L = _cabi.lib()
SHAPES = [(1024, 128, 64), (512, 256, 64), (128, 1024, 64), (16, 8192, 64), (1, 8192, 64), (16, 1024, 64), (128, 1024, 128), (16, 4096, 128), (128, 1024, 32),
          (16, 8192, 32), (40, 700, 128), (256, 100, 64)]
for dt, dtid in ((torch.float32, _cabi.FA_DTYPE_F32), (torch.bfloat16, _cabi.FA_DTYPE_BF16)):
    for bh, n, d in SHAPES:
        for causal in (False, True):
            q, k, v = (torch.randn(bh, n, d, device=dev, dtype=dt) for _ in range(3))
            name = L.fa_kernel_name_for(dtid, d, int(causal), bh, n).decode().replace("fa_fwd_", "").replace("_kernel", "")
            row = f"{str(dt)[6:]:8s} {bh:5d}x{n:<5d} d={d:<3d} causal={int(causal)} {name:12s}"
            for e in (0, -30, -45, -60):
                vv = (v.float() * 2.0 ** e).to(dt)
                ref = fa.forward(q.float(), k.float(), vv.float(), causal, kernel="naive")
                o = fa.forward(q, k, vv, causal).float()
                row += f"  2^{e}: {float((o - ref).abs().max()) / 2.0 ** e:.1e}"
                if dt == torch.bfloat16:
                    o2 = fa.forward(q, k, vv, causal, out_dtype=torch.float32)
                    row += f" / {float((o2 - ref).abs().max()) / 2.0 ** e:.1e}"
            print(row, flush=True)
