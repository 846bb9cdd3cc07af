"""Explicit kernel choices (the ids a caller may pass) over sequence lengths: every family and tiling of the PRODUCT library that accepts the
shape, against rung 0; unsupported (kernel, shape) pairs must fail with a status, not with wrong numbers."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import flashattention_c_amd as fa  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
F32 = ["exact", "exact:1", "exact:2", "split", "split:1", "split:3", "split:4", "split:5", "mfma"]
BF16 = ["mfma", "mfma:7", "mfma:30", "mfma:50", "mfma:25", "mfma:42", "pb2", "pb2:1", "split"]
LENGTHS = [(3, n) for n in list(range(1, 130)) + [191, 192, 193, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1024, 1025, 2047, 2048, 2049]] + [(1, 4096), (2, 4097), (1, 8192), (16, 1024), (130, 256)]
stat = {}
for d in (32, 64, 128):
    for bh, n in LENGTHS:
        q, k, v = (torch.randn(bh, n, d, generator=g).to(dev) for _ in range(3))
        qb, kb, vb = (t.to(torch.bfloat16) for t in (q, k, v))
        for causal in (False, True):
            ref = fa.forward(q, k, v, causal, kernel="naive")
            refb = fa.forward(qb.float(), kb.float(), vb.float(), causal, kernel="naive")
            for kerns, args, r, tol in ((F32, (q, k, v), ref, 3e-4), (BF16, (qb, kb, vb), refb, None)):
                for kern in kerns:
                    for odt in ((torch.float32,) if args[0].dtype == torch.float32 else (torch.bfloat16, torch.float32)):
                        key = (str(args[0].dtype)[6:], kern, str(odt)[6:], d, causal)
                        st = stat.setdefault(key, [0, 0, 0.0, None])
                        try:
                            out = torch.full((bh, n, d), float("nan"), device=dev, dtype=odt)
                            fa.forward(*args, causal, kernel=kern, out=out)
                        except Exception as e:     # a status from the C ABI: refused, fine
                            st[1] += 1
                            continue
                        err = float((out.float() - r).abs().max())
                        t = tol if tol is not None else (2.5e-2 if (odt == torch.bfloat16 or kern.startswith("mfma")) else 2e-4)
                        st[0] += 1
                        if err != err or err > st[2]:
                            st[2], st[3] = err, (bh, n)
                        if not err < t:
                            print("OUTSIDE", key, bh, n, err, flush=True)
for key in sorted(stat, key=str):
    st = stat[key]
    print(key, f"ran {st[0]}, refused {st[1]}, worst {st[2]:.2e} at {st[3]}")
