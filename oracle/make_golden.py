#!/usr/bin/env python3
"""oracle/make_golden.py -- generates tests/golden/*.npz.  TEST INFRASTRUCTURE.

Runs ONLY in the build container (needs /root/reference); the GPU box never executes it.  It pins
the oracle to the reference by executing the reference's own code:

* ``manual_attention_unmasked`` / ``manual_attention_masking`` -- the two oracle functions of
  ``/root/reference/bench_flashattention.py:36-48``.  The script itself cannot be imported (it JIT
  builds a CUDA extension and calls ``.cuda()`` at import, ``:10,31-33``), so the two FunctionDef
  nodes are lifted out of its AST and compiled as they are, with ``torch``, ``F`` and the module
  global ``v`` they close over supplied here.  They are evaluated on fp64 copies of the inputs
  (key ``o``) and, for the first case, also in fp32 exactly as the reference bench does (``o_f32``).
* ``attention_forward_cpu`` -- ``/root/reference/src/llm.c/attention_forward.cu:53-125`` compiled
  into ``oracle/_ref`` by ``oracle/Makefile`` (packed-QKV, causal, 1/sqrt(hs)).

What is written is data only: inputs, the reference's outputs, and the case parameters.
Scaled cases: the reference functions have no scale argument (the ``/math.sqrt`` is commented out,
``:37,44``); a power-of-two scale is applied by pre-multiplying q, which is exact in binary
floating point, so the reference's output on (scale*q, k, v) IS softmax(scale * q k^T) v.
"""
from __future__ import annotations

import ast
import math
import os
import sys

import numpy as np
import torch
from torch.nn import functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

REF_BENCH = "/root/reference/bench_flashattention.py"
OUT_DIR = os.path.join(ROOT, "tests", "golden")


def load_reference_oracle_functions():
    """Compile the reference's two oracle functions straight from its source file."""
    with open(REF_BENCH, "r") as f:
        tree = ast.parse(f.read(), filename=REF_BENCH)
    wanted = {"manual_attention_unmasked", "manual_attention_masking"}
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in nodes} == wanted, "reference oracle functions not found"
    ns = {"torch": torch, "F": F, "math": math, "print": lambda *a, **k: None}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), REF_BENCH, "exec"), ns)
    return ns


def reference_attention(ns, q, k, v, causal):
    ns["v"] = v  # the reference functions read `v` from module scope (bench_flashattention.py:39,47)
    fn = ns["manual_attention_masking"] if causal else ns["manual_attention_unmasked"]
    return fn(q, k)


def randn(gen, *shape):
    return torch.randn(*shape, generator=gen, dtype=torch.float32)


def to_bf16_values(t):
    return t.to(torch.bfloat16).to(torch.float32)


def main():
    os.makedirs(OUT_DIR, exist_ok=True)
    ns = load_reference_oracle_functions()
    cases = []

    def add(name, q, k, v, causal, scale, with_f32=False, note=""):
        assert math.log2(scale) == int(math.log2(scale)), "scale must be a power of two (exact pre-scaling)"
        qs = q * scale
        o64 = reference_attention(ns, qs.double(), k.double(), v.double(), causal).numpy()
        arrs = dict(q=q.numpy(), k=k.numpy(), v=v.numpy(), o=o64,
                    causal=np.int32(causal), scale=np.float64(scale), note=np.array(note))
        if with_f32:
            arrs["o_f32"] = reference_attention(ns, qs, k, v, causal).numpy()
        # cross-check the generator against our restatements before committing anything
        mine = orc.attention_f64(q.numpy(), k.numpy(), v.numpy(), causal=causal, scale=scale)
        err = float(np.abs(mine - o64).max())
        assert err < 1e-12 * max(1.0, float(np.abs(o64).max())) + 1e-12, (name, err)
        np.savez_compressed(os.path.join(OUT_DIR, name + ".npz"), **arrs)
        cases.append((name, tuple(q.shape), causal, scale, err))

    g = torch.Generator().manual_seed(20240611)
    # (i) README-family head dim, three key tiles of 32
    for causal in (False, True):
        q, k, v = randn(g, 2, 96, 64), randn(g, 2, 96, 64), randn(g, 2, 96, 64)
        add(f"bh2_n96_d64_{'causal' if causal else 'full'}", q, k, v, causal, 1.0, with_f32=not causal)
    # (ii) config-1 family: d = 32
    for causal in (False, True):
        q, k, v = randn(g, 2, 128, 32), randn(g, 2, 128, 32), randn(g, 2, 128, 32)
        add(f"bh2_n128_d32_{'causal' if causal else 'full'}", q, k, v, causal, 1.0)
    # (iii) ragged sequence length (N % 32 != 0): oracle semantics, not the CUDA kernel's F8 behaviour
    for causal in (False, True):
        q, k, v = randn(g, 3, 80, 64), randn(g, 3, 80, 64), randn(g, 3, 80, 64)
        add(f"bh3_n80_d64_{'causal' if causal else 'full'}", q, k, v, causal, 1.0, note="ragged N")
    # (iv) scaled softmax (1/sqrt(64) = 2^-3)
    q, k, v = randn(g, 1, 256, 64), randn(g, 1, 256, 64), randn(g, 1, 256, 64)
    add("bh1_n256_d64_scale8th_full", q, k, v, False, 0.125)
    add("bh1_n256_d64_scale8th_causal", q, k, v, True, 0.125)
    # (v) forced-rescale spike: key row 200 is 6x query row 37, so row 37's running max jumps by
    #     ~6*|q37|^2 ~ 380 at the tile holding key 200, long after earlier tiles were accumulated
    q, k, v = randn(g, 1, 256, 64), randn(g, 1, 256, 64), randn(g, 1, 256, 64)
    k[0, 200] = 6.0 * q[0, 37]
    k[0, 13] = 3.0 * q[0, 150]
    add("bh1_n256_d64_spike_full", q, k, v, False, 1.0, note="spike keys 200<-q37, 13<-q150")
    # (vi) d = 128 and a single short tile
    q, k, v = randn(g, 1, 64, 128), randn(g, 1, 64, 128), randn(g, 1, 64, 128)
    add("bh1_n64_d128_full", q, k, v, False, 1.0)
    q, k, v = randn(g, 2, 17, 64), randn(g, 2, 17, 64), randn(g, 2, 17, 64)
    add("bh2_n17_d64_causal", q, k, v, True, 1.0, note="N smaller than one tile")
    # (vii) bf16-valued inputs (what the bf16 MFMA path consumes), scale 1 and 1/8
    q, k, v = (to_bf16_values(randn(g, 2, 192, 64)) for _ in range(3))
    add("bh2_n192_d64_bf16vals_full", q, k, v, False, 1.0, note="inputs exactly representable in bf16")
    add("bh2_n192_d64_bf16vals_scale8th_causal", q, k, v, True, 0.125, note="inputs exactly representable in bf16")

    # (viii) llm.c packed layout from the reference's own C function (oracle/_ref)
    orc.build()
    assert orc.have_reference_build(), "oracle/_ref missing: run `make -C oracle ref`"
    B, T, C, NH = 2, 96, 128, 2
    rng = np.random.default_rng(7)
    inp = (rng.random((B, T, 3 * C), dtype=np.float32) * 2.0 - 1.0).astype(np.float32)  # U(-1,1) as common.h:46-52
    out_ref = orc.reference_attention_packed_f32(inp, NH)
    mine = orc.attention_packed_f32(inp, NH)
    err = float(np.abs(mine - out_ref).max())
    assert err <= 1e-6, err
    np.savez_compressed(os.path.join(OUT_DIR, "llmc_packed_b2_t96_c128_nh2.npz"),
                        inp=inp, out=out_ref, n_head=np.int32(NH))
    cases.append(("llmc_packed_b2_t96_c128_nh2", inp.shape, True, 1 / math.sqrt(C // NH), err))

    with open(os.path.join(OUT_DIR, "MANIFEST.txt"), "w") as f:
        f.write("# generated by oracle/make_golden.py from the reference's own oracle code; data only\n")
        f.write("# name | input shape | causal | scale | max|our C restatement - reference output|\n")
        for c in cases:
            f.write(" | ".join(str(x) for x in c) + "\n")
    for c in cases:
        print(c)


if __name__ == "__main__":
    main()
