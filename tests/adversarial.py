"""Adversarial input families for the fp32 path (test infrastructure, numpy only).

Random data hides COHERENT rounding: when many components of q or k share a value, the residuals a split (or a rounding
fp32 FMA chain) leaves behind all point the same way and the error of a logit grows like sum |q_i k_i| instead of like its
root-mean-square.  VERDICT r04 (weak #1) constructed such inputs against the round-4 default (two bf16 terms per operand:
5.7e-2 in O under a guard that promised 1e-3).  This module generates that family -- constant-component rows, rows with 2-4
distinct values, one token broadcast N times, inputs quantised to a few bits plus an fp32 offset, two dominant keys with
v = +-vmax, many near-equal keys (LSE) -- at a requested LOGIT WIDTH  w = max |q|_2 * max |k|_inf * scale, the quantity the
round-4 guard compared with 90, and the two yardsticks a result is held against:

  attention_f64      the fp64 oracle (oracle/oracle.py restates the reference's Python oracle, bench_flashattention.py:36-48)
  fma_chain_logits   the reference KERNEL's own arithmetic for the logits: a k-ordered fp32 FMA chain
                     (/root/reference/src/flashattention.cu:236-252), one rounding per step -- what "exact fp32" means.

Every generator returns float32 arrays (bh, n, d) and is deterministic in (name, d, width, seed).
"""
from __future__ import annotations

import zlib

import numpy as np

FAMILIES = ("const_two_keys", "const_many_keys", "few_valued", "broadcast_token", "quantised_offset", "gaussian")


def _width(q: np.ndarray, k: np.ndarray) -> float:
    return float(np.sqrt((q.astype(np.float64) ** 2).sum(-1)).max() * np.abs(k).max())


def make(name: str, d: int, width: float, n: int = 512, bh: int = 2, seed: int = 0, vmax: float = 5.0):
    """(q, k, v) of family `name` with max |q|_2 * max |k|_inf == width (scale 1.0)."""
    rng = np.random.default_rng(zlib.crc32(repr((name, d, int(width * 100), seed)).encode()))
    one = np.ones((1, 1, d), np.float32)
    if name == "const_two_keys":
        # every row constant, two kinds of keys 1e-4 apart carrying v = +vmax / -vmax: O = vmax * tanh(half the logit difference)
        a = (1.0 + 1e-3 * rng.uniform(-1, 1, (bh, n, 1))).astype(np.float32)
        q = a * one
        c = np.where(np.arange(n) % 2 == 0, 1.0, 1.0 - 1.1e-4 * (1 + np.arange(n) // 2 % 7))[None, :, None].astype(np.float32)
        k = np.broadcast_to(c * one * np.float32(1.7419), (bh, n, d)).copy()
        v = np.where(np.arange(n) % 2 == 0, vmax, -vmax)[None, :, None].astype(np.float32) * np.ones((bh, 1, d), np.float32)
    elif name == "const_many_keys":
        # n near-equal constant-component keys: the row sum (LSE) sees every logit error with the same sign
        q = (1.0 + 1e-3 * rng.uniform(-1, 1, (bh, n, 1))).astype(np.float32) * one
        k = (1.3077 * (1.0 + 1e-3 * rng.uniform(-1, 1, (bh, n, 1)))).astype(np.float32) * one
        v = (rng.standard_normal((bh, n, d)) * 2).astype(np.float32)
    elif name == "few_valued":
        vals = np.array([0.7071, -0.7071, 1.3183, -1.3183], np.float32)[: 2 + seed % 3]
        q = vals[rng.integers(0, len(vals), (bh, n, d))]
        k = (vals * np.float32(1.1307))[rng.integers(0, len(vals), (bh, n, d))]
        v = (rng.standard_normal((bh, n, d)) * 2).astype(np.float32)
        v[:, ::2] = np.abs(v[:, ::2]) + 1
        v[:, 1::2] = -np.abs(v[:, 1::2]) - 1
    elif name == "broadcast_token":
        t = rng.standard_normal((bh, 1, d)).astype(np.float32)
        q = np.broadcast_to(t, (bh, n, d)).copy()
        k = (t * (1.0 + 2e-4 * rng.standard_normal((bh, n, 1)))).astype(np.float32)
        v = (rng.standard_normal((bh, n, d)) * 2).astype(np.float32)
    elif name == "quantised_offset":
        bits = 2 + seed % 3
        q = (np.round(rng.standard_normal((bh, n, d)) * (1 << bits)) / (1 << bits) + 0.33331).astype(np.float32)
        k = (np.round(rng.standard_normal((bh, n, d)) * (1 << bits)) / (1 << bits) - 0.14287).astype(np.float32)
        v = (rng.standard_normal((bh, n, d)) * 2).astype(np.float32)
    elif name == "gaussian":
        q = rng.standard_normal((bh, n, d)).astype(np.float32)
        k = rng.standard_normal((bh, n, d)).astype(np.float32)
        v = rng.standard_normal((bh, n, d)).astype(np.float32)
    else:
        raise ValueError(name)
    v = np.clip(v, -vmax, vmax).astype(np.float32)
    q = (q * np.float32(width / _width(q, k))).astype(np.float32)
    return q, k, v


def attention_f64(q, k, v, causal=False, scale=1.0):
    """fp64 oracle with the LSE (natural log): the direct formula of bench_flashattention.py:36-48."""
    s = np.einsum("bqd,bkd->bqk", q.astype(np.float64), k.astype(np.float64)) * scale
    if causal:
        n = s.shape[-1]
        s = np.where(np.tril(np.ones((n, n), bool))[None], s, -np.inf)
    m = s.max(-1, keepdims=True)
    p = np.exp(s - m)
    l = p.sum(-1, keepdims=True)
    return (p / l) @ v.astype(np.float64), (m + np.log(l))[..., 0]


def fma_chain_logits(q, k, scale=1.0):
    """The logits as the reference kernel computes them: fp32 operands, a k-ordered chain of fp32 FMAs (one rounding per step,
    flashattention.cu:236-252), scale applied to q first as this repo's exact kernel does.  Returned as float64."""
    qs = (q.astype(np.float32) * np.float32(scale)).astype(np.float32)
    s = np.zeros((q.shape[0], q.shape[1], k.shape[1]), np.float32)
    for i in range(q.shape[-1]):
        prod = qs[:, :, i, None].astype(np.float64) * k[:, None, :, i].astype(np.float64)   # exact in fp64 (24 x 24 bits)
        s = (s.astype(np.float64) + prod).astype(np.float32)                                # one rounding: an FMA
    return s.astype(np.float64)


def attention_from_logits(s, v, causal=False):
    if causal:
        n = s.shape[-1]
        s = np.where(np.tril(np.ones((n, n), bool))[None], s, -np.inf)
    m = s.max(-1, keepdims=True)
    p = np.exp(s - m)
    l = p.sum(-1, keepdims=True)
    return (p / l) @ v.astype(np.float64), (m + np.log(l))[..., 0]


def reference_arithmetic_error(q, k, v, causal=False, scale=1.0):
    """(max |O| error, max |LSE| error) that the reference's OWN logit arithmetic (fp32 FMA chain) leaves against fp64, everything
    after the logits exact: the floor no fp32-accumulating kernel can be asked to beat on this input."""
    o_ref, l_ref = attention_f64(q, k, v, causal, scale)
    o, l = attention_from_logits(fma_chain_logits(q, k, scale), v, causal)
    return float(np.abs(o - o_ref).max()), float(np.abs(l - l_ref).max())


def rows_f64(q, k, v, rows, causal=False, scale=1.0, chain=False):
    """O[rows] and LSE[rows] of ONE slab (q, k, v: (n, d)) in fp64 -- the oracle on a sample of query rows, cheap at any n (soak slices and
    full-size configs).  chain=True: the logits through the reference kernel's fp32 FMA chain instead (fma_chain_logits)."""
    rows = np.asarray(rows)
    if chain:
        s = fma_chain_logits(q[None, rows], k[None], scale)[0]
    else:
        s = (q[rows].astype(np.float64) @ k.astype(np.float64).T) * scale
    if causal:
        s = np.where(np.arange(k.shape[0])[None, :] <= rows[:, None], s, -np.inf)
    m = s.max(-1, keepdims=True)
    p = np.exp(s - m)
    l = p.sum(-1, keepdims=True)
    return (p / l) @ v.astype(np.float64), (m + np.log(l))[:, 0]


def p_rounding_bound(q, k, v, rows, causal=False, scale=1.0, rel=2.0 ** -8):
    """Worst-case |O| error of a kernel that rounds each softmax weight to 8 significant bits (bf16 P: relative error <= 2^-8 per weight,
    signs free): sum_j w_j |delta_j| |v_j - O| <= rel * max over (row, column) of sum_j w_j |v_jc - O_c| -- the weighted mean absolute
    deviation of V under the row's softmax.  Two equally dominant keys with v = +-V give rel * V = 1/4 * 2^-7 * |v1 - v2| (VERDICT r04
    weak #2).  Computed from the data in fp64 for the sampled rows of one slab."""
    rows = np.asarray(rows)
    s = (q[rows].astype(np.float64) @ k.astype(np.float64).T) * scale
    if causal:
        s = np.where(np.arange(k.shape[0])[None, :] <= rows[:, None], s, -np.inf)
    w = np.exp(s - s.max(-1, keepdims=True))
    w /= w.sum(-1, keepdims=True)
    v64 = v.astype(np.float64)
    o = w @ v64
    mad = 0.0
    for r0 in range(0, len(rows), 8):   # (chunks: the (rows, keys, d) deviation tensor of a long slab does not fit at once)
        wr, orr = w[r0:r0 + 8], o[r0:r0 + 8]
        mad = max(mad, float(np.einsum("rj,rjc->rc", wr, np.abs(v64[None, :, :] - orr[:, None, :])).max()))
    return float(rel * mad), o
