"""CPU: the yardsticks of tests/test_gpu_adversarial.py (tests/adversarial.py, numpy only) against closed forms and the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import adversarial as adv  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def test_generators_hit_the_requested_width_and_are_deterministic():
    for name in adv.FAMILIES:
        for d in (32, 128):
            q, k, v = adv.make(name, d, 77.0, n=96, bh=2, seed=3)
            assert q.dtype == k.dtype == v.dtype == np.float32 and q.shape == (2, 96, d)
            assert abs(adv._width(q, k) - 77.0) < 1e-3 and np.abs(v).max() <= 5.0
            q2, k2, v2 = adv.make(name, d, 77.0, n=96, bh=2, seed=3)
            assert np.array_equal(q, q2) and np.array_equal(k, k2) and np.array_equal(v, v2)


def test_fp64_yardstick_agrees_with_the_pinned_oracle():
    rng = np.random.default_rng(0)
    q, k, v = (rng.standard_normal((2, 70, 64)).astype(np.float32) for _ in range(3))
    for causal in (False, True):
        o, lse = adv.attention_f64(q, k, v, causal, 0.5)
        ref, lse_ref = orc.attention_f64(q, k, v, causal=causal, scale=0.5, return_lse=True)
        assert np.abs(o - ref).max() < 1e-12 and np.abs(lse - lse_ref).max() < 1e-12
        rows = np.array([0, 3, 69])
        o_r, l_r = adv.rows_f64(q[1], k[1], v[1], rows, causal, 0.5)
        assert np.abs(o_r - o[1][rows]).max() < 1e-12 and np.abs(l_r - lse[1][rows]).max() < 1e-12


def test_fma_chain_is_a_float32_fmaf_chain():
    """One rounding per step, k-ordered: against a scalar loop with np.float32 arithmetic on exactly representable products, and within
    the textbook bound d * 2^-24 * sum |q_i k_i| of the exact value otherwise."""
    rng = np.random.default_rng(1)
    q = rng.integers(-8, 9, (1, 5, 16)).astype(np.float32)
    k = rng.integers(-8, 9, (1, 7, 16)).astype(np.float32)
    assert np.array_equal(adv.fma_chain_logits(q, k), np.einsum("bqd,bkd->bqk", q.astype(np.float64), k.astype(np.float64)))   # integers: exact
    q, k = rng.standard_normal((1, 9, 128)).astype(np.float32) * 3, rng.standard_normal((1, 11, 128)).astype(np.float32) * 3
    s = adv.fma_chain_logits(q, k)
    exact = np.einsum("bqd,bkd->bqk", q.astype(np.float64), k.astype(np.float64))
    bound = 128 * 2.0 ** -24 * np.einsum("bqd,bkd->bqk", np.abs(q).astype(np.float64), np.abs(k).astype(np.float64))
    assert (np.abs(s - exact) <= bound).all() and np.abs(s - exact).max() > 0.0
    assert (s.astype(np.float32) == s).all()   # every value is a float32


def test_reference_arithmetic_is_not_an_oracle_on_coherent_inputs():
    """The finding of round 5 in numbers that need no GPU: on constant-component rows with two dominant keys at width 89.5 the fp32 FMA
    chain alone is off by more than 1e-3 in O (d = 64) -- fp32 accumulation rounds at the magnitude of the partial sum."""
    q, k, v = adv.make("const_two_keys", 64, 89.5, n=256, bh=1, seed=64)
    e_o, e_l = adv.reference_arithmetic_error(q, k, v)
    assert 1e-3 < e_o < 2e-2 and e_l < e_o
    q, k, v = adv.make("gaussian", 64, 89.5, n=256, bh=1, seed=64)
    e_o, _ = adv.reference_arithmetic_error(q, k, v)
    assert e_o < 1e-4


def test_p_rounding_bound_on_two_equal_keys():
    """Two equally dominant keys with v = +-V: the bound is rel * V = 1/4 * 2^-7 * |v1 - v2| at rel = 2^-8."""
    d, V = 8, 5.0
    q = np.zeros((1, d), np.float32)
    k = np.zeros((2, d), np.float32)
    v = np.array([[V] * d, [-V] * d], np.float32)
    b, o = adv.p_rounding_bound(q, k, v, [0])
    assert abs(b - 2.0 ** -8 * V) < 1e-12 and np.abs(o).max() < 1e-12
    assert abs(b - 0.25 * 2.0 ** -7 * 2 * V) < 1e-12
