"""CPU: the oracle against the reference-generated golden vectors (tests/golden, made by oracle/make_golden.py from the
reference's own oracle code) and its restatements against each other."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests.conftest import GOLDEN_DIR, golden_cases


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return z["q"], z["k"], z["v"], z["o"], bool(z["causal"]), float(z["scale"]), z


@pytest.mark.parametrize("name", golden_cases())
def test_c_oracle_matches_reference_vectors(name):
    q, k, v, o_ref, causal, scale, _ = _load(name)
    o = orc.attention_f64(q, k, v, causal=causal, scale=scale)
    assert o.shape == o_ref.shape
    # both are fp64 evaluations of the same formula; only summation order differs
    assert np.abs(o - o_ref).max() <= 1e-12 * max(1.0, np.abs(o_ref).max())


@pytest.mark.parametrize("name", golden_cases())
def test_numpy_oracle_matches_reference_vectors(name):
    q, k, v, o_ref, causal, scale, _ = _load(name)
    o = orc.attention_numpy(q, k, v, causal=causal, scale=scale)
    assert np.abs(o - o_ref).max() <= 1e-12 * max(1.0, np.abs(o_ref).max())


def test_reference_fp32_output_is_within_its_own_tolerance():
    # the reference bench evaluates its oracle in fp32 (bench_flashattention.py:62-63); our fp32 oracle agrees with that
    q, k, v, o_ref, causal, scale, z = _load("bh2_n96_d64_full")
    o32 = orc.attention_f32(q, k, v, causal=causal, scale=scale)
    assert np.abs(o32 - z["o_f32"]).max() < 2e-5
    assert np.abs(o32.astype(np.float64) - o_ref).max() < 1e-6


@pytest.mark.parametrize("name", golden_cases())
def test_tiled_recurrence_matches(name):
    """The CUDA kernel's tile-by-tile online softmax (flashattention.cu:214-354) restated in fp32 agrees with the direct formula."""
    q, k, v, o_ref, causal, scale, _ = _load(name)
    o = orc.flash_tiled_f32(q, k, v, causal=causal, scale=scale)
    tol = 2e-5 if "spike" not in name else 1e-4
    assert np.abs(o.astype(np.float64) - o_ref).max() < tol


def test_lse():
    q, k, v, _, _, _, _ = _load("bh2_n96_d64_full")
    for causal in (False, True):
        o_c, lse_c = orc.attention_f64(q, k, v, causal=causal, scale=0.5, return_lse=True)
        o_n, lse_n = orc.attention_numpy(q, k, v, causal=causal, scale=0.5, return_lse=True)
        assert np.abs(lse_c - lse_n).max() < 1e-10
        assert np.abs(o_c - o_n).max() < 1e-12


def test_packed_layout_against_reference_vector():
    z = np.load(os.path.join(GOLDEN_DIR, "llmc_packed_b2_t96_c128_nh2.npz"))
    inp, out_ref, nh = z["inp"], z["out"], int(z["n_head"])
    out = orc.attention_packed_f32(inp, nh)
    assert np.abs(out - out_ref).max() <= 1e-6  # same loop order as attention_forward_cpu; observed 0.0
    # and the packed layout is the plain (BH, N, d) causal op with scale 1/sqrt(hs) after a head split
    q, k, v = orc.split_packed_qkv(inp, nh)
    o = orc.attention_f64(q, k, v, causal=True, scale=1.0 / np.sqrt(q.shape[-1]))
    assert np.abs(orc.merge_heads(o, inp.shape[0], nh) - out_ref).max() < 1e-5


@pytest.mark.skipif(not orc.have_reference_build(), reason="oracle/_ref not built (reference tree absent)")
def test_packed_layout_against_live_reference_build():
    rng = np.random.default_rng(11)
    inp = (rng.random((1, 40, 3 * 64), dtype=np.float32) * 2 - 1).astype(np.float32)
    ref = orc.reference_attention_packed_f32(inp, 2)
    assert np.abs(orc.attention_packed_f32(inp, 2) - ref).max() <= 1e-6


def test_known_answer_iota_ones():
    # test.cu:615-631: Q = K = iota, V = 1  ->  every output element is exactly 1 (softmax rows sum to 1)
    bh, n, d = 2, 64, 32
    q = np.arange(bh * n * d, dtype=np.float32).reshape(bh, n, d)
    v = np.ones_like(q)
    for causal in (False, True):
        o = orc.attention_f32(q, q, v, causal=causal)
        assert np.all(o == 1.0)
        assert np.all(orc.flash_tiled_f32(q, q, v, causal=causal) == 1.0)


def test_bf16_helpers_round_trip():
    rng = np.random.default_rng(3)
    x = rng.standard_normal(1000).astype(np.float32)
    r = orc.round_to_bf16(x)
    assert np.all(orc.bf16_bits_to_f32(orc.bf16_bits(x)) == r)
    assert np.abs(r - x).max() <= np.abs(x).max() * 2.0 ** -8
    assert np.all(orc.round_to_bf16(r) == r)


def test_linearity_in_v_and_shift_invariance():
    # size-independent properties used at full size on the GPU: O is linear in V; adding a constant to all scores
    # of a row (here: shifting k along q's direction is not constant, so use q scaling trick) leaves softmax unchanged
    rng = np.random.default_rng(5)
    q, k, v1, v2 = (rng.standard_normal((1, 48, 32)).astype(np.float32) for _ in range(4))
    a = orc.attention_f64(q, k, v1) * 2.0 + orc.attention_f64(q, k, v2) * -0.5
    b = orc.attention_f64(q, k, (2.0 * v1 - 0.5 * v2).astype(np.float32))
    assert np.abs(a - b).max() < 1e-6
