"""GPU (-m gpu): the HIP path, called through the C ABI, against the CPU oracle and the committed golden vectors.

Tolerances (max-abs, stated once here; BASELINE.md section 4 gives the arithmetic behind them):
  fp32 kernels                                  1e-3  north_star bar; observed <= 3e-5 (exact) / <= 2.5e-4 (split) at scale 1
  bf16 kernel, fp32 out, scale 1/8, long rows   1e-3  north_star bar with 1/sqrt(d) scaling; observed 4e-4 (non-causal, N >= 1000)
  bf16 kernel, fp32 out, scale 1/8, short rows  4e-3  rows that attend to few keys (causal head of the sequence, N < 1000) keep the
                                                      full 2^-9 relative rounding of each bf16 P value un-averaged; observed <= 2.7e-3
  bf16 kernel, fp32 out, scale 1.0              1.2e-2 unscaled scores: P is near one-hot, so the error tends to 2^-9 * max|v| (one bf16
                                                      rounding of the dominant P); max|v| ~ 5.4 over 8M randn -> 1.05e-2; observed <= 9.0e-3
  bf16 kernel, bf16 out                         2.5e-2 adds half a bf16 ulp of |O| (|O| < 4 -> 7.8e-3); observed <= 1.5e-2
The bf16 kernel is always compared with the oracle evaluated on the SAME bf16-valued inputs.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

import flashattention_c_amd as fa
from flashattention_c_amd import _cabi
from oracle import oracle as orc
from tests.conftest import GOLDEN_DIR, golden_cases

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-3
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev():
    return torch.device("cuda", 0)


def to_dev(*arrs, dtype=torch.float32):
    return [torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(dev()) for a in arrs]


def max_err(t: torch.Tensor, ref: np.ndarray) -> float:
    got = t.detach().float().cpu().numpy().astype(np.float64)
    assert not np.isnan(got).any(), "NaN in output (unwritten or invalid element)"
    return float(np.abs(got - ref).max())


def check(t: torch.Tensor, ref: np.ndarray, tol: float, what: str = ""):
    e = max_err(t, ref)
    OBSERVED.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split("::")[-1].replace(" (call)", "") + " " + what, e, tol))
    assert e < tol, f"max abs err {e:.3e} >= tol {tol:.1e} {what}"


OBSERVED = []


@pytest.fixture(scope="module", autouse=True)
def _dump_observed_errors():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_observed.txt"), "w") as f:
            for what, e, tol in OBSERVED:
                f.write(f"{e:.3e}  tol {tol:.1e}  {what}\n")


def bf16_tol(scale: float, out_f32: bool, causal: bool = False, n: int = 1 << 20) -> float:
    if not out_f32:
        return 2.5e-2
    if scale >= 0.5:
        return 1.2e-2
    return 1e-3 if (not causal and n >= 1000) else 4e-3


def randn(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)).numpy()


def test_extension_is_the_in_tree_library():
    # the product path is the hipcc-built library next to the package, not a fallback
    assert os.path.samefile(_cabi.LIB_PATH, os.path.join(ROOT, "flashattention.c_amd", "libflashattn_amd.so"))
    assert _cabi.lib().fa_device_count() >= 1


# ---------------------------------------------------------------------------------------------------------------
# golden vectors (outputs of the reference's own oracle code, tests/golden)
# ---------------------------------------------------------------------------------------------------------------
# fp32 tensors: "auto" = the split kernel (bf16 matrix pipe, three products of two-term splits), "exact" = fp32 MFMA arithmetic
@pytest.mark.parametrize("kernel", ["auto", "exact", "naive"])
@pytest.mark.parametrize("name", golden_cases())
def test_fp32_against_golden(name, kernel):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    q, k, v = to_dev(z["q"], z["k"], z["v"])
    o = fa.forward(q, k, v, bool(z["causal"]), scale=float(z["scale"]), kernel=kernel)
    check(o, z["o"], TOL_F32)


@pytest.mark.parametrize("out_f32", [False, True])
@pytest.mark.parametrize("name", [c for c in golden_cases() if "bf16vals" in c])
def test_bf16_against_golden(name, out_f32):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    q, k, v = to_dev(z["q"], z["k"], z["v"], dtype=torch.bfloat16)  # exactly representable: no input rounding
    scale = float(z["scale"])
    o = fa.forward(q, k, v, bool(z["causal"]), scale=scale, out_dtype=torch.float32 if out_f32 else None)
    assert o.dtype == (torch.float32 if out_f32 else torch.bfloat16)
    check(o, z["o"], bf16_tol(scale, out_f32, bool(z["causal"]), q.shape[1]))


def test_packed_qkv_against_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "llmc_packed_b2_t96_c128_nh2.npz"))
    (inp,) = to_dev(z["inp"])
    out = fa.forward_packed_qkv(inp, int(z["n_head"]))
    # the reference validates this path at 1e-4 (src/llm.c/attention_forward.cu:1262)
    check(out, z["out"].astype(np.float64), 1e-4)


# ---------------------------------------------------------------------------------------------------------------
# seeded inputs vs the oracle: shapes, raggedness, causal, head dims
# ---------------------------------------------------------------------------------------------------------------
SHAPES = [
    # bh, n, d
    (4, 1024, 64),   # README shape family (c2), fewer slabs
    (2, 1000, 64),   # ragged tail in the last tile
    (3, 333, 32),    # d = 32 (README rows 2 and 4)
    (2, 257, 128),   # d = 128, one row past a workgroup boundary
    (1, 1, 64),      # single token
    (5, 31, 64),     # shorter than one MFMA block
    (2, 129, 64),    # one row past the 128-row q tile
]


@pytest.mark.parametrize("kernel", ["auto", "exact"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("bh,n,d", SHAPES)
def test_fp32_vs_oracle(bh, n, d, causal, kernel):
    q, k, v = (randn(s, bh, n, d) for s in (1, 2, 3))
    ref = orc.attention_f64(q, k, v, causal=causal, scale=1.0)
    o = fa.forward(*to_dev(q, k, v), causal, kernel=kernel)
    check(o, ref, TOL_F32)


# every tiling of the split kernel: 1 / 2 = one / two 32-row blocks per wave (first-tile reference), 3 / 4 = the software-
# pipelined reference-free pass; ragged length (700 = 21 tiles + 28 keys), both scales, LSE as well
@pytest.mark.parametrize("d,mode", [(64, 1), (64, 2), (64, 3), (64, 4), (128, 1), (128, 3), (128, 5), (32, 1), (32, 2), (32, 3), (32, 4)])
@pytest.mark.parametrize("causal", [False, True])
def test_split_kernel_tilings(d, mode, causal):
    q, k, v = (randn(s, 2, 700, d) for s in (1, 2, 3))
    for scale in (1.0, 0.125):
        ref, lse_ref = orc.attention_f64(q, k, v, causal=causal, scale=scale, return_lse=True)
        o, lse = fa.forward(*to_dev(q, k, v), causal, scale=scale, kernel=f"split:{mode}", return_lse=True)
        check(o, ref, TOL_F32, f"split:{mode} scale {scale}")
        check(lse, lse_ref, TOL_F32, f"split:{mode} lse scale {scale}")
    # the hi/lo splits are exact to 2^-16: 16-bit inputs give the fp32-exact kernel's answer to its own rounding
    qb, kb, vb = (orc.round_to_bf16(t) for t in (q, k, v))
    ref = orc.attention_f64(qb, kb, vb, causal=causal, scale=0.125)
    check(fa.forward(*to_dev(qb, kb, vb), causal, scale=0.125, kernel=f"split:{mode}"), ref, 2e-5, "bf16-valued inputs")


@pytest.mark.parametrize("d,mode", [(64, 0), (64, 2), (64, 3), (64, 4), (128, 0), (128, 1), (128, 3), (128, 5), (32, 0), (32, 1), (32, 3), (32, 4), (64, 1)])
@pytest.mark.parametrize("causal", [False, True])
def test_split_kernel_redo_outside_the_optimistic_range(d, mode, causal):
    """Rows whose scores leave the range the optimistic pass can prove (exp2-domain row sums outside 2^-100 .. 2^100, or a
    row growing by more than 2^100 past its first tile): the workgroup must redo its tile with the running maximum, for
    single rows, for whole blocks, in the middle of the sequence and in the first tile."""
    bh, n = 2, 1536
    q, k, v = (randn(s, bh, n, d) for s in (41, 42, 43))
    q *= np.sqrt(64.0 / d)
    unit = lambda x: x / np.linalg.norm(x, axis=-1, keepdims=True)
    for r, key, gain in ((3, 700, 14.0), (40, 701, 16.0), (200, 1100, 12.0), (1300, 900, 15.0), (1301, 650, 18.0), (1535, 333, 6.0), (70, 9, 13.0)):
        k[:, key] = gain * unit(q[:, r])           # score ~ 8 gain  ->  up to 2^200 in the exp2 domain
    k[0, 800] = 20.0 * unit(q[0, 64:96].mean(axis=0))
    q[1, 500] *= -4.0                               # a whole row of wide scores (sigma 32): its sum leaves the proven range
    ref, lse_ref = orc.attention_f64(q, k, v, causal=causal, return_lse=True)
    o, lse = fa.forward(*to_dev(q, k, v), causal, kernel=f"split:{mode}", return_lse=True)
    check(o, ref, 3e-3, f"split:{mode}")           # errors scale with |q||k| (2^-17 relative per product): 3x the bar here
    check(lse, lse_ref, 3e-3, f"split:{mode} lse")


@pytest.mark.parametrize("scale", [1.0, 0.125])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("bh,n,d", SHAPES)
def test_bf16_vs_oracle(bh, n, d, causal, scale):
    q, k, v = (orc.round_to_bf16(randn(s, bh, n, d)) for s in (4, 5, 6))
    ref = orc.attention_f64(q, k, v, causal=causal, scale=scale)
    qd, kd, vd = to_dev(q, k, v, dtype=torch.bfloat16)
    check(fa.forward(qd, kd, vd, causal, scale=scale, out_dtype=torch.float32), ref, bf16_tol(scale, True, causal, n), "f32-out")
    check(fa.forward(qd, kd, vd, causal, scale=scale), ref, bf16_tol(scale, False), "bf16-out")


# 0 = product dispatch, 1 = phase-structured kernel, 7 / 24 = pipelined kernel with 4- / 2-wave workgroups (optimistic mix with
# verified redo), 25 / 26 = the same with the lazily rescaled mix only, 30 / 31 = one-wave-per-SIMD 128-rows-per-wave kernel
# (barrier every 2 / every stage), 42 = that kernel with the lazily rescaled mix only, 50 / 51 / 52 = one-wave-per-SIMD kernel
# with 64 rows per wave (optimistic, barrier every 2 / every stage; rescaled mix only)
@pytest.mark.parametrize("variant", [0, 1, 7, 24, 25, 26, 30, 31, 42, 50, 51, 52])
@pytest.mark.parametrize("causal", [False, True])
def test_bf16_tiling_variants_agree(variant, causal):
    q, k, v = (orc.round_to_bf16(randn(s, 3, 700, 64)) for s in (7, 8, 9))
    ref = orc.attention_f64(q, k, v, causal=causal, scale=0.125)
    qd, kd, vd = to_dev(q, k, v, dtype=torch.bfloat16)
    o = fa.forward(qd, kd, vd, causal, scale=0.125, kernel=f"mfma:{variant}", out_dtype=torch.float32)
    check(o, ref, bf16_tol(0.125, True, causal, 700))


def test_lse_output():
    q, k, v = (randn(s, 2, 300, 64) for s in (10, 11, 12))
    for causal in (False, True):
        _, lse_ref = orc.attention_f64(q, k, v, causal=causal, scale=0.25, return_lse=True)
        for kern in ("auto", "exact", "naive"):
            _, lse = fa.forward(*to_dev(q, k, v), causal, scale=0.25, return_lse=True, kernel=kern)
            check(lse, lse_ref, 1e-3)
        qb, kb, vb = (orc.round_to_bf16(t) for t in (q, k, v))
        _, lse_ref_b = orc.attention_f64(qb, kb, vb, causal=causal, scale=0.25, return_lse=True)
        _, lse_b = fa.forward(*to_dev(qb, kb, vb, dtype=torch.bfloat16), causal, scale=0.25, return_lse=True)
        # the bf16 kernel sums the bf16-rounded P on the matrix core (same values as the numerator): each term carries
        # 2^-9 relative rounding, so log(l) is good to a few 1e-3
        check(lse_b, lse_ref_b, 5e-3)


def test_known_answer_iota_ones():
    """test.cu:615-631 workload (Q = K = iota, V = 1): O must be exactly 1; scores reach ~1e12, exercising overflow safety."""
    bh, n, d = 2, 512, 64
    q = torch.arange(bh * n * d, dtype=torch.float32).reshape(bh, n, d).to(dev())
    v = torch.ones_like(q)
    for causal in (False, True):
        assert torch.all(fa.forward(q, q, v, causal) == 1.0)
        ob = fa.forward(q.bfloat16(), q.bfloat16(), v.bfloat16(), causal)
        assert torch.all(ob.float() == 1.0)


def test_forced_rescale_spike():
    """A key that beats the running max by hundreds, late in the sequence, and one early (T13 / rule 26 input)."""
    bh, n, d = 2, 2048, 64
    q, k, v = (randn(s, bh, n, d) for s in (13, 14, 15))
    k[0, 1900] = 6.0 * q[0, 37]
    k[1, 5] = 4.0 * q[1, 1500]
    for causal in (False, True):
        ref = orc.attention_f64(q, k, v, causal=causal)
        check(fa.forward(*to_dev(q, k, v), causal), ref, TOL_F32)
        qb, kb, vb = (orc.round_to_bf16(t) for t in (q, k, v))
        refb = orc.attention_f64(qb, kb, vb, causal=causal)
        check(fa.forward(*to_dev(qb, kb, vb, dtype=torch.bfloat16), causal, out_dtype=torch.float32), refb, 1.2e-2)


@pytest.mark.parametrize("out_f32", [False, True])
@pytest.mark.parametrize("n", [512, 1024, 1000])
def test_dominant_key_in_the_last_keys(n, out_f32):
    """Rows whose maximum sits in the last few keys: the running-max update happens in the kernel's tail code, right
    behind the final K.Q^T MFMAs (regression: an asm v_max3 there read the accumulator before the MFMA had retired, the
    rescale was skipped and the dominant key was clamped -- one instantiation only, ~1 row in 65 000 on random data)."""
    bh, d = 3, 64
    q, k, v = (randn(s, bh, n, d) for s in (31, 32, 33))
    for r, key, gain in ((5, n - 1, 1.2), (77, n - 2, 0.9), (300, n - 9, 1.5), (n - 1, n - 17, 1.0), (200, n - 33, 1.1)):
        k[:, key] = gain * q[:, r] / np.linalg.norm(q[:, r], axis=-1, keepdims=True) * 8.0
    qb, kb, vb = (orc.round_to_bf16(t) for t in (q, k, v))
    for causal in (False, True):
        check(fa.forward(*to_dev(q, k, v), causal), orc.attention_f64(q, k, v, causal=causal), TOL_F32, "fp32")
        refb = orc.attention_f64(qb, kb, vb, causal=causal)
        ob = fa.forward(*to_dev(qb, kb, vb, dtype=torch.bfloat16), causal, out_dtype=torch.float32 if out_f32 else None)
        check(ob, refb, bf16_tol(1.0, out_f32), "bf16")
        _, lse_ref = orc.attention_f64(qb, kb, vb, causal=causal, return_lse=True)
        _, lse = fa.forward(*to_dev(qb, kb, vb, dtype=torch.bfloat16), causal, return_lse=True)
        check(lse, lse_ref, 2e-2, "bf16 lse")


# d = 128: 0 = product dispatch (too few workgroups here: phase-structured kernels), 10 = 4-waves/SIMD diet kernel, 50 / 51 =
# one-wave-per-SIMD kernel (optimistic mix with verified redo; barrier every 2 / every stage), 52 = its lazily rescaled mix only
@pytest.mark.parametrize("variant", [0, 10, 50, 51, 52])
@pytest.mark.parametrize("causal", [False, True])
def test_bf16_d128_tiling_variants_agree(variant, causal):
    _tiling_variant_case(128, variant, causal)


# d = 32: 0 = product dispatch (one-wave-per-SIMD kernel), 1 = phase-structured, 7 / 24 = pipelined two-wave kernel, 50 / 52 =
# one-wave-per-SIMD kernel (optimistic / lazily rescaled mix)
@pytest.mark.parametrize("variant", [0, 1, 7, 24, 50, 52])
@pytest.mark.parametrize("causal", [False, True])
def test_bf16_d32_tiling_variants_agree(variant, causal):
    _tiling_variant_case(32, variant, causal)


def _tiling_variant_case(d, variant, causal):
    q, k, v = (orc.round_to_bf16(randn(s, 3, 700, d)) for s in (17, 18, 19))
    ref = orc.attention_f64(q, k, v, causal=causal, scale=0.125)
    qd, kd, vd = to_dev(q, k, v, dtype=torch.bfloat16)
    o = fa.forward(qd, kd, vd, causal, scale=0.125, kernel=f"mfma:{variant}", out_dtype=torch.float32)
    check(o, ref, bf16_tol(0.125, True, causal, 700))
    check(fa.forward(qd, kd, vd, causal, scale=0.125, kernel=f"mfma:{variant}"), ref, bf16_tol(0.125, False))


@pytest.mark.parametrize("d,variant", [(64, 0), (64, 7), (64, 24), (64, 25), (64, 26), (64, 30), (64, 42), (64, 50), (64, 52), (128, 50), (128, 52),
                                       (32, 50), (32, 52), (32, 7)])
@pytest.mark.parametrize("causal", [False, True])
def test_rescale_inside_the_pipelined_loop(d, variant, causal):
    """Keys that outgrow a row's first-sub-tile maximum by 2^140 .. 2^230, placed in the middle of the sequence.  Lazily
    rescaled mix (variants 25, 26, 42 and every redo): the rare rescale branch of the software-pipelined main loop has
    to fire, for single rows, for a whole 32-row block and for neighbouring blocks of one wave, and everything already
    accumulated at the old reference has to be scaled exactly once.  Optimistic mix: growth below 2^200 must come out
    right without any rescale (the LSE exposes a clamped or saturated P that O / l would hide), growth above it must
    fail the end-of-tile verification and be redone."""
    bh, n = 2, 1536
    q, k, v = (randn(s, bh, n, d) for s in (41, 42, 43))
    q *= np.sqrt(64.0 / d)                         # |q| ~ 8 at either head dim (the gains below are tuned to that)
    unit = lambda x: x / np.linalg.norm(x, axis=-1, keepdims=True)
    for r, key, gain in ((3, 700, 14.0), (40, 701, 16.0), (200, 1100, 12.0), (1300, 900, 15.0), (1301, 650, 18.0), (1535, 333, 13.0)):
        k[:, key] = gain * unit(q[:, r])           # score ~ gain * |q| ~ 8 gain  ->  > 64 / log2(e) above the crowd
    k[0, 800] = 20.0 * unit(q[0, 64:96].mean(axis=0))  # one key that lifts a whole 32-row block at once
    qb, kb, vb = (orc.round_to_bf16(t) for t in (q, k, v))
    ref = orc.attention_f64(qb, kb, vb, causal=causal)
    o = fa.forward(*to_dev(qb, kb, vb, dtype=torch.bfloat16), causal, kernel=f"mfma:{variant}", out_dtype=torch.float32)
    check(o, ref, bf16_tol(1.0, True), f"variant {variant}")
    _, lse_ref = orc.attention_f64(qb, kb, vb, causal=causal, return_lse=True)
    _, lse = fa.forward(*to_dev(qb, kb, vb, dtype=torch.bfloat16), causal, kernel=f"mfma:{variant}", return_lse=True)
    check(lse, lse_ref, 2e-2, f"lse variant {variant}")


def test_fuzz_shapes_through_the_dispatch_against_rung0():
    """Random (bh, n, d, causal, scale) through the product dispatch -- every kernel family and the ragged / tiny / one-round /
    many-round branches of choose_bf16() get hit -- against the rung-0 kernel on the same bf16-valued inputs."""
    rng = np.random.default_rng(2024)
    worst = 0.0
    for case in range(48):
        d = int(rng.choice([32, 64, 128]))
        bh = int(rng.integers(1, 41))
        n = int(rng.choice([rng.integers(1, 130), rng.integers(130, 1100), rng.integers(1100, 3000)]))
        causal = bool(rng.integers(0, 2))
        scale = float(rng.choice([1.0, 0.5, d ** -0.5]))
        g = torch.Generator(device="cpu").manual_seed(1000 + case)
        q, k, v = (torch.randn(bh, n, d, generator=g).to(torch.bfloat16).to(dev()) for _ in range(3))
        ref = fa.forward(q.float(), k.float(), v.float(), causal, scale=scale, kernel="naive")
        out = fa.forward(q, k, v, causal, scale=scale, out_dtype=torch.float32)
        assert not torch.isnan(out).any(), f"NaN: case {case} bh={bh} n={n} d={d} causal={causal}"
        err = float((out - ref).abs().max())
        worst = max(worst, err)
        assert err < bf16_tol(1.0, True), f"case {case} bh={bh} n={n} d={d} causal={causal} scale={scale}: {err:.3e}"
    OBSERVED.append(("fuzz through dispatch, worst of 48", worst, bf16_tol(1.0, True)))


# bf16 tensors through the split machinery (kernel="split"): K and V are exact in one bf16 term, Q*scale*log2e and P are carried
# as hi + lo -- the bf16 path that meets the 1e-3 bar of the north star at scale 1 (the fast kernels round P to 8 bits: 5e-3)
@pytest.mark.parametrize("d,mode", [(64, 0), (64, 1), (64, 3), (64, 4), (128, 0), (128, 1), (128, 3), (128, 5), (32, 0), (32, 1), (32, 3), (32, 4)])
@pytest.mark.parametrize("causal", [False, True])
def test_bf16_tensors_accurate_mode(d, mode, causal):
    q, k, v = (orc.round_to_bf16(randn(s, 2, 700, d)) for s in (1, 2, 3))
    qd, kd, vd = to_dev(q, k, v, dtype=torch.bfloat16)
    for scale in (1.0, 0.125):
        ref, lse_ref = orc.attention_f64(q, k, v, causal=causal, scale=scale, return_lse=True)
        o, lse = fa.forward(qd, kd, vd, causal, scale=scale, kernel=f"split:{mode}", out_dtype=torch.float32, return_lse=True)
        check(o, ref, TOL_F32, f"bf16 split:{mode} scale {scale}")          # observed <= 1.3e-4 at scale 1
        check(lse, lse_ref, TOL_F32, f"bf16 split:{mode} lse scale {scale}")
        ob = fa.forward(qd, kd, vd, causal, scale=scale, kernel=f"split:{mode}")   # bf16 output: its own rounding only
        check(ob, ref, bf16_tol(scale, False), f"bf16 split:{mode} bf16 out")


@pytest.mark.parametrize("d,mode", [(64, 0), (64, 1), (64, 3), (128, 0), (128, 3), (32, 0)])
@pytest.mark.parametrize("causal", [False, True])
def test_bf16_tensors_accurate_mode_redo(d, mode, causal):
    """The accurate bf16 mode outside the optimistic range: same inputs as the fp32 redo test, bf16-valued."""
    bh, n = 2, 1536
    q, k, v = (randn(s, bh, n, d) for s in (41, 42, 43))
    q *= np.sqrt(64.0 / d)
    unit = lambda x: x / np.linalg.norm(x, axis=-1, keepdims=True)
    for r, key, gain in ((3, 700, 14.0), (40, 701, 16.0), (200, 1100, 12.0), (1300, 900, 15.0), (1301, 650, 18.0), (70, 9, 13.0)):
        k[:, key] = gain * unit(q[:, r])
    q[1, 500] *= -4.0
    qb, kb, vb = (orc.round_to_bf16(t) for t in (q, k, v))
    ref, lse_ref = orc.attention_f64(qb, kb, vb, causal=causal, return_lse=True)
    o, lse = fa.forward(*to_dev(qb, kb, vb, dtype=torch.bfloat16), causal, kernel=f"split:{mode}", out_dtype=torch.float32, return_lse=True)
    check(o, ref, 3e-3, f"bf16 split:{mode}")
    check(lse, lse_ref, 3e-3, f"bf16 split:{mode} lse")


def test_fuzz_fp32_shapes_through_the_dispatch_against_rung0():
    """fp32 tensors through FA_KERNEL_AUTO (the split kernel and its per-shape tiling choice) against the rung-0 fp32
    kernel on random shapes, including grids that switch between the tilings, ragged lengths and short causal rows."""
    rng = np.random.default_rng(4321)
    worst = 0.0
    for case in range(40):
        d = int(rng.choice([32, 64, 128]))
        bh = int(rng.integers(1, 41))
        n = int(rng.choice([rng.integers(1, 130), rng.integers(130, 1100), rng.integers(1100, 4500)]))
        causal = bool(rng.integers(0, 2))
        scale = float(rng.choice([1.0, 0.5, d ** -0.5]))
        g = torch.Generator(device="cpu").manual_seed(2000 + case)
        q, k, v = (torch.randn(bh, n, d, generator=g).to(dev()) for _ in range(3))
        ref = fa.forward(q, k, v, causal, scale=scale, kernel="naive")
        out = fa.forward(q, k, v, causal, scale=scale)
        assert not torch.isnan(out).any(), f"NaN: case {case} bh={bh} n={n} d={d} causal={causal}"
        err = float((out - ref).abs().max())
        worst = max(worst, err)
        assert err < TOL_F32, f"case {case} bh={bh} n={n} d={d} causal={causal} scale={scale}: {err:.3e}"
    OBSERVED.append(("fp32 fuzz through dispatch, worst of 40", worst, TOL_F32))


def test_graph_replay_timing_entry():
    q, k, v = (torch.randn(4, 512, 64, device=dev(), dtype=torch.bfloat16) for _ in range(3))
    ms_stream = fa.time_forward(q, k, v, False, warmup=1, iters=5)
    ms_graph = fa.time_forward(q, k, v, False, warmup=1, iters=5, graph=True)
    assert 0.0 < ms_graph < 50.0 and 0.0 < ms_stream < 50.0


def test_transpose_detecting_structured_input():
    """Asymmetric, structured Q/K/V: a swapped row/col map in any MFMA fragment or a transposed V changes the answer."""
    bh, n, d = 1, 192, 64
    r = np.arange(n, dtype=np.float32)[:, None]
    c = np.arange(d, dtype=np.float32)[None, :]
    q = (0.02 * r - 0.05 * c + 0.001 * r * c / d)[None].astype(np.float32) * 0.1
    k = (0.03 * np.sin(0.1 * r) + 0.04 * np.cos(0.3 * c) + 0.002 * c)[None].astype(np.float32)
    v = (r / n - 2.0 * c / d + 0.01 * r * c / (n * d) * 7)[None].astype(np.float32)
    for causal in (False, True):
        ref = orc.attention_f64(q, k, v, causal=causal)
        check(fa.forward(*to_dev(q, k, v), causal), ref, 1e-4)
        qb, kb, vb = (orc.round_to_bf16(t) for t in (q, k, v))
        refb = orc.attention_f64(qb, kb, vb, causal=causal)
        check(fa.forward(*to_dev(qb, kb, vb, dtype=torch.bfloat16), causal, out_dtype=torch.float32), refb, 5e-3)


def test_packed_qkv_vs_oracle_random():
    B, T, C, NH = 2, 300, 256, 4  # hs = 64, ragged T
    inp = (np.random.default_rng(21).random((B, T, 3 * C), dtype=np.float32) * 2 - 1).astype(np.float32)
    ref = orc.attention_packed_f32(inp, NH)
    out = fa.forward_packed_qkv(torch.from_numpy(inp).to(dev()), NH)
    check(out, ref.astype(np.float64), 1e-4)


def test_noncontiguous_and_out_argument():
    q, k, v = (randn(s, 2, 130, 64) for s in (16, 17, 18))
    ref = orc.attention_f64(q, k, v)
    qd, kd, vd = to_dev(q, k, v)
    qt = qd.transpose(0, 1).contiguous().transpose(0, 1)  # non-contiguous view of the same values
    assert not qt.is_contiguous()
    out = torch.full_like(qd, float("nan"))
    res = fa.forward(qt, kd, vd, False, out=out)
    assert res.data_ptr() == out.data_ptr()
    check(out, ref, TOL_F32)


def test_sharded_entry_point_on_one_device():
    """fa_forward_sharded with two shards that both live on device 0 (a 1-GPU box can still exercise the entry point)."""
    q, k, v = (randn(s, 5, 200, 64) for s in (19, 20, 21))
    ref = orc.attention_f64(q, k, v, causal=True)
    qd, kd, vd = to_dev(q, k, v)
    (b0, e0), (b1, e1) = fa.shard_range(5, 2, 0), fa.shard_range(5, 2, 1)
    outs = fa.forward_sharded([qd[b0:e0], qd[b1:e1]], [kd[b0:e0], kd[b1:e1]], [vd[b0:e0], vd[b1:e1]], True)
    torch.cuda.synchronize()
    check(torch.cat(outs), ref, TOL_F32)


def test_runs_on_callers_stream_without_sync():
    q, k, v = to_dev(*(randn(s, 4, 512, 64) for s in (22, 23, 24)))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        o = fa.forward(q, k, v, False)
    s.synchronize()
    ref = orc.attention_f64(q.cpu().numpy(), k.cpu().numpy(), v.cpu().numpy())
    check(o, ref, TOL_F32)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: exact oracle on sampled slabs + size-independent properties on the whole tensor
# ---------------------------------------------------------------------------------------------------------------
FULL = [
    ("c2", 128, 1024, 64, torch.float32),
    ("c3", 16, 8192, 64, torch.float32),
    ("c4", 16, 8192, 64, torch.bfloat16),
]


@pytest.mark.parametrize("name,bh,n,d,dtype", FULL)
def test_full_size_configs(name, bh, n, d, dtype):
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(bh, n, d, generator=g) for _ in range(3))
    if dtype == torch.bfloat16:
        q, k, v = (t.bfloat16() for t in (q, k, v))
    qd, kd, vd = q.to(dev()), k.to(dev()), v.to(dev())
    bf = dtype == torch.bfloat16
    kw = dict(out_dtype=torch.float32) if bf else {}
    tol = 1.2e-2 if bf else TOL_F32
    o = fa.forward(qd, kd, vd, False, **kw)
    # (a) exact oracle on two slabs (first and last)
    for s in (0, bh - 1):
        ref = orc.attention_f64(q[s:s + 1].float().numpy(), k[s:s + 1].float().numpy(), v[s:s + 1].float().numpy())
        check(o[s:s + 1], ref, tol)
    # (b) every slab against the rung-0 kernel on device (independent code path, fp32 on the same values)
    o_naive = fa.forward(qd.float(), kd.float(), vd.float(), False, kernel="naive")
    assert float((o.float() - o_naive).abs().max()) < tol
    # (c) V == 1  =>  O == 1: every softmax row sums to 1 (checks l, m, masking and the whole write-out).  Not bitwise:
    #     the numerator is summed by the matrix core (from bf16-rounded P on the bf16 path), the denominator by the VALU,
    #     in different orders over up to 8192 terms (observed 1.4e-5 in fp32 at N = 8192).
    ones = torch.ones_like(vd)
    assert float((fa.forward(qd, kd, ones, False, **kw) - 1.0).abs().max()) < (4e-3 if bf else 1e-4)
    # (d) linearity in V: O(q, k, 2 v1 - v2) == 2 O(q, k, v1) - O(q, k, v2)  (fp32 only; bf16 V rounding breaks exactness)
    if not bf:
        v2 = torch.randn(bh, n, d, generator=g).to(dev())
        lhs = fa.forward(qd, kd, 2.0 * vd - v2, False)
        rhs = 2.0 * o - fa.forward(qd, kd, v2, False)
        lin_err = float((lhs - rhs).abs().max())
        assert lin_err < 2e-4, f"linearity residual {lin_err:.3e}"
    # (e) causal: row 0 attends to key 0 only, so O[:, 0, :] == V[:, 0, :] (fp32: bitwise; bf16 path: the exponent of the
    #     row maximum is fma(m, c, -round(c*m)) = O(ulp), so p = 1 + O(1e-7) -- see fa_fwd_bf16.hip)
    oc = fa.forward(qd, kd, vd, True, **kw)
    if bf:
        assert float((oc[:, 0, :] - vd[:, 0, :].float()).abs().max()) < 1e-5
    else:
        # the split kernel carries V as hi + lo (16 significant bits); the exact kernel reproduces V bit for bit
        assert float((oc[:, 0, :] - vd[:, 0, :]).abs().max()) < 1e-4
        assert torch.equal(fa.forward(qd, kd, vd, True, kernel="exact")[:, 0, :], vd[:, 0, :])
    # (f) causal vs the oracle on one slab
    refc = orc.attention_f64(q[:1].float().numpy(), k[:1].float().numpy(), v[:1].float().numpy(), causal=True)
    check(oc[:1], refc, tol)


def test_c_driver_known_answer():
    """The torch-less driver (test.cu counterpart) on its iota/ones workload."""
    drv = os.path.join(ROOT, "flashattention.c_amd", "fa_driver")
    assert os.path.exists(drv), "fa_driver not built"
    for dtype in ("f32", "bf16"):
        r = subprocess.run([drv, "--mode", "kat", "--bh", "2", "--n", "1024", "--d", "64", "--dtype", dtype, "--causal", "1"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert '"pass": true' in r.stdout


@pytest.mark.parametrize("extra", [[], ["--masking"], ["--dtype", "bf16"]])
def test_bench_harness_reports_pass(extra):
    """The bench_flashattention.py counterpart: README shape family, verdict line at the stated tolerance."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "flashattention.c_amd", "harness", "bench_flashattention.py"),
                        "--batch_size", "2", "--seq_len", "1024", "--iters", "3"] + extra, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[Correctness] attn values sanity check: PASSED" in r.stdout
