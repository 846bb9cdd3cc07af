"""GPU (-m gpu): worst-case inputs, not seeded random ones ("the reference validates whatever tensor it just ran",
bench_flashattention.py:74-79 -- so must this suite).

THE CONTRACT OF THE fp32 DEFAULT (plain ``fa.forward(q32, k32, v32)``; include/flashattn_amd.h states the same):

  * on every input,  |O - O_fp64| <= max(1e-3, E_ref(O))  and  |LSE - LSE_fp64| <= max(1e-3, E_ref(LSE)),
    where E_ref is the error the REFERENCE'S OWN ARITHMETIC leaves on that input: fp32 operands through a k-ordered chain of rounding fp32
    FMAs (flashattention.cu:236-252; tests/adversarial.py: fma_chain_logits).  In words: inside the north star's 1e-3 wherever the
    reference kernel itself is, and never further from fp64 than the reference kernel's own arithmetic where it is not;
  * on the coherent-rounding family of tests/adversarial.py (constant-component rows, few-valued rows, a broadcast token, quantised +
    offset inputs, two dominant keys with v = +-5, many near-equal keys) at logit widths up to 89.5 -- the family VERDICT r04 built against
    the round-4 default, which read 3e-2 .. 6e-2 on it -- 1e-3 OUTRIGHT, O and LSE, d in {32, 64, 128}, causal or not: observed <= 2.7e-4.
    kernel="exact" (the reference's arithmetic, measured) reads 2.6e-3 / 5.9e-3 there;
  * how: K and Q' as two FP16 terms (22 bits; the operand error is below the FMA chain's own rounding bound for d >= 12), the hi.hi products
    first (the matrix core truncates every partial sum at the magnitude it has then), and KEY CENTRING: the kernel works on k_j - kbar,
    kbar the coordinate-wise median of three keys of the share, so that a magnitude all keys share never enters a rounded sum -- the
    softmax only needs differences, and one fp32 subtraction of nearly equal values is exact (csrc/fa_split_kernel.h).

The bf16-P kernels (bf16 tensors, bf16 out) are held to the bound derived from the data: every softmax weight off by 2^-8 with the worst
signs (tests/adversarial.py: p_rounding_bound) -- not to a typical value of seeded data."""
import os
import sys

import numpy as np
import pytest
import torch

import flashattention_c_amd as fa

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import adversarial as adv  # noqa: E402
import soak_fuzz  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-3
REF_SHARE = 1.0     # of the reference arithmetic's own error, where that exceeds 1e-3: the general contract (range cases, outlier keys)
OBSERVED = []


def dev():
    return torch.device("cuda", 0)


def record(what, err, tol):
    OBSERVED.append((what, err, tol))
    assert err < tol, f"{what}: {err:.3e} >= {tol:.3e}"


@pytest.fixture(scope="module", autouse=True)
def _dump_observed():
    yield
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_observed_adversarial.txt"), "w") as f:
            for what, e, tol in OBSERVED:
                f.write(f"{e:.3e}  tol {tol:.1e}  ratio {e / tol:.2f}  {what}\n")


def run(q, k, v, kernel="auto", causal=False, scale=1.0):
    o, lse = fa.forward(torch.from_numpy(q).to(dev()), torch.from_numpy(k).to(dev()), torch.from_numpy(v).to(dev()), causal, scale=scale,
                        kernel=kernel, return_lse=True)
    return o.cpu().numpy().astype(np.float64), lse.cpu().numpy().astype(np.float64)


# ---------------------------------------------------------------------------------------------------------------
# fp32 tensors through the DEFAULT call on the coherent-rounding family
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("width", [60.0, 85.0, 89.5])
@pytest.mark.parametrize("family", adv.FAMILIES)
@pytest.mark.parametrize("d", [32, 64, 128])
def test_fp32_default_on_coherent_rounding_family(d, family, width):
    """constant-component rows, few-valued rows, a broadcast token, quantised + offset inputs, two dominant keys with
    v = +-5 and many near-equal keys, at the logit widths the round-4 guard let through.  Plain fa.forward(q, k, v): O and LSE."""
    for causal in (False, True):
        q, k, v = adv.make(family, d, width, n=512, bh=2, seed=d)
        o_ref, l_ref = adv.attention_f64(q, k, v, causal)
        ce_o, ce_l = adv.reference_arithmetic_error(q, k, v, causal)
        o, lse = run(q, k, v, causal=causal)
        record(f"fp32 default, {family} d={d} w={width} causal={int(causal)}: O (fp32 FMA chain: {ce_o:.1e})", float(np.abs(o - o_ref).max()), TOL)
        record(f"fp32 default, {family} d={d} w={width} causal={int(causal)}: LSE (fp32 FMA chain: {ce_l:.1e})", float(np.abs(lse - l_ref).max()), TOL)


@pytest.mark.parametrize("d", [32, 64, 128])
def test_the_two_cases_of_verdict_r04(d):
    """The two inputs VERDICT r04 constructed by emulation against the round-4 default (guard quantity 89.8 < 90, O error 5.7e-2; 2048 near-
    equal constant-component keys, LSE error 4.2e-3), the first one scaled to d = 32 / 128 at the same guard quantity."""
    a = 2.5387 * np.sqrt(64.0 / d)
    q = np.full((1, 64, d), a, np.float32)
    k = np.zeros((1, 64, d), np.float32)
    k[:, 0::2] = 4.42197
    k[:, 1::2] = 4.42147
    v = np.zeros((1, 64, d), np.float32)
    v[:, 0::2] = 5.0
    v[:, 1::2] = -5.0
    assert 85.0 < adv._width(q, k) < 90.0
    o_ref, l_ref = adv.attention_f64(q, k, v)
    ce_o, ce_l = adv.reference_arithmetic_error(q, k, v)
    o, lse = run(q, k, v)
    record(f"VERDICT r04 case 1 at d={d}: O (fp32 FMA chain: {ce_o:.1e})", float(np.abs(o - o_ref).max()), TOL)
    record(f"VERDICT r04 case 1 at d={d}: LSE", float(np.abs(lse - l_ref).max()), TOL)
    rng = np.random.default_rng(d)
    q = np.full((1, 2048, d), a, np.float32)
    k = ((4.42 + rng.uniform(-1e-3, 1e-3, (1, 2048, 1))) * np.ones((1, 1, d))).astype(np.float32)
    v = (rng.standard_normal((1, 2048, d)) * 2).astype(np.float32)
    o_ref, l_ref = adv.attention_f64(q, k, v)
    ce_o, ce_l = adv.reference_arithmetic_error(q, k, v)
    o, lse = run(q, k, v)
    record(f"VERDICT r04 case 2 at d={d}: O", float(np.abs(o - o_ref).max()), TOL)
    record(f"VERDICT r04 case 2 at d={d}: LSE (fp32 FMA chain: {ce_l:.1e})", float(np.abs(lse - l_ref).max()), TOL)


@pytest.mark.parametrize("d", [64, 128])
def test_exact_kernel_is_the_reference_arithmetic_not_an_oracle(d):
    """kernel="exact" (v_mfma_f32_32x32x2_f32: a k-ordered fp32 FMA chain) reproduces the reference kernel's OWN rounding on coherent inputs:
    its error against fp64 is the emulated FMA chain's (both ~2e-3 .. 6e-3 on the two-dominant-key rows at width 89.5), 40 .. 60 times the
    default's (which works on centred keys).  Documented here so that nobody mistakes "exact" for "error free" again (VERDICT r04 assumed it immune)."""
    q, k, v = adv.make("const_two_keys", d, 89.5, n=512, bh=2, seed=d)
    o_ref, _ = adv.attention_f64(q, k, v)
    ce_o, _ = adv.reference_arithmetic_error(q, k, v)
    e_exact = float(np.abs(run(q, k, v, "exact")[0] - o_ref).max())
    e_auto = float(np.abs(run(q, k, v, "auto")[0] - o_ref).max())
    OBSERVED.append((f"exact kernel on const_two_keys d={d} w=89.5 (fp32 FMA chain emulated: {ce_o:.1e}; default: {e_auto:.1e})", e_exact, 2.0 * ce_o + 1e-4))
    assert 0.3 * ce_o < e_exact < 2.0 * ce_o + 1e-4, (e_exact, ce_o)
    assert e_auto < 0.2 * e_exact, (e_auto, e_exact)


# ---------------------------------------------------------------------------------------------------------------
# fp32 tensors: key centring (k_j - kbar, kbar = the coordinate-wise median of three keys of the share)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [32, 64, 128])
def test_key_centring_is_robust_to_outlier_reference_candidates(d):
    """The reference key is the median of keys 0, n/2 and n - 1 of the share: ONE outlier among them (an attention sink in position 0, a huge
    last key) is rejected, and two outliers only make the centring as good as no centring (|kbar_c| never exceeds the second largest of three
    actual values).  Asserted on random data with outliers planted in exactly those positions -- scaled up AND anti-aligned with the queries'
    mean so that the rows' mass stays on the ordinary keys, whose logits a bad reference would spoil -- at the fp32 bar, O and LSE."""
    rng = np.random.default_rng(d)
    bh, n = 3, 1000
    q = rng.standard_normal((bh, n, d)).astype(np.float32) + 0.5
    k = rng.standard_normal((bh, n, d)).astype(np.float32)
    v = rng.standard_normal((bh, n, d)).astype(np.float32)
    sink = (-30.0 * q.mean(axis=1) / np.linalg.norm(q.mean(axis=1), axis=-1, keepdims=True)).astype(np.float32)
    for planted in ([0], [n - 1], [0, n - 1], [0, n // 2, n - 1]):
        kk = k.copy()
        for j in planted:
            kk[:, j] = sink * (1.0 + 0.01 * j / n)
        for causal in (False, True):
            o_ref, l_ref = adv.attention_f64(q, kk, v, causal)
            ce_o, ce_l = adv.reference_arithmetic_error(q, kk, v, causal)
            o, lse = run(q, kk, v, causal=causal)
            record(f"key centring, outliers at {planted}, d={d}, causal={int(causal)}: O", float(np.abs(o - o_ref).max()), max(TOL, REF_SHARE * ce_o))
            record(f"key centring, outliers at {planted}, d={d}, causal={int(causal)}: LSE", float(np.abs(lse - l_ref).max()), max(TOL, REF_SHARE * ce_l))


@pytest.mark.parametrize("n", [500, 512, 8192])
def test_key_centring_removes_common_mode_and_keeps_the_guard_quiet(n):
    """Keys with a large common offset (k = 64 + N(0, 1) at d = 128): UNCENTRED they would trip the range guard (D max|k| alone is 8700 > 8192)
    and carry a row constant of several hundred into every partial sum; the guard looks at the centred keys, so the default stays on the
    16-bit pipes (route 1) -- for a ragged length too (rows past the end are kept out of the guard's max |k|), and over key shares
    (n = 8192 on one slab: every share centres on its own reference and adds its own row constant back to its log-sum-exp before the
    combine) -- and reads far below the reference arithmetic's own error.  Then the constant-component family at width 205."""
    rng = np.random.default_rng(n)
    q = rng.standard_normal((1, n, 128)).astype(np.float32)
    k = (64.0 + rng.standard_normal((1, n, 128))).astype(np.float32)
    v = rng.standard_normal((1, n, 128)).astype(np.float32)
    rows = np.unique(np.concatenate([[0, 1, 2, n - 1], rng.integers(0, n, 200)]))
    for causal in (False, True):
        o, lse = run(q, k, v, causal=causal)
        assert fa.last_forward_route() == 1, (n, causal)
        o_ref, l_ref = adv.rows_f64(q[0], k[0], v[0], rows, causal)
        o_ch, l_ch = adv.rows_f64(q[0], k[0], v[0], rows, causal, chain=True)
        record(f"key centring, k = 64 + N(0,1), d=128 n={n} causal={int(causal)}: O (fp32 FMA chain: {np.abs(o_ch - o_ref).max():.1e})",
               float(np.abs(o[0][rows] - o_ref).max()), TOL)
        record(f"key centring, k = 64 + N(0,1), d=128 n={n} causal={int(causal)}: LSE (fp32 FMA chain: {np.abs(l_ch - l_ref).max():.1e})",
               float(np.abs(lse[0][rows] - l_ref).max()), TOL)
    q, k, v = adv.make("const_two_keys", 128, 205.0, n=n, bh=1, seed=n)
    for causal in (False, True):
        o, lse = run(q, k, v, causal=causal)
        assert fa.last_forward_route() == 1, (n, causal)
        o_ref, l_ref = adv.rows_f64(q[0], k[0], v[0], rows, causal)
        o_ch, l_ch = adv.rows_f64(q[0], k[0], v[0], rows, causal, chain=True)
        ce_o = float(np.abs(o_ch - o_ref).max())
        e_o, e_l = float(np.abs(o[0][rows] - o_ref).max()), float(np.abs(lse[0][rows] - l_ref).max())
        record(f"key centring, const_two_keys d=128 w=205 n={n} causal={int(causal)}: O (fp32 FMA chain: {ce_o:.1e})", e_o, TOL)
        record(f"key centring, const_two_keys d=128 w=205 n={n} causal={int(causal)}: LSE", e_l, TOL)
        assert e_o < 0.2 * ce_o, (e_o, ce_o)


@pytest.mark.parametrize("d", [32, 64, 128])
def test_value_centring_makes_the_pv_terms_relative_to_the_spread_of_v(d):
    """V = offset + N(0, 1) under a peaked softmax (scale 1: a row's weight sits on a few keys, nothing averages out).  The P.V terms are
    bf16 hi + lo (16 bits): uncentred they cost 3 * 2^-17 * max|v| -- 1.6e-3 at offset 100, 1.6e-2 at 1000, ten times what the reference's
    own fp32 recurrence leaves (profiles/r05_v_offset.txt) --; the kernel splits v_j - vbar (vbar: the median of three rows) and adds vbar
    back, so what remains is the spread's 3 * 2^-17 * max|v - vbar| plus the output's own fp32 rounding at the offset's magnitude.  Also a V
    that is constant over the slab (all zeros after centring: the redo path) and an outlier row in a reference position."""
    rng = np.random.default_rng(d)
    bh, n = 2, 700
    q, k, x = (rng.standard_normal((bh, n, d)).astype(np.float32) for _ in range(3))
    for off in (0.0, 100.0, 1000.0, -3000.0):
        for causal in (False, True):
            v = (x + np.float32(off)).astype(np.float32)
            o_ref, _ = adv.attention_f64(q, k, v, causal)
            o, _ = run(q, k, v, causal=causal)
            tol = 2e-4 + 2.0 ** -22 * abs(off)       # (fp32 itself: half an ulp of |O| ~ |off| is 2^-24 |off|)
            record(f"value centring, V = {off:g} + N(0,1), d={d}, causal={int(causal)}", float(np.abs(o - o_ref).max()), tol)
    v = np.full((bh, n, d), 1.25, np.float32)
    o, _ = run(q, k, v)
    assert np.abs(o - 1.25).max() == 0.0, "a constant V must come back exactly"
    v = (x + np.float32(500.0)).astype(np.float32)
    v[:, 0] = -4.0e4                                  # an outlier in a reference position: rejected by the median
    o_ref, _ = adv.attention_f64(q, k, v)
    o, _ = run(q, k, v)
    sc = np.einsum("bqd,bkd->bqk", q.astype(np.float64), k.astype(np.float64))
    w0 = (np.exp(sc - sc.max(-1, keepdims=True)) / np.exp(sc - sc.max(-1, keepdims=True)).sum(-1, keepdims=True))[..., 0]   # weight on key 0
    # a row pays for the outlier in proportion to the weight it puts on it (its 16-bit terms are relative to |v_0 - vbar| ~ 4e4), and for
    # the ordinary rows as if the outlier were not there
    # WORST CASE, and attained (0.95 of it at d = 64: over 1400 rows x 64 columns some weight and some value sit at the bottom of their
    # binades): hi = bf16(x) to nearest leaves <= 2^-8 |x|, lo = bf16(x - hi) leaves <= 2^-16 |x| -- for the weight and for the value --,
    # and the lo.lo product the kernel drops is <= 2^-16 |p v|: 3 * 2^-16 = 6 * 2^-17 of w_0 |v_0 - vbar|; + fp32 accumulation at that magnitude
    tol_rows = 2e-4 + 2.0 ** -22 * 500 + w0 * (6 * 2.0 ** -17 + 2.0 ** -21) * 4.05e4
    ratio = float((np.abs(o - o_ref).max(axis=-1) / tol_rows).max())
    record(f"value centring, outlier V row 0, d={d}: worst row error / its own WORST-CASE bound (attainable)", ratio, 1.0)


# ---------------------------------------------------------------------------------------------------------------
# fp32 tensors: the RANGE of the fp16 operand terms (what the guard of FA_KERNEL_AUTO bounds since round 5)
# ---------------------------------------------------------------------------------------------------------------
def test_fp32_default_outside_the_fp16_range():
    """|x| >= 65520 (hi = inf -> NaN scores -> the workgroup's rows in fp32 arithmetic), operands so unbalanced that the small side's lo terms
    are fp16 subnormals (q 2^-13, k 2^13: the unguarded split products read ~1e-3, the guarded default hands over), tiny and huge magnitudes
    that fit: every case inside 1e-3 through the default call, and the route says which arithmetic produced it."""
    rng = np.random.default_rng(5)
    g = lambda *s: rng.standard_normal(s).astype(np.float32)  # noqa: E731
    cases = [
        ("q 2^-13, k 2^13", g(2, 512, 64) * 2.0 ** -13, g(2, 512, 64) * 2.0 ** 13, 1.0, 2),
        ("q 2^13, k 2^-13", g(2, 512, 64) * 2.0 ** 13, g(2, 512, 64) * 2.0 ** -13, 1.0, 2),
        ("one 7e4 element in k", g(2, 512, 64), np.where(np.arange(2 * 512 * 64).reshape(2, 512, 64) == 777, 7e4, g(2, 512, 64)).astype(np.float32), 1e-4, 2),
        ("k 1e-30 (uniform weights)", g(2, 512, 64), g(2, 512, 64) * 1e-30, 1.0, 1),
        ("q, k 1e-6", g(2, 512, 64) * 1e-6, g(2, 512, 64) * 1e-6, 1.0, 1),
        ("q x 6 at d = 32 (wide logits, inside the range budget)", g(2, 512, 32) * 6, g(2, 512, 32), 1.0, 1),
    ]
    for name, q, k, scale, want_route in cases:
        v = g(*q.shape)
        o_ref, l_ref = adv.attention_f64(q, k, v, scale=scale)
        o, lse = run(q, k, v, scale=scale)
        route = fa.last_forward_route()
        assert route == want_route, (name, route)
        ce_o, ce_l = adv.reference_arithmetic_error(q, k, v, scale=scale)
        record(f"fp32 default, range: {name}: O", float(np.abs(o - o_ref).max()), max(TOL, REF_SHARE * ce_o))
        record(f"fp32 default, range: {name}: LSE", float(np.abs(lse - l_ref).max()), max(TOL, REF_SHARE * ce_l))
    # premise of the guard: the unguarded split products ARE wrong on the unbalanced operands (else the guard could go)
    q, k, v = g(2, 512, 64) * 2.0 ** -13, g(2, 512, 64) * 2.0 ** 13, g(2, 512, 64)
    o_ref, _ = adv.attention_f64(q, k, v)
    assert float(np.abs(run(q, k, v, "split")[0] - o_ref).max()) > 10 * float(np.abs(run(q, k, v, "auto")[0] - o_ref).max())


# ---------------------------------------------------------------------------------------------------------------
# bf16-P kernels: the tolerance as a BOUND derived from the data
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("out_f32", [True, False])
@pytest.mark.parametrize("d", [32, 64, 128])
def test_bf16_p_kernels_against_the_worst_case_of_two_dominant_keys(d, out_f32):
    """two comparably dominant keys with v1 = -v2 = max|v|; bf16 rounds the two weights by up to 2^-8 each, in the worst
    case with opposite signs: |O error| <= 2^-8 * sum_j w_j |v_j - O| = 1/4 * 2^-7 * |v1 - v2| for equal weights (+ 2^-8 |O| for a bf16
    output).  Many rows sweep the difference of the two logits finely, so the weights land all over their bf16 intervals; asserted: every
    row inside the bound computed from the data, and the construction gets within a factor of three of it (it IS the worst case)."""
    n, vmax = 4096, 5.4
    e = np.zeros(d, np.float32)
    e[0] = 1.0
    # Row r: q = a_r e0, so every logit is a_r * 1.44 * k_j0 (exp2 domain).  Key 0 (v = 0) is a little above the two that matter and sets the
    # row's exponent reference (a reference key's weight is an exact power of two: with only two keys one of them would never be rounded);
    # keys 1 / 2 carry v = +-vmax, 2^-5 and 2^-5 + 2^-7 below it: as a_r sweeps 8 .. 40 both weights walk across several bf16 binades at
    # different speeds, so some row finds them rounded by almost 2^-8 each with opposite signs.  Every other key: 20+ binades below.
    a = (8.0 + 32.0 * np.arange(n) / n).astype(np.float32)
    q = (a[:, None] * e[None, :])[None].astype(np.float32)
    k = np.zeros((1, n, d), np.float32)
    k[0, :, 0] = -8.0
    k[0, 0, 0] = 1.0
    k[0, 1, 0] = 1.0 - 2.0 ** -5
    k[0, 2, 0] = 1.0 - 2.0 ** -5 - 2.0 ** -7
    v = np.zeros((1, n, d), np.float32)
    v[0, 1] = vmax
    v[0, 2] = -vmax
    qb, kb, vb = (torch.from_numpy(t).to(torch.bfloat16).to(dev()) for t in (q, k, v))
    q32, k32, v32 = (t.float().cpu().numpy() for t in (qb, kb, vb))
    rows = np.arange(n)
    bound, o_ref = adv.p_rounding_bound(q32[0], k32[0], v32[0], rows, False, 1.0, rel=2.0 ** -8 + 2.0 ** -10)
    if not out_f32:
        bound += 2.0 ** -8 * float(np.abs(o_ref).max())
    bound += 1e-4
    o = fa.forward(qb, kb, vb, False, kernel="mfma", out_dtype=torch.float32 if out_f32 else None).float().cpu().numpy().astype(np.float64)[0]
    err = float(np.abs(o - o_ref).max())
    record(f"bf16 P, two dominant keys v = +-{vmax}, d={d}, {'fp32' if out_f32 else 'bf16'} out (data-derived bound {bound:.2e})", err, bound)
    assert err > bound / 4.0, f"the construction no longer comes near the worst case: {err:.3e} of {bound:.3e}"
    # the accurate P (FA_KERNEL_AUTO for an fp32 output) on the same rows: the fp32 bar
    if out_f32:
        oa = fa.forward(qb, kb, vb, False, out_dtype=torch.float32).cpu().numpy().astype(np.float64)[0]
        record(f"two bf16 terms of P on the same rows, d={d}", float(np.abs(oa - o_ref).max()), 2e-4)


@pytest.mark.parametrize("bh,n,d,causal", [(16, 8192, 64, False), (16, 8192, 64, True), (128, 1024, 64, False), (4, 2048, 128, False), (8, 1000, 32, True)])
def test_bf16_p_kernels_inside_their_data_derived_bound_on_random_data(bh, n, d, causal):
    """The regression thresholds of tests/test_gpu_parity.py (1.2e-2 / 2.5e-2 at scale 1) are typical values of N(0,1) data; the statement
    that holds for ANY data is the bound computed from it.  Here: the BASELINE shapes, sampled rows of two slabs, bf16 and fp32 output."""
    g = torch.Generator().manual_seed(n + d)
    qb, kb, vb = (torch.randn(bh, n, d, generator=g).to(torch.bfloat16).to(dev()) for _ in range(3))
    rng = np.random.default_rng(n)
    for out_f32 in (True, False):
        o = fa.forward(qb, kb, vb, causal, kernel="mfma", out_dtype=torch.float32 if out_f32 else None)
        for sb in (0, bh - 1):
            rows = np.unique(np.concatenate([[0, n - 1], rng.integers(0, n, 48)]))
            q32, k32, v32 = (t[sb].float().cpu().numpy() for t in (qb, kb, vb))
            bound, o_ref = adv.p_rounding_bound(q32, k32, v32, rows, causal, 1.0, rel=2.0 ** -8 + 2.0 ** -10)
            bound += (2.0 ** -8 * float(np.abs(o_ref).max()) if not out_f32 else 0.0) + 1e-4
            err = float(np.abs(o[sb].float().cpu().numpy().astype(np.float64)[rows] - o_ref).max())
            record(f"bf16 P vs its data-derived bound, {bh}x{n}x{d} causal={int(causal)} slab {sb} {'fp32' if out_f32 else 'bf16'} out", err, bound)


# ---------------------------------------------------------------------------------------------------------------
# a bounded slice of the soak generator, fp64 oracle on sampled rows (tests/soak_fuzz.py)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [501, 502])
def test_soak_slice(seed):
    """60 random cases per seed (shapes around the tiling boundaries, eleven data families incl. the coherent ones, three head dims, causal
    or not, three scales, fp32 / bf16 tensors, the packed-QKV entry): every output against the fp64 oracle on sampled rows and against
    rung 0 on every slab.  The by-hand soak (thousands of cases) runs the same code."""
    worst, routes = soak_fuzz.run(cases=60, seed=seed, max_n=4500, verbose=False)
    for key, (err, tol, desc) in worst.items():
        OBSERVED.append((f"soak slice seed {seed}: {key} ({desc})", err, tol))
    assert sum(routes.values()) == 120
