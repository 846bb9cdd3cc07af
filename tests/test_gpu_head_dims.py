"""GPU: head dims outside {32, 64, 128} through the plain call (round 6), and the causal reference rows of the fp32 default
(ADVICE r05).

The reference is generic over ``d % 32 == 0`` by editing one macro (/root/reference/src/flashattention.cu:15, ``num_tiles = d / BK`` at :164);
here ``forward(q, k, v, causal)`` takes every head dim up to 256: the other multiples of 32 run the exact fp32 MFMA kernel (fa_fwd_f32_wide*.hip;
bf16 tensors widened on load), every other head dim the rung-0 kernel.  Everything is checked against the
fp64 oracle on the same inputs."""
import ctypes

import numpy as np
import pytest
import torch

import flashattention_c_amd as fa
from flashattention_c_amd import _cabi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-4   # fp32 arithmetic on unit-variance data (observed <= 3e-5)


def dev():
    return torch.device("cuda", 0)


def rand(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def err(t, ref):
    got = t.detach().float().cpu().numpy().astype(np.float64)
    assert not np.isnan(got).any()
    return float(np.abs(got - ref).max())


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [96, 160, 192, 224, 256])
def test_exact_fp32_kernel_at_the_other_multiples_of_32(d, causal):
    """AUTO and kernel="exact" alike: v_mfma_f32_32x32x2_f32 for both contractions; ragged lengths, single tile, LSE."""
    assert _cabi.lib().fa_kernel_name_for(_cabi.FA_DTYPE_F32, d, int(causal), 3, 333) == b"fa_fwd_f32_kernel"
    for n in (1, 31, 333, 1024):
        q, k, v = (rand(100 * d + n + i, 3, n, d) for i in range(3))
        ref, lref = orc.attention_f64(q.numpy(), k.numpy(), v.numpy(), causal=causal, scale=1.0, return_lse=True)
        o, lse = fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), causal, return_lse=True)
        assert err(o, ref) < TOL and err(lse, lref) < TOL, (d, n, causal)
        assert fa.last_forward_route() == 0
        oe = fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), causal, kernel="exact", scale=0.125)
        assert err(oe, orc.attention_f64(q.numpy(), k.numpy(), v.numpy(), causal=causal, scale=0.125)) < TOL


@pytest.mark.parametrize("d", [96, 256])
def test_exact_fp32_kernel_wide_heads_long_rows_and_key_shares(d):
    """Rows of 8192 keys: a full grid (16 slabs) against the rung-0 kernel on every slab, and an idle grid (one slab) whose rows run over key
    shares + combine (fa_workspace_bytes > 0), causal and not."""
    for bh, causal in ((16, False), (1, False), (1, True), (4, True)):
        n = 8192 if bh <= 4 else 2048
        q, k, v = (torch.randn(bh, n, d, device=dev(), generator=torch.Generator(device=dev()).manual_seed(7 * bh + d)) for _ in range(3))
        if bh == 1:
            assert fa.workspace_bytes(bh, n, d, causal) > 0
        o = fa.forward(q, k, v, causal)
        ref = fa.forward(q, k, v, causal, kernel="naive")
        assert float((o - ref).abs().max()) < 2 * TOL, (bh, causal, d)
        rows = [0, 1, n // 2, n - 1]
        r64 = orc.attention_f64(q[:1].cpu().numpy(), k[:1].cpu().numpy(), v[:1].cpu().numpy(), causal=causal, scale=1.0)
        assert float(np.abs(o[0, rows].cpu().numpy().astype(np.float64) - r64[0, rows]).max()) < TOL


@pytest.mark.parametrize("d", [1, 8, 48, 80, 100, 200, 255])
def test_any_other_head_dim_runs_on_the_rung_0_kernel(d):
    assert _cabi.lib().fa_kernel_name_for(_cabi.FA_DTYPE_F32, d, 0, 2, 300) == b"fa_naive_f32_kernel"
    for causal in (False, True):
        q, k, v = (rand(d + i, 2, 300, d) for i in range(3))
        ref, lref = orc.attention_f64(q.numpy(), k.numpy(), v.numpy(), causal=causal, scale=1.0, return_lse=True)
        o, lse = fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), causal, return_lse=True)
        assert err(o, ref) < TOL and err(lse, lref) < TOL, (d, causal)
    with pytest.raises(_cabi.FlashAttnError) as ei:   # an explicit family that exists at 32 / 64 / 128 only says so
        fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), False, kernel="split")
    assert ei.value.code == 2
    with pytest.raises(ValueError):
        fa.forward(torch.zeros(1, 4, 300, device=dev()), torch.zeros(1, 4, 300, device=dev()), torch.zeros(1, 4, 300, device=dev()), False)


@pytest.mark.parametrize("d", [48, 96, 256])
def test_bf16_tensors_outside_the_instantiated_head_dims(d):
    """bf16 tensors: fp32 arithmetic -- the exact MFMA kernel at multiples of 32 (widened on load), the rung-0 kernel elsewhere --, bf16 or fp32
    output (the explicit kernel="naive" takes bf16 tensors at 64 too)."""
    assert _cabi.lib().fa_kernel_name_for(_cabi.FA_DTYPE_BF16, d, 1, 2, 300) == (b"fa_naive_f32_kernel" if d % 32 else b"fa_fwd_f32_kernel")
    q, k, v = (rand(3 * d + i, 2, 300, d).to(torch.bfloat16) for i in range(3))
    for causal in (False, True):
        ref = orc.attention_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), causal=causal, scale=1.0)
        ob = fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), causal)
        assert ob.dtype == torch.bfloat16 and err(ob, ref) < 2.0 ** -8 * max(1.0, float(np.abs(ref).max()))
        of = fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), causal, out_dtype=torch.float32)
        assert of.dtype == torch.float32 and err(of, ref) < TOL
    q64, k64, v64 = (rand(9 + i, 2, 200, 64).to(torch.bfloat16) for i in range(3))
    o64 = fa.forward(q64.to(dev()), k64.to(dev()), v64.to(dev()), True, kernel="naive", out_dtype=torch.float32)
    assert err(o64, orc.attention_f64(q64.float().numpy(), k64.float().numpy(), v64.float().numpy(), causal=True, scale=1.0)) < TOL


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("d", [96, 160, 192, 224, 256])
def test_bf16_tensors_on_the_exact_kernel_at_the_wide_head_dims(d, causal):
    """Round 6: bf16 tensors at the other multiples of 32 ran on the rung-0 kernel (16 x 8192 x 96: 246 ms where fp32 tensors took 3.3).  Now the
    exact fp32 MFMA kernel widens them on their way into its fp32 LDS images: ragged lengths, a single key, both output types, the LSE; AUTO
    and kernel="exact" are the same launch."""
    for n in (1, 31, 333, 1024, 1300):
        q, k, v = (rand(7 * d + n + i, 3, n, d).to(torch.bfloat16) for i in range(3))
        ref, lref = orc.attention_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), causal=causal, scale=1.0, return_lse=True)
        of, lse = fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), causal, out_dtype=torch.float32, return_lse=True)
        assert of.dtype == torch.float32 and err(of, ref) < TOL and err(lse, lref) < TOL, (d, n, causal)
        ob = fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), causal)
        assert ob.dtype == torch.bfloat16 and err(ob, ref) < 2.0 ** -8 * max(1.0, float(np.abs(ref).max())), (d, n, causal)
        oe = fa.forward(q.to(dev()), k.to(dev()), v.to(dev()), causal, kernel="exact", scale=0.125, out_dtype=torch.float32)
        assert err(oe, orc.attention_f64(q.float().numpy(), k.float().numpy(), v.float().numpy(), causal=causal, scale=0.125)) < TOL


@pytest.mark.parametrize("d", [96, 256])
def test_bf16_tensors_wide_heads_long_rows_and_key_shares(d):
    """Rows of 8192 keys: a full grid against the rung-0 kernel on every slab, idle grids over key shares + combine (fp32 partials, the combine
    stores the caller's output type), causal pairs."""
    for bh, causal in ((16, False), (1, False), (1, True), (4, True)):
        n = 8192 if bh <= 4 else 2048
        q, k, v = (torch.randn(bh, n, d, device=dev(), generator=torch.Generator(device=dev()).manual_seed(5 * bh + d)).to(torch.bfloat16) for _ in range(3))
        if bh == 1:
            assert fa.workspace_bytes(bh, n, d, causal, dtype=torch.bfloat16) > 0
        ref = fa.forward(q, k, v, causal, kernel="naive", out_dtype=torch.float32)
        of = fa.forward(q, k, v, causal, out_dtype=torch.float32)
        assert float((of - ref).abs().max()) < 2 * TOL, (bh, causal, d)
        ob = fa.forward(q, k, v, causal)
        assert ob.dtype == torch.bfloat16 and float((ob.float() - ref).abs().max()) < 2.0 ** -8 * max(1.0, float(ref.abs().max()))
        rows = [0, 1, n // 2, n - 1]
        r64 = orc.attention_f64(q[:1].float().cpu().numpy(), k[:1].float().cpu().numpy(), v[:1].float().cpu().numpy(), causal=causal, scale=1.0)
        assert float(np.abs(of[0, rows].cpu().numpy().astype(np.float64) - r64[0, rows]).max()) < TOL


def test_packed_qkv_entry_at_a_head_size_of_96():
    """fa_forward_packed_qkv (llm.c layout, attention_forward.cu:1106-1179) at C / NH = 96: the exact kernel through the strided addressing,
    against the llm.c CPU restatement of the oracle at the harness' own 1e-4 (attention_forward.cu:1262)."""
    B, T, NH, hs = 2, 130, 3, 96
    C = NH * hs
    inp = rand(5, B, T, 3 * C) * 0.5
    out = fa.forward_packed_qkv(inp.to(dev()), NH)
    ref = orc.attention_packed_f32(inp.numpy(), NH)      # the restatement of attention_forward_cpu (attention_forward.cu:53-125)
    assert err(out, ref) < 1e-4


@pytest.mark.parametrize("d", [32, 64, 128])
def test_causal_reference_rows_come_from_keys_the_tile_attends_to(d):
    """the centring references of the fp32 default (kbar, vbar: medians of three rows) were rows 0, n / 2, n - 1 of the share --
    for most causal tiles future tokens.  Two outliers among them (inf / 1e30 garbage in padding, |v| ~ 6e4) became the reference of every
    value the tile does attend to: ~0.5 absolute error in O with the guard quiet.  Now a tile takes the three rows below its own horizon: rows
    that attend to none of the outliers are as accurate as without them."""
    n = 1024
    q, k, v = (rand(11 * d + i, 2, n, d) for i in range(3))
    clean = orc.attention_f64(q.numpy(), k.numpy(), v.numpy(), causal=True, scale=1.0)
    for bad in (6.0e4, 1.0e30, float("inf")):
        v2, k2 = v.clone(), k.clone()
        v2[:, n // 2] = bad
        v2[:, n - 1] = -bad if np.isfinite(bad) else bad
        k2[:, n // 2] *= 50.0          # (outlier keys in the same rows: the key reference must not come from there either)
        k2[:, n - 1] *= 50.0
        o, lse = fa.forward(q.to(dev()), k2.to(dev()), v2.to(dev()), True, return_lse=True)
        got = o.cpu().numpy().astype(np.float64)[:, : n // 2]          # rows whose horizon ends below the first outlier
        assert np.isfinite(got).all()
        assert float(np.abs(got - clean[:, : n // 2]).max()) < 2e-4, (d, bad)
    # non-causal, every row attends to the outliers: the result is dominated by them and only has to be what fp32 arithmetic gives
    v3 = v.clone()
    v3[:, n // 2] = 6.0e4
    o = fa.forward(q.to(dev()), k.to(dev()), v3.to(dev()), False)
    ex = fa.forward(q.to(dev()), k.to(dev()), v3.to(dev()), False, kernel="exact")
    ref = orc.attention_f64(q.numpy(), k.numpy(), v3.numpy(), causal=False, scale=1.0)
    bound = 3 * 2.0 ** -17 * 1.2e5 + 1e-3                                # the header's contract: max(1e-3, E_ref) + 3 * 2^-17 * max|v - vbar|
    assert err(o, ref) < bound and err(ex, ref) < bound


def test_null_workspace_runs_every_kernel_choice_unsplit():
    """fa_forward_ws(kernel = FA_KERNEL_MFMA, fp32, idle grid, workspace = NULL) answered FA_ERR_INVALID_ARGUMENT under ABI 5
    although the exact kernel runs fine unsplit."""
    L = _cabi.lib()
    q, k, v = (torch.randn(1, 4096, 64, device=dev()) for _ in range(3))
    o = torch.zeros_like(q)
    ref = fa.forward(q, k, v, False, kernel="naive")
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for kern in (_cabi.FA_KERNEL_MFMA, _cabi.FA_KERNEL_AUTO, _cabi.FA_KERNEL_SPLIT):
        o.zero_()
        assert L.fa_forward_ws(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), None, 1, 4096, 64, 1.0, 0, _cabi.FA_DTYPE_F32, kern, None, 0, s) == 0
        torch.cuda.synchronize()
        assert float((o - ref).abs().max()) < 1e-3
