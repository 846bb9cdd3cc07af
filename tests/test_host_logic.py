"""CPU: host-side logic -- C-ABI exports, argument validation (no compute without a GPU), sharding arithmetic."""
import ctypes
import os
import re

import pytest
import torch

import flashattention_c_amd as fa
from flashattention_c_amd import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "flashattn_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)   # prototypes only, not the prose
    declared = set(re.findall(r"\b(fa_[a-z_]+)\s*\(", hdr))
    assert declared == set(_cabi.EXPORTED_SYMBOLS), declared ^ set(_cabi.EXPORTED_SYMBOLS)
    L = _cabi.lib()
    for sym in declared:
        assert getattr(L, sym) is not None
    assert b"gfx950" in L.fa_version()
    assert L.fa_device_count() >= 0
    assert L.fa_kernel_name(_cabi.FA_DTYPE_BF16, 64, 0) == b"fa_fwd_bf16_x4_kernel"       # headline shape: 512-row workgroups
    assert L.fa_kernel_name(_cabi.FA_DTYPE_BF16, 64, 1) == b"fa_fwd_bf16_x2_kernel"        # causal, few items: 256-row tiles, one per CU
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 1, 128, 8192) == b"fa_fwd_bf16_x2_kernel"         # long causal rows: NB = 2 tiles whatever the grid
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 1, 128, 2048) == b"fa_fwd_bf16_pp3_kernel"        # short causal rows, many tiles: the two-wave kernel
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 3, 700) == b"fa_fwd_bf16_kernel"      # at most 128 tiles of 256 rows: 128-row workgroups (round 3)
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 48, 1024) == b"fa_fwd_bf16_x2_kernel"  # at most one round of 256-row tiles
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 24, 8192) == b"fa_fwd_bf16_x2_kernel"   # 1.5 rounds of 512-row tiles, long rows: NB = 2 (round 3)
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 48, 3072) == b"fa_fwd_bf16_pp3_kernel"  # ... rows below 4096 keys: the two-wave kernel
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 48, 4096) == b"fa_fwd_bf16_x2_kernel"   # (round 6: from 4096 keys on the re-centring kernel)
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 12, 8192) == b"fa_fwd_bf16_x4_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 128, 0, 16, 8192) == b"fa_fwd_bf16_x2_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 128, 0, 2, 300) == b"fa_fwd_bf16_w4_kernel"      # too few workgroups
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 128, 1, 2, 300) == b"fa_fwd_bf16_kernel"
    # head dims outside {32, 64, 128} (round 6): fp32 tensors at the other multiples of 32 up to 256 run the exact fp32 MFMA kernel, every
    # other head dim up to 256 the rung-0 kernel -- the reference compiles any d % 32 == 0 (flashattention.cu:15)
    assert L.fa_kernel_name(_cabi.FA_DTYPE_F32, 48, 0) == b"fa_naive_f32_kernel" and L.fa_kernel_name(_cabi.FA_DTYPE_F32, 300, 0) is None
    for d in (96, 160, 192, 224, 256):
        assert L.fa_kernel_name(_cabi.FA_DTYPE_F32, d, 1) == b"fa_fwd_f32_kernel" and L.fa_kernel_name(_cabi.FA_DTYPE_BF16, d, 0) == b"fa_fwd_f32_kernel"
    assert L.fa_kernel_name(7, 64, 0) is None and L.fa_kernel_name_for(_cabi.FA_DTYPE_F32, 64, 0, 0, 5) is None
    # bf16 tensors with fp32 output (round 4): P as two bf16 terms in one launch -- the tilings of the bf16-P one-wave-per-SIMD kernels
    assert L.fa_kernel_name(_cabi.FA_DTYPE_BF16_OUT_F32, 64, 0) == b"fa_fwd_bf16_x4_pb2_kernel"     # c4: 512-row workgroups
    assert L.fa_kernel_name(_cabi.FA_DTYPE_BF16_OUT_F32, 64, 1) == b"fa_fwd_bf16_x2_pb2_kernel"
    assert L.fa_kernel_name(_cabi.FA_DTYPE_BF16_OUT_F32, 128, 0) == b"fa_fwd_bf16_x2_pb2_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16_OUT_F32, 32, 0, 16, 8192) == b"fa_fwd_bf16_x2_pb2_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16_OUT_F32, 64, 0, 1, 8192) == b"fa_fwd_bf16_x2_pb2_kernel"   # idle grid: key-split launch of the NB = 2 kernel
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 1, 8192) == b"fa_fwd_bf16_x2_kernel"                 # idle grid: key-split launch
    # ... at every launch size (round 2 sent small launches to the hi + lo bf16 split kernel, whose 16-bit Q' is not data-independent)
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16_OUT_F32, 32, 0, 4, 300) == b"fa_fwd_bf16_x2_pb2_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16_OUT_F32, 64, 0, 16, 1024) == b"fa_fwd_bf16_x2_pb2_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16_OUT_F32, 64, 1, 128, 1024) == b"fa_fwd_bf16_x2_pb2_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16_OUT_F32, 128, 0, 2, 1 << 24) == b"fa_fwd_f32_split_kernel"    # a 4 GiB slab: beyond 32-bit byte offsets
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16_OUT_F32, 64, 0, 128, 8192) == b"fa_fwd_bf16_x4_pb2_kernel"   # c5's per-GPU shard
    # rows of two stages: the phase-structured kernel (round 3); longer rows keep the one-wave-per-SIMD kernels
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 1024, 128) == b"fa_fwd_bf16_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 32, 1, 1024, 100) == b"fa_fwd_bf16_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 128, 1, 256, 512) == b"fa_fwd_bf16_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 128, 0, 256, 512) == b"fa_fwd_bf16_x2_kernel"
    assert L.fa_kernel_name_for(_cabi.FA_DTYPE_BF16, 64, 0, 512, 256) == b"fa_fwd_bf16_pp3_kernel"


def test_kernel_ids_match_the_header_and_the_python_names():
    hdr = open(os.path.join(ROOT, "include", "flashattn_amd.h")).read()
    for name in ("FA_KERNEL_AUTO", "FA_KERNEL_NAIVE", "FA_KERNEL_MFMA", "FA_KERNEL_SPLIT", "FA_KERNEL_PB2", "FA_DTYPE_F32", "FA_DTYPE_BF16",
                 "FA_DTYPE_BF16_OUT_F32"):
        m = re.search(name + r"\s*=\s*(\d+)\b", hdr)
        assert m and int(m.group(1)) == getattr(_cabi, name), name
    from flashattention_c_amd import flash
    assert flash._kernel_id("split:4") == _cabi.FA_KERNEL_SPLIT | (4 << 8)
    assert flash._kernel_id("exact") == flash._kernel_id("mfma") == _cabi.FA_KERNEL_MFMA
    assert flash._kernel_id("auto") == _cabi.FA_KERNEL_AUTO
    assert flash._kernel_id("pb2") == _cabi.FA_KERNEL_PB2 == 6
    assert "FA_KERNEL_P16" not in hdr            # the retired fp16-P kernel ids (4, 5) are not part of the product header
    with pytest.raises(ValueError):
        flash._kernel_id("p16x2")
    with pytest.raises(ValueError):
        flash._kernel_id("fast")
    L = _cabi.lib()
    assert L.fa_kernel_name(_cabi.FA_DTYPE_F32, 64, 0) == b"fa_fwd_f32_split_kernel"   # fp32 tensors: the bf16 matrix pipe


def test_workspace_sizes_are_host_arithmetic():
    """fa_workspace_bytes needs no device: the plan of a forward (which launch chain, how much scratch) is decided on the host from the
    shape alone, and it is the same plan fa_forward_ws executes."""
    L = _cabi.lib()
    B16, B16F, F32 = _cabi.FA_DTYPE_BF16, _cabi.FA_DTYPE_BF16_OUT_F32, _cabi.FA_DTYPE_F32
    A = _cabi.FA_KERNEL_AUTO
    assert L.fa_workspace_bytes(16, 8192, 64, 0, F32, A) == 0                       # fp32 tensors: ONE launch, no scratch (ABI 6: no report-word header)
    assert L.fa_workspace_bytes(16, 8192, 64, 0, F32, _cabi.FA_KERNEL_SPLIT) == 0 and L.fa_workspace_bytes(16, 8192, 64, 0, F32, _cabi.FA_KERNEL_MFMA) == 0
    assert L.fa_workspace_bytes(16, 8192, 64, 0, B16, A) == 0                       # c4, bf16 output: one launch
    for shape in ((16, 8192, 64, 0), (16, 8192, 64, 1), (32, 1024, 64, 0), (16, 1024, 64, 1), (16, 1023, 64, 0), (128, 8192, 64, 0), (4, 300, 32, 0), (16, 8192, 128, 1)):
        assert L.fa_workspace_bytes(*shape, B16F, A) == 0                           # fp32 output (round 4): P as two bf16 terms -- one launch, no scratch
        assert L.fa_workspace_bytes(*shape, B16F, _cabi.FA_KERNEL_PB2) == 0
    part = lambda S, bh, n, d: S * bh * n * d * 4 + S * bh * n * 4
    assert L.fa_workspace_bytes(1, 8192, 64, 0, B16, A) == part(8, 1, 8192, 64)         # idle grid: key-split partials
    assert L.fa_workspace_bytes(2, 8192, 64, 0, B16, A) == part(4, 2, 8192, 64)
    assert L.fa_workspace_bytes(1, 8192, 64, 1, B16, A) == part(8, 1, 8192, 64)         # causal: shares of 1024 keys (multiples of the tile height)
    assert L.fa_workspace_bytes(8, 8192, 64, 1, B16, A) == part(2, 8, 8192, 64)         # causal: up to 256 tiles are split
    assert L.fa_workspace_bytes(8, 8192, 64, 0, B16, A) == 0 and L.fa_workspace_bytes(16, 8192, 64, 1, B16, A) == 0
    assert L.fa_workspace_bytes(1, 8192, 64, 0, F32, A) == part(8, 1, 8192, 64)         # fp32 tensors: the split kernel over key shares
    assert L.fa_workspace_bytes(1, 8192, 64, 1, F32, A) == part(8, 1, 8192, 64)         # ... causal too
    assert L.fa_workspace_bytes(16, 8192, 64, 1, F32, A) == 0
    # exact fp32 arithmetic (round 5): fewer than 256 tiles of 128 rows -> key shares until the launch has 256 .. 512 workgroups
    M = _cabi.FA_KERNEL_MFMA
    assert L.fa_workspace_bytes(1, 8192, 64, 0, F32, M) == part(8, 1, 8192, 64) and L.fa_workspace_bytes(2, 8192, 64, 0, F32, M) == part(4, 2, 8192, 64)
    assert L.fa_workspace_bytes(4, 8192, 64, 0, F32, M) == 0 and L.fa_workspace_bytes(4, 8192, 64, 1, F32, M) == part(4, 4, 8192, 64)   # (a causal round is still split)
    assert L.fa_workspace_bytes(16, 8192, 64, 0, F32, M) == 0 and L.fa_workspace_bytes(1, 1024, 64, 0, F32, M) == 0
    assert L.fa_workspace_bytes(1, 8192, 64, 0, F32, M | (1 << 8)) == 0                                 # an explicit tiling runs unsplit
    assert L.fa_workspace_bytes(1, 8192, 64, 0, B16F, A) == part(8, 1, 8192, 64)        # the accurate path splits idle grids the same way
    assert L.fa_workspace_bytes(1, 8192, 64, 0, B16F, _cabi.FA_KERNEL_PB2) == part(8, 1, 8192, 64)
    assert L.fa_workspace_bytes(1, 8192, 64, 0, B16, _cabi.FA_KERNEL_SPLIT) == 0
    # the two-term kernel has one tiling: short non-causal rows (1024 .. 4095 keys) on at most 64 tiles are key-split too (shares >= 256 keys)
    assert L.fa_workspace_bytes(16, 1024, 64, 0, B16F, A) == part(4, 16, 1024, 64)
    assert L.fa_workspace_bytes(8, 2048, 128, 0, B16F, _cabi.FA_KERNEL_PB2) == part(4, 8, 2048, 128)
    assert L.fa_workspace_bytes(1, 1024, 32, 0, B16F, A) == part(4, 1, 1024, 32)        # (shares of 256 keys at least)
    assert L.fa_workspace_bytes(16, 1024, 64, 0, B16, A) == 0                                  # bf16 P has finer tilings for these
    # the fp16-P kernels left the product library in round 4 (ablation library only): a size query answers 0, a forward FA_ERR_UNSUPPORTED
    assert L.fa_workspace_bytes(4, 300, 32, 0, B16F, 4) == 0 and L.fa_workspace_bytes(4, 300, 32, 0, B16F, 5) == 0
    # head dims outside {32, 64, 128}: the exact kernel splits idle grids like at 64; the rung-0 kernel needs nothing
    assert L.fa_workspace_bytes(1, 8192, 96, 0, F32, A) == part(8, 1, 8192, 96) == L.fa_workspace_bytes(1, 8192, 96, 0, F32, M)
    assert L.fa_workspace_bytes(16, 8192, 256, 0, F32, A) == 0 and L.fa_workspace_bytes(16, 8192, 48, 0, F32, A) == 0 and L.fa_workspace_bytes(1, 8192, 80, 0, B16, A) == 0
    assert L.fa_workspace_bytes(1, 8192, 96, 0, B16, A) == part(8, 1, 8192, 96)   # bf16 tensors on the exact kernel: the same fp32 partials
    # arguments fa_forward_ws would reject size to 0 and leave fa_last_error alone
    assert L.fa_workspace_bytes(0, 8192, 64, 0, B16F, A) == 0 and L.fa_workspace_bytes(16, 8192, 300, 0, B16F, A) == 0
    assert L.fa_workspace_bytes(16, 8192, 64, 0, 7, A) == 0 and L.fa_workspace_bytes(16, 8192, 64, 0, F32, _cabi.FA_KERNEL_PB2) == 0
    import torch
    assert fa.workspace_bytes(16, 8192, 64, dtype=torch.bfloat16, out_dtype=torch.float32) == 0
    assert fa.workspace_bytes(1, 8192, 64, dtype=torch.bfloat16, out_dtype=torch.float32) == part(8, 1, 8192, 64)


def test_fp32_auto_choice_follows_the_environment_switch():
    """FA_F32_AUTO=exact (read once per process) turns FA_KERNEL_AUTO for fp32 tensors into the fp32-arithmetic kernel."""
    import subprocess, sys
    code = ("from flashattention_c_amd import _cabi; L = _cabi.lib(); "
            "print(L.fa_kernel_name(_cabi.FA_DTYPE_F32, 64, 0).decode())")
    for env_val, want in ((None, "fa_fwd_f32_split_kernel"), ("exact", "fa_fwd_f32_kernel"), ("split", "fa_fwd_f32_split_kernel")):
        env = dict(os.environ)
        env.pop("FA_F32_AUTO", None)
        if env_val is not None:
            env["FA_F32_AUTO"] = env_val
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip().splitlines()[-1] == want


def test_cabi_rejects_bad_arguments_without_touching_a_device():
    L = _cabi.lib()
    buf = ctypes.create_string_buffer(5 * 16384)
    base = (ctypes.addressof(buf) + 15) & ~15
    p, k, v, o = (ctypes.c_void_p(base + i * 16384) for i in range(4))   # (1, 32, 64) fp32 = 8 KiB each, disjoint
    # null pointer
    assert L.fa_forward(None, k, v, o, 1, 32, 64, 1.0, 0, 0, None) == 1
    assert b"null" in L.fa_last_error()
    # misaligned
    assert L.fa_forward(ctypes.c_void_p(base + 4), k, v, o, 1, 32, 64, 1.0, 0, 0, None) == 1
    # bad sizes / scale / dtype
    assert L.fa_forward(p, k, v, o, 0, 32, 64, 1.0, 0, 0, None) == 1
    assert L.fa_forward(p, k, v, o, 1, 0, 64, 1.0, 0, 0, None) == 1
    assert L.fa_forward(p, k, v, o, 1, 32, 64, 0.0, 0, 0, None) == 1
    assert L.fa_forward(p, k, v, o, 1, 32, 64, float("nan"), 0, 0, None) == 1
    assert L.fa_forward(p, k, v, o, 1, 32, 64, 1.0, 0, 7, None) == 2
    # the output must not overlap an input (a tile that fails its verification is recomputed after o was written)
    for alias in (p, k, v, ctypes.c_void_p(base + 8192 - 16)):
        assert L.fa_forward(p, k, v, alias, 1, 32, 64, 1.0, 0, 0, None) == 1
        assert b"overlaps" in L.fa_last_error()
    # inputs may alias each other (self-attention on one tensor: the reference's own test.cu passes Q = K)
    assert L.fa_forward(p, p, p, o, 1, 8, 300, 1.0, 0, 0, None) == 2
    # unsupported head dim (above 256 for FA_KERNEL_AUTO; outside {32, 64, 128} for a family that is instantiated there only), unknown kernel id
    assert L.fa_forward(p, k, v, o, 1, 8, 300, 1.0, 0, 0, None) == 2
    assert b"300" in L.fa_last_error()
    assert L.fa_forward_ex(p, k, v, o, None, 1, 32, 48, 1.0, 0, 0, _cabi.FA_KERNEL_SPLIT, None) == 2 and b"48" in L.fa_last_error()
    assert L.fa_forward_ex(p, k, v, o, None, 1, 32, 48, 1.0, 0, 0, _cabi.FA_KERNEL_MFMA, None) == 2      # exact kernel: multiples of 32
    assert L.fa_forward_ex(p, k, v, o, None, 1, 32, 64, 1.0, 0, 0, 9, None) == 2
    # the two-term-P kernel exists for bf16 tensors only (head dims 32, 64, 128); the fp16-P kernels are not in the product library
    assert L.fa_forward_ex(p, k, v, o, None, 1, 32, 64, 1.0, 0, _cabi.FA_DTYPE_F32, _cabi.FA_KERNEL_PB2, None) == 2
    assert L.fa_forward_ex(p, k, v, o, None, 1, 32, 48, 1.0, 0, _cabi.FA_DTYPE_BF16, _cabi.FA_KERNEL_PB2, None) == 2
    assert L.fa_forward_ex(p, k, v, o, None, 1, 32, 64, 1.0, 0, _cabi.FA_DTYPE_BF16, 5, None) == 2                    # retired kernel id
    assert b"ablation" in L.fa_last_error()
    # fa_stats carries its size (ABI 6): host counters only, never blocks; a caller built against a shorter struct gets its prefix
    st = _cabi.FaStats()
    assert L.fa_get_stats(ctypes.byref(st), ctypes.sizeof(st)) == 0 and st.struct_bytes == ctypes.sizeof(st) == 3 * 8
    first = ctypes.c_uint64(0)
    assert L.fa_get_stats(ctypes.cast(ctypes.byref(first), ctypes.POINTER(_cabi.FaStats)), 8) == 0 and first.value == 24
    assert L.fa_get_stats(None, 24) == 1 and L.fa_get_stats(ctypes.byref(st), 4) == 1
    t, w = ctypes.c_uint64(7), ctypes.c_uint64(7)
    assert L.fa_read_device_counters(ctypes.byref(t), ctypes.byref(w)) == 0 and (t.value, w.value) == (0, 0)   # nothing ran here
    assert L.fa_read_device_counters(None, None) == 0
    assert b"abi 6" in L.fa_version()
    hdr = open(os.path.join(ROOT, "include", "flashattn_amd.h")).read()
    assert "#define FLASHATTN_AMD_ABI_VERSION 6" in hdr and max(len(l) for l in hdr.splitlines()) <= 120
    assert L.fa_forward_packed_qkv(p, o, 1, 8, 96, 5, None) == 1      # C % NH != 0
    assert L.fa_forward_packed_qkv(p, o, 1, 8, 600, 2, None) == 2     # hs = 300: above the largest head dim
    ms = ctypes.c_float()
    assert L.fa_time_forward(p, k, v, o, 1, 32, 64, 1.0, 0, 0, 0, None, 0, 0, ctypes.byref(ms)) == 1
    r = ctypes.c_int32(7)
    assert L.fa_last_forward_route(None, ctypes.byref(r)) == 0 and r.value == 0   # no chain launched on this thread
    assert L.fa_last_forward_route(None, None) == 1


def test_python_surface_validates_like_the_reference_signature():
    q = torch.randn(2, 32, 64)
    with pytest.raises(ValueError, match="GPU"):
        fa.forward(q, q, q, False)                      # CPU tensors: no CPU implementation, fail loudly
    with pytest.raises(ValueError, match="3-D"):
        fa.forward(q[0], q[0], q[0], False)
    with pytest.raises(ValueError, match="identical shapes"):
        fa.forward(q, q[:, :16], q, False)
    with pytest.raises(TypeError):
        fa.forward(q, q.double(), q, False)
    with pytest.raises(TypeError, match="not supported"):
        fa.forward(q.half(), q.half(), q.half(), False)
    with pytest.raises(ValueError):
        fa.forward_packed_qkv(torch.randn(2, 8, 100), 2)
    mod = fa.load(name="flash", sources=["src/main.cpp", "src/flashattention.cu"], extra_cuda_cflags=["-O3"])
    assert mod.forward is fa.forward


def test_missing_extension_fails_loudly(monkeypatch):
    monkeypatch.setattr(_cabi, "_lib", None)
    monkeypatch.setattr(_cabi, "LIB_PATH", "/nonexistent/libflashattn_amd.so")
    with pytest.raises(_cabi.ExtensionMissing, match="no CPU/PyTorch fallback"):
        _cabi.lib()


@pytest.mark.parametrize("bh,world", [(1024, 8), (16, 1), (16, 2), (10, 4), (3, 8), (128, 3)])
def test_shard_ranges_partition_the_slab_axis(bh, world):
    ranges = [fa.shard_range(bh, world, r) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == bh
    for (b0, e0), (b1, e1) in zip(ranges, ranges[1:]):
        assert e0 == b1 and b0 <= e0
    sizes = fa.shard_sizes(bh, world)
    assert sum(sizes) == bh and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        fa.shard_range(bh, world, world)


def test_bench_flop_and_byte_model():
    import bench
    # c4 = 2.749e11 FLOP, 67.1 MB; c2 = 3.436e10 FLOP, 134.2 MB (fp32)
    assert abs(bench.fwd_flop(16, 8192, 64, False) - 2.749e11) / 2.749e11 < 1e-3
    assert abs(bench.algorithmic_bytes(16, 8192, 64, 2) - 67.1e6) / 67.1e6 < 1e-3
    assert abs(bench.fwd_flop(128, 1024, 64, False) - 3.436e10) / 3.436e10 < 1e-3
    assert abs(bench.algorithmic_bytes(128, 1024, 64, 4) - 134.2e6) / 134.2e6 < 1e-3
    assert bench.fwd_flop(16, 8192, 64, True) * 2 == bench.fwd_flop(16, 8192, 64, False)


def test_harness_cpu_plumbing_config_c1():
    """BASELINE.json configs[0]: B=2 H=8 d=32 N=1024 fp32 via PyTorch CPU SDPA in the bench harness (plumbing, no GPU)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "flashattention.c_amd", "harness", "bench_flashattention.py"),
                        "--device", "cpu", "--batch_size", "2", "--seq_len", "1024", "--head_dim", "32"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PyTorch CPU SDPA" in r.stdout and "sanity check: PASSED" in r.stdout
    assert "no CPU implementation" in r.stdout
