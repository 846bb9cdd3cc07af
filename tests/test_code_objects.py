"""CPU: properties of the gfx950 code objects inside the in-tree library (no GPU needed -- hipcc cross-compiles here).

What DESIGN.md claims about the binaries is checked on the binaries: no kernel spills to scratch, the kernels whose matrix
instructions are inline asm (invisible to hipcc's hazard padding) drain the matrix pipe before VALU code reads an accumulator,
and the timing-only ablation instantiations are not part of the product library."""
import os
import re

import pytest

from tests import codeobj

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "flashattention.c_amd", "libflashattn_amd.so")
ABL_LIB = os.path.join(ROOT, "flashattention.c_amd", "libflashattn_amd_ablation.so")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    assert os.path.exists(LIB), "build the library first (python flashattention.c_amd/build.py)"
    return codeobj.kernels_of(LIB, str(tmp_path_factory.mktemp("co")))


def test_no_kernel_of_the_product_library_uses_scratch(kernels):
    """private_segment_fixed_size == 0 for every kernel (FA_KERNEL_AUTO can pick any family): a spill inside an attention loop is
    an HBM round trip per step, and register pressure regressions show up here before they show up in a benchmark."""
    assert len(kernels) > 60
    spilling = [f"{k.scratch} B: {k.name}" for k in kernels.values() if k.scratch != 0]
    assert not spilling, "\n".join(spilling)


def test_every_kernel_family_the_dispatch_names_is_present(kernels):
    names = "\n".join(k.name for k in kernels.values())
    for fam in ("fa_fwd_bf16_x4_kernel", "fa_fwd_bf16_x4_pb2_kernel", "fa_fwd_bf16_x2_kernel", "fa_fwd_bf16_x2_pb2_kernel",
                "fa_fwd_bf16_pp3_kernel", "fa_fwd_bf16_w4_kernel", "fa_fwd_bf16_kernel", "fa_fwd_f32_split_kernel",
                "fa_fwd_f32_kernel", "fa_naive_f32_kernel", "fa_combine_splits_kernel"):
        assert fam + "<" in names or fam + "(" in names, fam


def _family(name: str) -> str:
    m = re.match(r"_ZN2fa(\d+)", name)   # a name the demangler gave up on (template arguments of type __bf16: DF16b)
    if m:
        return name[m.end():m.end() + int(m.group(1))]
    return re.sub(r"^void ", "", name).split("<")[0].split("(")[0].split("::")[-1]


def test_every_kernel_of_the_product_library_is_reachable_or_documented(kernels):
    """Round 2 shipped five bf16 families x 11 selectable tilings, several of them unreachable from FA_KERNEL_AUTO.  Now: every kernel
    FAMILY in the product library is named by fa_kernel_name_for() for some (dtype, d, causal, bh, n) of a shape grid -- i.e. the
    dispatch reaches it -- or is one of the few documented helpers / explicit choices of include/flashattn_amd.h; and within the
    one-wave-per-SIMD families only the instantiations the dispatch launches are present (barrier every two stages, optimistic mix with
    its redo; 512-row workgroups non-causal only)."""
    from flashattention_c_amd import _cabi
    L = _cabi.lib()
    reachable = set()
    for dtype in (_cabi.FA_DTYPE_F32, _cabi.FA_DTYPE_BF16, _cabi.FA_DTYPE_BF16_OUT_F32):
        for d in (32, 64, 128):
            for causal in (0, 1):
                for bh in (1, 2, 3, 8, 12, 16, 24, 33, 64, 128, 130, 256, 1024, 70000):
                    for n in (1, 31, 300, 700, 1024, 2048, 3000, 4096, 8192, 16384, 40000):
                        nm = L.fa_kernel_name_for(dtype, d, causal, bh, n)
                        assert nm is not None
                        reachable.add(nm.decode())
    documented = {
        "fa_naive_f32_kernel",            # FA_KERNEL_NAIVE: rung 0, the on-device cross-check
        "fa_fwd_f32_kernel",              # FA_KERNEL_MFMA for fp32 tensors and the guarded chain's fallback
        "fa_combine_splits_kernel",       # combine of a key-split launch
        "fa_fwd_bf16_kernel", "fa_fwd_bf16_w4_kernel",              # slabs beyond 32-bit byte offsets / small d = 128 grids (phase-structured)
    }
    present = {_family(k.name) for k in kernels.values()}
    stray = present - reachable - documented
    assert not stray, f"kernel families neither reachable from FA_KERNEL_AUTO nor documented: {sorted(stray)}"
    assert reachable <= present | {"fa_fwd_f32_kernel"}, sorted(reachable - present)
    for k in kernels.values():
        assert not re.search(r"_p16(x2)?_kernel|fa_cvt_bf16_to_f16", k.name), f"fp16-P family in the product library (ablation library only since round 4): {k.name}"
        m = re.search(r"fa_fwd_bf16_x4(?:_pb2)?_kernel<(.*?)>\(", k.name)
        if m:   # <NWAVES, CAUSAL, OUT_F32, G, ...>: non-causal, barrier every two stages
            a = m.group(1).split(", ")
            assert a[1] == "false" and a[3] == "2", k.name
        m = re.search(r"fa_fwd_bf16_x2(?:_pb2)?_kernel<(.*?)>\(", k.name)
        if m:   # <D, NWAVES, CAUSAL, OUT_F32, G, ABL, OPTIMISTIC>
            a = m.group(1).split(", ")
            assert a[4] == "2" and a[6] == "true", k.name
        m = re.search(r"fa_fwd_bf16_pp3_kernel<(.*?)>\(", k.name)
        if m:   # <D, NWAVES, CAUSAL, OUT_F32, PROF, G, OPTIMISTIC>
            a = m.group(1).split(", ")
            assert a[0] == "64" and a[1] == "4" and a[6] == "true", k.name
        # the launchers of the NB = 2 kernels at d <= 64 assume two workgroups fit a CU (xn_launch_order pairs causal tiles on that
        # premise; a kernel that silently grew past half a SIMD's register file would run one per CU in the paired order)
        m = re.search(r"fa_fwd_bf16_x2(?:_pb2)?_kernel<(32|64), ", k.name)
        if m:
            assert k.vgprs <= 256, f"{k.name}: {k.vgprs} registers -- no longer fits a CU twice"
    # (3.7 MB through round 5; round 6 added the exact kernel at five more head dims, for fp32 and for bf16 tensors: 30 kernels, 4.3 MB)
    assert os.path.getsize(LIB) < 4.5 * 1024 * 1024, "the product library grew past 4.5 MB (it was 5.7 MB with every round-2 tiling in it)"


def test_timing_only_ablations_are_not_in_the_product_library(kernels):
    for k in kernels.values():
        assert "pp2" not in k.name, k.name
        m = re.search(r"fa_fwd_bf16_x4_kernel<(.*?)>\(", k.name)
        if m:   # <NWAVES, CAUSAL, OUT_F32, G, ABL, OPTIMISTIC>
            assert m.group(1).split(", ")[4] == "0", k.name
        m = re.search(r"fa_fwd_bf16_x2_kernel<(.*?)>\(", k.name)
        if m:   # <D, NWAVES, CAUSAL, OUT_F32, G, ABL, OPTIMISTIC>
            assert m.group(1).split(", ")[5] == "0", k.name
        m = re.search(r"fa_fwd_bf16_pp3_kernel<(.*?)>\(", k.name)
        if m:   # <D, NWAVES, CAUSAL, OUT_F32, PROF, G, OPTIMISTIC>: the in-kernel phase timers overwrite the lse buffer
            assert m.group(1).split(", ")[4] == "false", k.name
    if os.path.exists(ABL_LIB):   # ... and they do exist in the separate ablation library
        abl = codeobj.kernels_of(ABL_LIB)
        assert any("pp2" in k.name for k in abl.values())
        assert any(re.search(r"fa_fwd_bf16_x4_kernel<4, false, false, 2, [1-9]", k.name) for k in abl.values())


def test_asm_mfma_kernels_drain_the_matrix_pipe_before_reading_accumulators(kernels):
    """The x4 / x2 kernels issue their MFMAs from inline asm into AGPR accumulators; hipcc pads no hazard for them.  Every
    v_accvgpr_read of a register some v_mfma WRITES (epilogue, rescale branch) must sit at least 18 wait states behind the closest
    preceding v_mfma with that register in its destination, in program order (a 32x32x16 MFMA has 16 passes); the register-tied drains
    provide 64.  A scheduler that hoists a read above its drain fails here.  (Reads of AGPRs no matrix instruction of the last 18 wait
    states writes -- values hipcc parks there -- are not accumulator reads: round 6 met one three wait states behind an MFMA.)"""
    dis_cache = {}
    checked = 0
    for k in kernels.values():
        if not re.search(r"fa_fwd_bf16_x[24](_pb2)?_kernel<", k.name):
            continue
        dis = dis_cache.setdefault(k.code_object, codeobj.disassemble(k.code_object))
        i = dis.index("<" + k.mangled + ">:")
        body = dis[i:dis.find("\n\n", i)].splitlines()[1:]
        clock, last_write, worst, reads = 0, {}, None, 0   # last_write: AGPR index -> clock of the last v_mfma writing it
        for line in body:
            ins = line.split("//")[0].replace(",", " ").split()
            if not ins:
                continue
            op = ins[0]
            if op.startswith("v_mfma"):
                m = re.match(r"a\[(\d+):(\d+)\]", ins[1])
                if m:
                    for r in range(int(m.group(1)), int(m.group(2)) + 1):
                        last_write[r] = clock
                clock += 1
                continue
            if op.startswith("v_accvgpr_read"):
                m = re.match(r"a(\d+)$", ins[2])
                if m and int(m.group(1)) in last_write:
                    reads += 1
                    gap = clock - last_write[int(m.group(1))] - 1
                    worst = gap if worst is None else min(worst, gap)
            clock += int(ins[1]) + 1 if op == "s_nop" else 1
        assert reads > 0 and worst is not None and worst >= 18, f"{k.name}: {worst} wait states between an MFMA and a read of its accumulator"
        checked += 1
    assert checked >= 28   # x4: 2, x2: 12, x4_pb2: 2, x2_pb2: 12


def test_inspecting_the_library_does_not_modify_it(tmp_path):
    """The code-object extraction only READS the library: bench.py ties the committed PMC traffic to the library's sha256, and an
    llvm-objcopy call with one positional argument used to rewrite the file in place (same code, other bytes)."""
    import hashlib
    before = hashlib.sha256(open(LIB, "rb").read()).hexdigest()
    assert len(codeobj.kernels_of(LIB, str(tmp_path))) > 60
    assert hashlib.sha256(open(LIB, "rb").read()).hexdigest() == before


def test_no_agpr_operand_of_an_asm_mfma_is_written_right_in_front_of_it(kernels):
    """The other hazard hipcc cannot pad for an inline-asm MFMA (found in round 6, profiles/r06_exp3_d32_q2.txt): when register pressure
    makes it keep a B-operand fragment (Q) in VGPRs, it stages the fragment through AGPRs with v_accvgpr_write immediately in front of
    the asm statement -- a VALU write two wait states too close to the matrix instruction that reads it (wrong scores, no fault).  In the
    shipped kernels every Q fragment lives in its AGPRs from the prologue on: no v_accvgpr_write into an AGPR range within two wait states
    in front of an MFMA that reads that range."""
    dis_cache = {}
    checked = 0
    for k in kernels.values():
        if not re.search(r"fa_fwd_bf16_x[24](_pb2)?_kernel<", k.name):
            continue
        dis = dis_cache.setdefault(k.code_object, codeobj.disassemble(k.code_object))
        i = dis.index("<" + k.mangled + ">:")
        body = [line.split("//")[0].strip() for line in dis[i:dis.find("\n\n", i)].splitlines()[1:]]
        body = [b for b in body if b]
        for j, ins in enumerate(body):
            if not ins.startswith("v_mfma"):
                continue
            ranges = [(int(a), int(b)) for a, b in re.findall(r"a\[(\d+):(\d+)\]", ins.split(",", 1)[1] if "," in ins else "")]   # source operands
            wait = 0
            for prev in reversed(body[max(0, j - 4):j]):
                if prev.startswith("v_mfma") or wait >= 2:
                    break
                if prev.startswith("s_nop"):
                    wait += int(prev.split()[1]) + 1
                    continue
                w = re.match(r"v_accvgpr_write_b32 a(\d+),", prev)
                assert not (w and any(a <= int(w.group(1)) <= b for a, b in ranges)), f"{k.name}: `{prev}` {wait} wait state(s) in front of `{ins}`"
                wait += 1
        checked += 1
    assert checked >= 28
