"""Read the gfx950 code objects embedded in an in-tree HIP library (test infrastructure, CPU only).

hipcc stores one clang-offload-bundle per translation unit in the `.hip_fatbin` section; every bundle holds an AMDGPU ELF whose
note section lists the kernels with their register and scratch budgets.  Used by tests/test_code_objects.py to keep the
properties DESIGN.md claims (no scratch in the kernels FA_KERNEL_AUTO can pick, MFMA -> VALU drains in the asm-MFMA kernels)
from silently regressing with a compiler or source change.
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import tempfile
from typing import Dict, List, NamedTuple

LLVM_BIN = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


class Kernel(NamedTuple):
    mangled: str
    name: str          # demangled
    scratch: int       # .private_segment_fixed_size, bytes per lane
    vgprs: int
    agprs: int
    sgprs: int
    lds: int
    code_object: str   # path of the extracted ELF


def extract(lib_path: str, outdir: str) -> List[str]:
    fat = os.path.join(outdir, "fatbin")
    # with one positional argument llvm-objcopy rewrites its INPUT in place (a normalised copy: other bytes, another sha256 -- the
    # library bench.py ties the PMC traffic to): the copy goes to a throwaway file, the library under test is only read
    subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(outdir, "objcopy_out")],
                   check=True)
    data = open(fat, "rb").read()
    paths, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            break
        (nb,) = struct.unpack_from("<Q", data, i + 24)
        off = i + 32
        for _ in range(nb):
            eo, es, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode()
            off += ts
            if "gfx950" in triple and es > 0:
                p = os.path.join(outdir, f"k{len(paths)}.co")
                with open(p, "wb") as f:
                    f.write(data[i + eo:i + eo + es])
                paths.append(p)
        pos = i + len(MAGIC)
    return paths


def kernels_of(lib_path: str, outdir: str = None) -> Dict[str, Kernel]:
    outdir = outdir or tempfile.mkdtemp(prefix="fa_codeobj_")
    found = {}
    for co in extract(lib_path, outdir):
        notes = subprocess.run([os.path.join(LLVM_BIN, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("  - .agpr_count")[1:]:
            g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))  # noqa: E731
            mangled = re.search(r"\.name:\s+(\S+)", blk).group(1)
            found[mangled] = (g("private_segment_fixed_size"), g("vgpr_count"), int(re.search(r":\s+(\d+)", blk).group(1)), g("sgpr_count"),
                              g("group_segment_fixed_size"), co)
    names = list(found)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout.splitlines()
    return {m: Kernel(m, d, *found[m]) for m, d in zip(names, dem)}


def disassemble(co: str) -> str:
    return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout


if __name__ == "__main__":
    import sys
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "flashattention.c_amd", "libflashattn_amd.so")
    ks = kernels_of(lib)
    print(f"{len(ks)} kernels in {lib}")
    for k in sorted(ks.values(), key=lambda k: -k.scratch):
        if k.scratch:
            print(f"scratch {k.scratch:4d} B  vgpr {k.vgprs:3d} agpr {k.agprs:3d} lds {k.lds:6d}  {k.name}")
