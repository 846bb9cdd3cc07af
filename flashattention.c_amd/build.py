"""Build the gfx950 shared library and the torch-less C driver with hipcc (in-tree, no JIT cache).

Counterpart of the reference's import-time JIT build
(``torch.utils.cpp_extension.load(name='flash', sources=[main.cpp, flashattention.cu], -O3)``,
/root/reference/bench_flashattention.py:10): here the build is an explicit step that leaves
``flashattention.c_amd/libflashattn_amd.so`` next to the sources so that it travels with the tree.
hipcc cross-compiles for gfx950 without a GPU present.

    python flashattention.c_amd/build.py [--force] [--jobs N] [--ablation] [--torch-binding]

``--ablation`` additionally builds ``libflashattn_amd_ablation.so`` (+ ``fa_driver_ablation``): the same sources compiled with
``-DFA_ABLATION=1`` plus the timing-only ablation instantiations quoted in DESIGN.md section 4 (results are garbage by design).
They are NOT part of the product library: its ``fa_forward_ex`` rejects their variant numbers.
``--sanitize`` builds ``libflashattn_amd_asan.so``: the host translation units (``fa_plan / fa_counters / fa_launch / fa_shard / fa_timing / fa_api .cpp`` + ``fa_selftest.cpp``) recompiled with ``-fsanitize=address,undefined``
(device code and every other object unchanged) and ``-DFA_HOST_TEST=1``, which adds ``fa_host_selftest()`` -- plans over a shape grid,
key-split arithmetic and the head-dim routing, none of which needs a device.  ``sanitize_selfcheck()`` runs it (and
the no-device validation paths through ctypes) in a child process with the ASan runtime preloaded; ``__graft_entry__.build()`` calls it.
``--torch-binding`` builds ``flash_torch_binding*.so``, the compiled pybind translation unit of INTEGRATION.md section 1
(``torch::Tensor forward(Q, K, V, bool causal)``, /root/reference/src/main.cpp:3-6) against the installed torch-ROCm.
"""
from __future__ import annotations

import argparse
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(PKG_DIR, "libflashattn_amd.so")
DRIVER_PATH = os.path.join(PKG_DIR, "fa_driver")
ARCH = "gfx950"

# the host side behind include/flashattn_amd.h (csrc/fa_host.h lists what each holds)
HOST_SOURCES = ["fa_plan.cpp", "fa_counters.cpp", "fa_launch.cpp", "fa_shard.cpp", "fa_timing.cpp", "fa_api.cpp"]
# longest translation units first: the thread pool starts them in this order
LIB_SOURCES = ["fa_fwd_bf16_x4_pb2_f32out.hip", "fa_fwd_bf16_x4_pb2_bf16out.hip", "fa_fwd_bf16_x2.hip", "fa_fwd_bf16_x4.hip",
               "fa_fwd_bf16_x2_pb2_d128_f32out.hip", "fa_fwd_bf16_x2_pb2_d128_bf16out.hip", "fa_fwd_bf16_x2_pb2_d64_f32out.hip", "fa_fwd_bf16_x2_pb2_d64_bf16out.hip",
               "fa_fwd_bf16_pipelined.hip", "fa_split_f32_d64.hip", "fa_split_f32_d128.hip", "fa_split_f32_d32.hip",
               "fa_split_bf16_d64.hip", "fa_split_bf16_d128.hip", "fa_split_bf16_d32.hip", "fa_fwd_bf16.hip", "fa_combine.hip",
               "fa_fwd_bf16_x2_pb2_d32_f32out.hip", "fa_fwd_bf16_x2_pb2_d32_bf16out.hip",
               "fa_fwd_f32_wide.hip", "fa_fwd_f32_wide_bf16.hip", "fa_fwd_f32.hip", "fa_fwd_f32_split.hip", "fa_fwd_bf16_split.hip", "fa_naive.hip", *HOST_SOURCES]
# csrc/experiments/: only in libflashattn_amd_ablation.so -- timing-only instantiations (garbage results), superseded kernel generations
# and the fp16-P families the round-4 accurate path (P as two bf16 terms) replaced (correct, tested through fa_driver_ablation)
ABLATION_SOURCES = ["experiments/" + f for f in (
    "fa_fwd_bf16_x4_ablation.hip", "fa_fwd_bf16_x4_causal.hip", "fa_fwd_bf16_pp2.hip", "fa_fwd_f32_t3.hip", "fa_cvt.hip", "fa_fwd_bf16_x4_p16.hip",
    "fa_fwd_bf16_x2_p16_d128.hip", "fa_fwd_bf16_x2_p16_d64.hip", "fa_fwd_bf16_x2_p16_d32.hip", "fa_fwd_bf16_x2_p16x2_d128.hip",
    "fa_fwd_bf16_x2_p16x2_d64.hip", "fa_fwd_bf16_x2_p16x2_d32.hip")]
# product sources whose text depends on FA_ABLATION (the variant dispatch): recompiled for the ablation library, the rest is shared
ABLATION_DEPENDENT = ["fa_fwd_bf16.hip", "fa_fwd_bf16_x4.hip", "fa_fwd_bf16_pipelined.hip", "fa_fwd_bf16_x2.hip", "fa_plan.cpp", "fa_counters.cpp", "fa_launch.cpp", "fa_api.cpp",
                      "fa_split_f32_d32.hip", "fa_split_f32_d64.hip"]
ABL_LIB_PATH = os.path.join(PKG_DIR, "libflashattn_amd_ablation.so")
ABL_DRIVER_PATH = os.path.join(PKG_DIR, "fa_driver_ablation")
# every header under csrc/ (globbed: a new header cannot be forgotten -- ADVICE r04) plus the C ABI
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "experiments", "*.h"))) + [os.path.join(ROOT, "include", "flashattn_amd.h")]
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
                "-I", os.path.join(ROOT, "include"), "-I", CSRC]
# per-source extras.  The split kernel keeps its fp32 arithmetic scalar: the SLP vectoriser would pair it into
# v_pk_add_f32 / v_pk_mul_f32, which block the matrix pipe's issue for a full MFMA slot each on gfx950.
EXTRA_FLAGS = {f"fa_split_{dt}_d{d}.hip": ["-fno-slp-vectorize"] for dt in ("f32", "bf16") for d in (32, 64, 128)}
EXTRA_FLAGS["experiments/fa_fwd_f32_t3.hip"] = ["-fno-slp-vectorize"]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the ROCm toolchain is required to build flashattention.c_amd")
    return exe


def _mtime(p: str) -> float:
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _newest_dep() -> float:
    deps = [os.path.join(CSRC, h) if not os.path.isabs(h) else h for h in HEADERS]
    return max(_mtime(d) for d in deps + [os.path.abspath(__file__)])


def _compile(src: str, force: bool, ablation: bool = False) -> str:
    obj = os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + (".abl.o" if ablation else ".o"))
    srcp = os.path.join(CSRC, src)
    if not force and _mtime(obj) > max(_mtime(srcp), _newest_dep()):
        return obj
    # FA_EXTRA_ABL_FLAGS: experiment switch -- extra -D flags for the ablation library only (A/B of a kernel change in one process)
    extra_abl = os.environ.get("FA_EXTRA_ABL_FLAGS", "").split() if ablation else []
    cmd = [hipcc(), *COMMON_FLAGS, f"-DFA_ABLATION={1 if ablation else 0}", *extra_abl, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def _link(lib_path: str, objs, force: bool) -> None:
    if force or _mtime(lib_path) < max(_mtime(o) for o in objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_path, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")


def _build_driver(out: str, lib_path: str, libname: str, force: bool) -> None:
    drv_src = os.path.join(CSRC, "fa_driver.cpp")
    if os.path.exists(drv_src) and (force or _mtime(out) < max(_mtime(drv_src), _mtime(lib_path))):
        cmd = [hipcc(), "-O2", "-std=c++17", f"--offload-arch={ARCH}", "-x", "hip", "-I", os.path.join(ROOT, "include"),
               drv_src, "-o", out, "-L", PKG_DIR, f"-l{libname}", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"driver build failed:\n{r.stdout}\n{r.stderr}")


def build(force: bool = False, jobs: int = 8, verbose: bool = False, ablation: bool = False, torch_binding: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), LIB_SOURCES))
        abl_objs = []
        if ablation:
            abl_objs = list(ex.map(lambda s: _compile(s, force, True), ABLATION_SOURCES + ABLATION_DEPENDENT))
            abl_objs += [o for s_, o in zip(LIB_SOURCES, objs) if s_ not in ABLATION_DEPENDENT]
    _link(LIB_PATH, objs, force)
    _build_driver(DRIVER_PATH, LIB_PATH, "flashattn_amd", force)
    if ablation:
        _link(ABL_LIB_PATH, abl_objs, force)
        _build_driver(ABL_DRIVER_PATH, ABL_LIB_PATH, "flashattn_amd_ablation", force)
    if torch_binding:
        build_torch_binding(force=force, verbose=verbose)
    if verbose:
        print(f"built {LIB_PATH}" + (f" and {ABL_LIB_PATH}" if ablation else ""))
    return LIB_PATH


ASAN_LIB_PATH = os.path.join(PKG_DIR, "libflashattn_amd_asan.so")
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g"]


def build_sanitized(force: bool = False) -> str:
    """libflashattn_amd_asan.so: the host translation units (HOST_SOURCES + the self-test) under ASan + UBSan (no GPU sanitizer: not
    available on this pool); every kernel object is the product's."""
    build(force=False)
    san_objs = []
    for name in HOST_SOURCES + ["fa_selftest.cpp"]:
        src = os.path.join(CSRC, name)
        obj = os.path.join(OBJ_DIR, os.path.splitext(name)[0] + ".asan.o")
        if force or _mtime(obj) < max(_mtime(src), _newest_dep()):
            cmd = [hipcc(), *[f for f in COMMON_FLAGS if f != "-O3"], "-O1", "-DFA_ABLATION=0", "-DFA_HOST_TEST=1", *SAN_FLAGS, "-x", "hip", "-c", src, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"sanitizer build of {name} failed:\n{r.stdout}\n{r.stderr}")
        san_objs.append(obj)
    objs = [os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(s))[0] + ".o") for s in LIB_SOURCES if s not in HOST_SOURCES] + san_objs
    if force or _mtime(ASAN_LIB_PATH) < max(_mtime(o) for o in objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize", "-o", ASAN_LIB_PATH, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link of the sanitizer library failed:\n{r.stdout}\n{r.stderr}")
    return ASAN_LIB_PATH


def asan_runtime() -> str:
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc())), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    return subprocess.run([clang, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()


_SELFCHECK = r"""
import ctypes, sys
L = ctypes.CDLL(sys.argv[1])
L.fa_host_selftest.restype = ctypes.c_int
rc = L.fa_host_selftest()
assert rc == 0, f"fa_host_selftest: check {rc} failed"
# the no-device paths of the ABI through ctypes, as tests/test_host_logic.py drives them
L.fa_workspace_bytes.restype = ctypes.c_size_t
L.fa_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
assert L.fa_workspace_bytes(1, 8192, 64, 0, 1, 0) == 8 * 8192 * 64 * 4 + 8 * 8192 * 4
L.fa_kernel_name_for.restype = ctypes.c_char_p
L.fa_kernel_name_for.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64]
for dt in (0, 1, 2):
    for d in (32, 64, 128, 48, 96, 256, 300):
        for c in (0, 1):
            for bh in (1, 16, 130, 70000):
                for n in (1, 700, 8192, 40000, 1 << 24):
                    L.fa_kernel_name_for(dt, d, c, bh, n)
L.fa_last_error.restype = ctypes.c_char_p
assert L.fa_forward(None, None, None, None, 1, 1, 64, ctypes.c_float(1.0), 0, 0, None) == 1 and b"null" in L.fa_last_error()
print("sanitizer self-check ok")
"""


def sanitize_selfcheck(force: bool = False) -> str:
    """Build the sanitizer library and run its host self-test under ASan + UBSan in a child process (a finding aborts it: non-zero
    exit, the report on stderr).  No device is needed or touched."""
    lib = build_sanitized(force)
    env = dict(os.environ, LD_PRELOAD=asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, "-c", _SELFCHECK, lib], env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0 or "self-check ok" not in r.stdout:
        raise RuntimeError(f"sanitizer self-check failed (rc {r.returncode}):\n{r.stdout[-3000:]}\n{r.stderr[-6000:]}")
    return r.stdout.strip()


TORCH_BINDING_NAME = "flash_torch_binding"


def torch_binding_path() -> str:
    import sysconfig
    return os.path.join(PKG_DIR, TORCH_BINDING_NAME + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_torch_binding(force: bool = False, verbose: bool = False) -> str:
    """The reference's main.cpp surface as a compiled pybind11 module over the C ABI (csrc/fa_torch_binding.cpp)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "fa_torch_binding.cpp")
    out = torch_binding_path()
    if not force and _mtime(out) > max(_mtime(src), _mtime(LIB_PATH), _mtime(os.path.join(ROOT, "include", "flashattn_amd.h"))):
        return out
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    incs = [i for p in ce.include_paths() for i in ("-isystem", p)] + ["-isystem", sysconfig.get_paths()["include"],
                                                                        "-isystem", os.path.join(rocm, "include")]
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    abi = int(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True))
    cmd = [hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", src, "-o", out, "-I", os.path.join(ROOT, "include"), *incs,
           f"-D_GLIBCXX_USE_CXX11_ABI={abi}", f"-DTORCH_EXTENSION_NAME={TORCH_BINDING_NAME}", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-L", PKG_DIR, "-lflashattn_amd", "-L", tlib, "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip", "-ltorch_python",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}", "-Wno-unused-command-line-argument"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"torch binding build failed:\n{r.stdout[-3000:]}\n{r.stderr[-6000:]}")
    if verbose:
        print(f"built {out}")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--ablation", action="store_true", help="also build libflashattn_amd_ablation.so + fa_driver_ablation")
    ap.add_argument("--torch-binding", action="store_true", help="also build the pybind11 module of INTEGRATION.md section 1")
    ap.add_argument("--sanitize", action="store_true", help="also build libflashattn_amd_asan.so (host ASan + UBSan) and run its self-check")
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs, verbose=True, ablation=a.ablation, torch_binding=a.torch_binding)
    if a.sanitize:
        print(sanitize_selfcheck(force=a.force))
    sys.exit(0)
