"""Build the gfx950 shared library and the torch-less C driver with hipcc (in-tree, no JIT cache).

Counterpart of the reference's import-time JIT build
(``torch.utils.cpp_extension.load(name='flash', sources=[main.cpp, flashattention.cu], -O3)``,
/root/reference/bench_flashattention.py:10): here the build is an explicit step that leaves
``flashattention.c_amd/libflashattn_amd.so`` next to the sources so that it travels with the tree.
hipcc cross-compiles for gfx950 without a GPU present.

    python flashattention.c_amd/build.py [--force] [--jobs N]
"""
from __future__ import annotations

import argparse
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

# longest translation units first: the thread pool starts them in this order
LIB_SOURCES = ["fa_fwd_bf16_x4_ablation.hip", "fa_fwd_bf16_x4.hip", "fa_fwd_bf16_x4_causal.hip", "fa_fwd_bf16_x2.hip",
               "fa_fwd_bf16_pipelined.hip", "fa_split_f32_d64.hip", "fa_split_f32_d128.hip", "fa_split_f32_d32.hip",
               "fa_split_bf16_d64.hip", "fa_split_bf16_d128.hip", "fa_split_bf16_d32.hip", "fa_fwd_bf16_pp2.hip", "fa_fwd_bf16.hip",
               "fa_fwd_f32.hip", "fa_fwd_f32_split.hip", "fa_fwd_bf16_split.hip", "fa_naive.hip", "fa_api.cpp"]
HEADERS = ["fa_common.h", "fa_kernels.h", "fa_bf16_common.h", "fa_split_kernel.h", "fa_bf16_x4_kernel.h", "fa_bf16_step.h", os.path.join(ROOT, "include", "flashattn_amd.h")]
COMMON_FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
                "-I", os.path.join(ROOT, "include")]
# per-source extras.  The split kernel keeps its fp32 arithmetic scalar: the SLP vectoriser would pair it into
# v_pk_add_f32 / v_pk_mul_f32, which block the matrix pipe's issue for a full MFMA slot each on gfx950.
EXTRA_FLAGS = {f"fa_split_{dt}_d{d}.hip": ["-fno-slp-vectorize"] for dt in ("f32", "bf16") for d in (32, 64, 128)}


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


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    srcp = os.path.join(CSRC, src)
    if not force and _mtime(obj) > max(_mtime(srcp), _newest_dep()):
        return obj
    cmd = [hipcc(), *COMMON_FLAGS, *EXTRA_FLAGS.get(src, []), "-x", "hip", "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, jobs: int = 8, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), LIB_SOURCES))
    if force or _mtime(LIB_PATH) < max(_mtime(o) for o in objs):
        cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    drv_src = os.path.join(CSRC, "fa_driver.cpp")
    if os.path.exists(drv_src) and (force or _mtime(DRIVER_PATH) < max(_mtime(drv_src), _mtime(LIB_PATH))):
        cmd = [hipcc(), "-O2", "-std=c++17", f"--offload-arch={ARCH}", "-x", "hip", "-I", os.path.join(ROOT, "include"),
               drv_src, "-o", DRIVER_PATH, "-L", PKG_DIR, "-lflashattn_amd", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"driver build failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    build(force=a.force, jobs=a.jobs, verbose=True)
    sys.exit(0)
