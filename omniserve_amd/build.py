"""Build libomniserve_hip.so (gfx950) in-tree with hipcc.

    python -m omniserve_amd.build [--force]

hipcc cross-compiles without a GPU.  The library lands next to this file
(omniserve_amd/libomniserve_hip.so) so that it travels with the source tree;
objects go to omniserve_amd/csrc/build/ (git-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libomniserve_hip.so")
ARCH = "gfx950"
SOURCES = ["qgemm_plan.hip", "qgemm_chn.hip", "qgemm_grp.hip", "qgemm_w8.hip",
           "elementwise.hip", "offpath.hip", "kv_cache.hip", "attn_prefill.hip", "sparse_utils.hip", "tp_comm.hip", "row_dtypes.hip", "norm_gemv_fused.hip"]
# -ffp-contract=off: the fp32 epilogues / quantisers must round exactly like oracle/ (explicit
# fma where wanted).  No -ffast-math anywhere.
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function",
         # qgemm_midm.h's LDS-DMA statements name M0 as clobbered ("reserved register" warning, 1 240 of them per build): the M0
         # discipline of those kernels is asserted on the disassembly instead (tests/test_code_objects_cpu.py)
         "-Wno-inline-asm", "-I" + CSRC, "-I" + os.path.join(os.path.dirname(HERE), "include")]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built")
    return exe


def _deps_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".hip", ".h", ".cpp")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return max(m, os.path.getmtime(os.path.abspath(__file__)))


EXT_SRC = os.path.join(HERE, "csrc_ext", "omni_ext.cpp")


def ext_path() -> str:
    import sysconfig
    return os.path.join(HERE, "_omni_ext" + sysconfig.get_config_var("EXT_SUFFIX"))


def needs_build() -> bool:
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < _deps_mtime():
        return True
    ext = ext_path()
    return not os.path.exists(ext) or os.path.getmtime(ext) < max(os.path.getmtime(EXT_SRC), os.path.getmtime(LIB))


def build_ext(verbose: bool = False) -> str:
    """The pybind11 fast path of the mirror (csrc_ext/omni_ext.cpp): host-only C++ against the installed torch's headers,
    linked to libomniserve_hip.so next to it (rpath $ORIGIN); g++, ~25 s.  The mirror works without it (ctypes)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    out = ext_path()
    inc = list(ce.include_paths()) + ["/opt/rocm/include", os.path.join(os.path.dirname(HERE), "include"),
                                      sysconfig.get_paths()["include"]]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=_omni_ext", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + i for i in inc]
    tmp = out + ".tmp"
    cmd += [EXT_SRC, "-o", tmp, "-L" + torch_lib, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip", "-ltorch_hip",
            "-L" + HERE, "-l:libomniserve_hip.so", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + torch_lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the pybind11 fast path failed:\n%s" % r.stderr[-4000:])
    os.replace(tmp, out)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _deps_mtime():      # only the binding is stale
        try:
            build_ext(verbose)
        except Exception as exc:   # noqa: BLE001
            print("omniserve_amd.build: the pybind11 fast path was not built (%s)" % str(exc)[-400:], file=sys.stderr)
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
        extra = os.environ.get("OMNI_HIPCC_EXTRA", "").split()   # tuning experiments only (e.g. -DOMNI_GEMM_MIN_BLOCKS=2)
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp"
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", tmp]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    os.replace(tmp, LIB)
    try:
        build_ext(verbose)
    except Exception as exc:   # noqa: BLE001  (the mirror works without it: ctypes)
        print("omniserve_amd.build: the pybind11 fast path was not built (%s); the ctypes mirror stays in use" % str(exc)[-400:],
              file=sys.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
