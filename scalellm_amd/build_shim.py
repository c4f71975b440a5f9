"""Build the C++ libtorch shim (_slm_shim.so: reference C++ operator signatures + pybind test
surface) in-tree with g++ against the installed torch-ROCm headers.

    python -m scalellm_amd.build_shim [--force]

Host code only -- the device code lives in libslm_hip.so, which this links against.
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SHIM = os.path.join(CSRC, "shim")
ROOT = os.path.dirname(HERE)
OUT = os.path.join(CSRC, "_slm_shim.so")


def build(force: bool = False) -> str:
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce

    from . import build as slm_build
    slm_build.build()
    srcs = [os.path.join(SHIM, f) for f in ("slm_torch_shim.cpp", "slm_qlinear_hip.cpp", "slm_attn_handler_hip.cpp", "slm_llama_hip.cpp", "slm_shim_pybind.cpp")]
    deps = srcs + [os.path.join(SHIM, "slm_torch_shim.h"), os.path.join(SHIM, "slm_qlinear_hip.h"), os.path.join(SHIM, "slm_attn_handler_hip.h"), os.path.join(SHIM, "slm_llama_hip.h"),
                   os.path.join(ROOT, "include", "slm_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) > os.path.getmtime(d) for d in deps):
        return OUT
    tlib = ce.library_paths()[0]
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include",
        f"-I{sysconfig.get_paths()['include']}", f"-I{pybind11.get_include()}",
        f"-I{os.path.join(ROOT, 'include')}", f"-I{SHIM}"]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-D__HIP_PLATFORM_AMD__=1",
           "-DUSE_ROCM=1", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}",
           "-DTORCH_EXTENSION_NAME=_slm_shim", *inc, *srcs, "-o", OUT, f"-L{tlib}", f"-L{CSRC}",
           "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ltorch_python",
           "-lrccl", "-lslm_hip", f"-Wl,-rpath,{tlib}", "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(cmd)
    print(f"[scalellm_amd.build_shim] built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
