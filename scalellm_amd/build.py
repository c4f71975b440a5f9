"""Build libslm_hip.so (the C-ABI HIP kernel library) in-tree with hipcc for gfx950.

    python -m scalellm_amd.build            # incremental
    python -m scalellm_amd.build --force

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
INCLUDE = os.path.join(ROOT, "include")
OBJ_DIR = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libslm_hip.so")
ARCH = "gfx950"

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXXFLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}",
]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps_mtime():
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _file_flags(src: str):
    """Extra hipcc flags a source file asks for on a `// hipcc-flags: ...` line in its header."""
    flags = []
    with open(src) as f:
        for i, line in enumerate(f):
            if i > 80:
                break
            if line.startswith("// hipcc-flags:"):
                flags += line.split(":", 1)[1].split()
    return flags


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
    newest = max(os.path.getmtime(src), _deps_mtime())
    if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
        cmd = [HIPCC, *CXXFLAGS, *_file_flags(src), "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    if not srcs:
        raise RuntimeError("no .hip sources found")
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or any(
            os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB, *objs]
        subprocess.check_call(cmd)
        if verbose:
            print(f"[scalellm_amd.build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
