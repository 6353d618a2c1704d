"""Build libml2048_hip.so (gfx950) in-tree with hipcc.  `python -m ml2048_amd.build`."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libml2048_hip.so")
SOURCES = [os.path.join(CSRC, "ml2048_hip.hip"), os.path.join(CSRC, "devrng.hip")]
HOST_SOURCES = [os.path.join(CSRC, "hostrng.cpp")]  # plain host C++ (no device code)
HEADERS = [os.path.join(CSRC, "board_ops.hpp"), os.path.join(_HERE, "..", "include", "ml2048_hip.h")]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in SOURCES + HOST_SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    objs = []
    for src in HOST_SOURCES:
        obj = os.path.splitext(src)[0] + ".o"
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-mssse3", "-std=c++17", "-fPIC", "-fopenmp", "-pthread", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    for src in SOURCES:
        obj = os.path.splitext(src)[0] + ".o"
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH, *objs, "-lgomp", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
