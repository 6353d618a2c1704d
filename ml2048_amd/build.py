"""Build libml2048_hip.so (gfx950) in-tree with hipcc.  `python -m ml2048_amd.build`."""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libml2048_hip.so")
SOURCES = [os.path.join(CSRC, "ml2048_hip.hip"), os.path.join(CSRC, "step_prepare.hip"), os.path.join(CSRC, "devrng.hip"),
           os.path.join(CSRC, "devrng2.hip"), os.path.join(CSRC, "encoder.hip")]
HOST_SOURCES = [os.path.join(CSRC, "hostrng.cpp")]  # plain host C++ (no device code)
HEADERS = [os.path.join(CSRC, "board_ops.hpp"), os.path.join(CSRC, "device_helpers.hpp"),
           os.path.join(_HERE, "..", "include", "ml2048_hip.h")]


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


def build(force: bool = False, verbose: bool = False, extra_hipcc_flags=(), out: str | None = None) -> str:
    """Compile and link the library.  `extra_hipcc_flags` / `out` build an A/B variant next to the
    default one (`ML2048_HIP_LIB=<out>` selects it at load time); they never change the default build."""
    if out is None and not extra_hipcc_flags:
        if not force and not is_stale():
            return LIB_PATH
    lib_path = out or LIB_PATH
    suffix = ".ab.o" if out else ".o"
    jobs = []  # (command, object): independent compilations, run a few at a time (the big file is compiled four times)
    for src in HOST_SOURCES:
        obj = os.path.splitext(src)[0] + suffix
        jobs.append(([os.environ.get("CXX", "g++"), "-O2", "-mssse3", "-std=c++17", "-fPIC", "-fopenmp", "-pthread", "-c", src,
                      "-o", obj], obj))
    for src in SOURCES:
        obj = os.path.splitext(src)[0] + suffix
        # ML2048_SCHED=<strategy> (A/B only) applies a scheduling strategy to whole files; the default build uses
        # max-ilp for the plain rollout kernels alone (the extra passes below)
        sched = os.environ.get("ML2048_SCHED", "default")
        sched_flags = [] if sched == "default" else ["-mllvm", f"-amdgpu-sched-strategy={sched}"]
        jobs.append(([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *sched_flags, *extra_hipcc_flags,
                      "-c", src, "-o", obj], obj))
        if os.path.basename(src) == "ml2048_hip.hip":
            # three more passes over the same file: only the plain rollout instantiations and their launcher - the
            # lone-wave forms for small M scheduled with max-ilp (masked sampling, rocRAND mode) and with the default
            # strategy (unmasked table mode: 2 % faster at C2), and the general forms for large M (default strategy,
            # 3 % faster at 1 M boards than max-ilp); see ml2048_plain_launch in the source
            ilp = os.environ.get("ML2048_SCHED_PLAIN", "max-ilp")  # A/B: another strategy for the small-M max-ilp pass
            large = os.environ.get("ML2048_SCHED_LARGE", "default")  # A/B: a strategy for the large-M pass
            large_flags = [] if large == "default" else ["-mllvm", f"-amdgpu-sched-strategy={large}"]
            for tag, flags in (("ilp", ["-mllvm", f"-amdgpu-sched-strategy={ilp}", "-DML2048_LONE_WAVE=1"]),
                               ("lone", ["-DML2048_LONE_WAVE=1"]), ("dflt", large_flags)):
                obj2 = os.path.splitext(src)[0] + f".plain_{tag}" + suffix
                jobs.append(([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                              f"-DML2048_ROLLOUT_PLAIN_TU=ml2048_internal_launch_rollout_plain_{tag}", *flags,
                              *extra_hipcc_flags, "-c", src, "-o", obj2], obj2))

    def run(job):
        if verbose:
            print(" ".join(job[0]), flush=True)
        subprocess.check_call(job[0])
        return job[1]

    workers = max(1, min(len(jobs), int(os.environ.get("ML2048_BUILD_JOBS", str(min(4, os.cpu_count() or 1))))))
    with concurrent.futures.ThreadPoolExecutor(max_workers=workers) as pool:
        objs = list(pool.map(run, jobs))  # in job order: the link line does not depend on who finished first
    cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib_path, *objs, "-lgomp", "-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    # python -m ml2048_amd.build [--force] [--out variant.so -- <extra hipcc flags>]
    argv = sys.argv[1:]
    extra = argv[argv.index("--") + 1:] if "--" in argv else []
    out_path = argv[argv.index("--out") + 1] if "--out" in argv else None
    print(build(force="--force" in argv, verbose=True, extra_hipcc_flags=extra, out=out_path))
