"""Builds bvh_amd/lib/libbvh_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m bvh_amd.build [--force] [--developer]

`--developer` builds bvh_amd/lib/libbvh_amd_dev.so from the same sources with -DBVH_AMD_DEVELOPER: the A/B switches, profiling
aids and fault-injection knobs (csrc/common.h: BVH_DEV_*) exist only there. The release library reads the documented environment
variables and nothing else; the fault-injection tests load the developer library (BVH_AMD_LIB).

Flags that are part of correctness: -ffp-contract=off (fma only where the source
says so) and correctly rounded fp32 divide/sqrt. hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libbvh_amd.so")
DEV_LIB = os.path.join(LIBDIR, "libbvh_amd_dev.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, developer: bool = False) -> str:
    lib_path = DEV_LIB if developer else LIB
    objdir = os.path.join(OBJDIR, "dev") if developer else OBJDIR
    flags = FLAGS + (["-DBVH_AMD_DEVELOPER"] if developer else [])
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc"))) + [os.path.join(os.path.dirname(HERE), "include", "bvh_amd.h")]
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [HIPCC] + flags + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return s, r

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for s, r in ex.map(compile_one, jobs):
                if verbose or r.returncode:
                    sys.stderr.write(r.stdout + r.stderr)
                if r.returncode:
                    raise RuntimeError(f"hipcc failed on {s}")
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if jobs or _stale(lib_path, objs):
        # (librccl — the scene broadcast of the multi-GPU path, csrc/replicate.hip — is opened with dlopen on first use: a program
        #  that stays on one GPU loads this library without it)
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, developer="--developer" in sys.argv))
