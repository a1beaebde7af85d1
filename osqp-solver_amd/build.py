"""Build the C-ABI shared library (hipcc, gfx950 only) in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi_osqp.so")
SOURCES = ["solver.hip", "kernels.hip", "host_core.cpp"]
HEADERS = ["device_types.h", "sched_format.h", "host_core.hpp", os.path.join("..", "..", "include", "mi_osqp.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def _compile(lib, obj_suffix, extra_flags, verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for f in SOURCES:
        obj = os.path.join(CSRC, f + obj_suffix)
        cmd = [hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result",
               "-c", os.path.join(CSRC, f), "-o", obj] + extra_flags
        if f.endswith(".cpp"):
            cmd.insert(1, "-x"); cmd.insert(2, "c++")
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs + ["-lpthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    return _compile(LIB, ".o", os.environ.get("MI_OSQP_CXXFLAGS", "").split(), verbose)


# Diagnostic twin of the library (-DMI_OSQP_DEBUG_BUILD: fault injection into the grid-spinning launches, timing
# experiments).  Test infrastructure: tests/test_gpu_faults.py loads it in a child process (MI_OSQP_LIBRARY); the product
# never does.
LIB_DEBUG = os.path.join(HERE, "libmi_osqp_debug.so")


def build_debug(force=False, verbose=False):
    if not force and os.path.exists(LIB_DEBUG):
        t = os.path.getmtime(LIB_DEBUG)
        if not any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS):
            return LIB_DEBUG
    return _compile(LIB_DEBUG, ".dbg.o", ["-DMI_OSQP_DEBUG_BUILD"], verbose)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--debug" in sys.argv:
        print(build_debug(force="--force" in sys.argv, verbose=True))
