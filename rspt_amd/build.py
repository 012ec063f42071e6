"""Build the gfx950 shared library in-tree: rspt_amd/librspt_hip.so.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to
the GPU box with the gpurun snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librspt_hip.so")
SOURCES = ["rspt_hip.hip", "signal_packer_hip.cpp"]
DEPS = SOURCES + ["common.hpp", "preprocess.hip", "hzr_kernels.hip", "transforms.hip", "decode.hip"]
INCLUDES = [os.path.join(os.path.dirname(HERE), "include", f) for f in ("rspt_hip.h", "signal_packer.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in [os.path.join(CSRC, f) for f in DEPS] + INCLUDES)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-o", LIB]
    cmd += [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
