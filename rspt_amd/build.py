"""Build the gfx950 shared library in-tree: rspt_amd/librspt_hip.so.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels to
the GPU box with the gpurun snapshot.  Staleness is decided by a content
fingerprint of the sources (kept beside the library), not by mtimes -- a snapshot
copy need not preserve those.  Concurrent callers (the ranks of a torchrun job)
serialise on a file lock and the library is published with an atomic rename, so
nobody ever maps a half-written file.
"""
import fcntl
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librspt_hip.so")
STAMP = LIB + ".src-sha"
SOURCES = ["rspt_hip.hip", "signal_packer_hip.cpp"]
DEPS = SOURCES + ["common.hpp", "preprocess.hip", "hzr_kernels.hip", "hzr_rows.hip", "transforms.hip", "decode.hip", "filter.hip"]
INCLUDES = [os.path.join(os.path.dirname(HERE), "include", f) for f in ("rspt_hip.h", "signal_packer.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"]


def fingerprint():
    h = hashlib.sha256(" ".join(FLAGS + sorted(os.environ.get("RSPT_EXTRA_FLAGS", "").split())).encode())
    for p in [os.path.join(CSRC, f) for f in DEPS] + INCLUDES:
        if os.path.exists(p):
            h.update(os.path.basename(p).encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()


def stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != fingerprint()


def build(force=False, verbose=False):
    if "-DRSPT_DIAG" in os.environ.get("RSPT_EXTRA_FLAGS", "").split():
        # the diagnostic library (timing probes that skip work) must never become what api.lib() loads
        raise RuntimeError("rspt_amd.build: -DRSPT_DIAG does not belong in RSPT_EXTRA_FLAGS; build the diagnostic library with "
                           "`python -m rspt_amd.build --diag` and load it through RSPT_HIP_LIB")
    if not force and not stale():
        return LIB
    with open(LIB + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not force and not stale():  # another process built it while we waited
                return LIB
            fp = fingerprint()
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = "%s.tmp.%d" % (LIB, os.getpid())
            cmd = [hipcc] + FLAGS + os.environ.get("RSPT_EXTRA_FLAGS", "").split() + ["-o", tmp]
            cmd += [os.path.join(CSRC, f) for f in SOURCES]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
            with open(STAMP + ".tmp", "w") as f:
                f.write(fp + "\n")
            os.replace(STAMP + ".tmp", STAMP)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB


DIAG_LIB = os.path.join(HERE, "librspt_hip_diag.so")


def build_diag(verbose=False):
    """The diagnostic build (-DRSPT_DIAG: timing probes that skip work, tuning knobs from the environment) never replaces the
    product library: it goes to librspt_hip_diag.so, which is only ever loaded through RSPT_HIP_LIB.  Same care as build():
    a content stamp, a file lock, publication by atomic rename."""
    flags = [f for f in os.environ.get("RSPT_EXTRA_FLAGS", "").split() if f != "-DRSPT_DIAG"]
    stamp = DIAG_LIB + ".src-sha"
    h = hashlib.sha256(("diag " + " ".join(FLAGS + sorted(flags))).encode())
    for p in [os.path.join(CSRC, f) for f in DEPS] + INCLUDES:
        if os.path.exists(p):
            h.update(os.path.basename(p).encode())
            h.update(open(p, "rb").read())
    fp = h.hexdigest()

    def fresh():
        return os.path.exists(DIAG_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == fp

    if fresh():
        return DIAG_LIB
    with open(DIAG_LIB + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if fresh():
                return DIAG_LIB
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = "%s.tmp.%d" % (DIAG_LIB, os.getpid())
            cmd = [hipcc] + FLAGS + ["-DRSPT_DIAG"] + flags + ["-o", tmp] + [os.path.join(CSRC, f) for f in SOURCES]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, DIAG_LIB)
            with open(stamp + ".tmp", "w") as f:
                f.write(fp + "\n")
            os.replace(stamp + ".tmp", stamp)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return DIAG_LIB


if __name__ == "__main__":
    print(build_diag(verbose=True) if "--diag" in sys.argv else build(force="--force" in sys.argv, verbose=True))
