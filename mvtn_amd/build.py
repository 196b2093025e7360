"""Builds libmvr_hip.so (HIP kernels + C ABI, gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libmvr_hip.so")
HOSTMATH_PATH = os.path.join(_HERE, "libmvr_hostmath.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               # coverage / depth arithmetic must be the exact IEEE sequence written in mvr_math.h
               "-ffp-contract=off", "-fno-fast-math"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _sources():
    inc = os.path.join(os.path.dirname(_HERE), "include", "mvr.h")
    return [os.path.join(CSRC, "mvr_kernels.hip"), os.path.join(CSRC, "mvr_math.h"), inc, os.path.join(CSRC, "mvr_pack.cpp"),
            os.path.join(CSRC, "mvr_regularize.hip")]


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 ... -> mvtn_amd/libmvr_hip.so"""
    srcs = _sources()
    if not force and not _stale(LIB_PATH, srcs):
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # the host-only mesh packer is plain C++: g++ builds the object (function multi-versioning for its
    # narrowing loop), hipcc links it into the one library
    pack_o = os.path.join(CSRC, "mvr_pack.o")
    cmds = [["g++", "-O3", "-std=c++17", "-fPIC", "-fvisibility-inlines-hidden", "-c", "-o", pack_o, srcs[3]],
            [hipcc] + HIPCC_FLAGS + ["-o", LIB_PATH, srcs[0], srcs[4], "-Wl," + pack_o]]
    try:
        for cmd in cmds:
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("%s failed:\n%s\n%s" % (os.path.basename(cmd[0]), " ".join(cmd), r.stderr))
            if verbose:
                print(" ".join(cmd))
    finally:
        if os.path.exists(pack_o):
            os.remove(pack_o)
    return LIB_PATH


def build_hostmath(force=False):
    """TEST-ONLY host build (g++) of the templated per-pixel math in csrc/mvr_math.h, in float and
    double, so tests can finite-difference the analytic backward without a GPU."""
    srcs = [os.path.join(CSRC, "mvr_hostmath.cpp"), os.path.join(CSRC, "mvr_math.h")]
    if not force and not _stale(HOSTMATH_PATH, srcs):
        return HOSTMATH_PATH
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", HOSTMATH_PATH, srcs[0], "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return HOSTMATH_PATH


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
