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
    # [0] the kernel translation unit (it #includes the mvr_*.inc sections), [3] the host packer, [4] the regulariser;
    # the rest only take part in the staleness check
    import glob
    return [os.path.join(CSRC, "mvr_kernels.hip"), os.path.join(CSRC, "mvr_math.h"), inc, os.path.join(CSRC, "mvr_pack.cpp"),
            os.path.join(CSRC, "mvr_regularize.hip")] + sorted(glob.glob(os.path.join(CSRC, "*.inc")))


def build_hip(force=False, verbose=False, out=None, extra_flags=()):
    """hipcc --offload-arch=gfx950 ... -> mvtn_amd/libmvr_hip.so.  `out` + `extra_flags` (-D switches): developer
    variants for tools/ (ablation / diagnostic builds loaded through MVR_LIB), never the product library."""
    srcs = _sources()
    if out is not None:
        return _compile(srcs, out, list(extra_flags), verbose)
    if not force and not _stale(LIB_PATH, srcs):
        return LIB_PATH
    return _compile(srcs, LIB_PATH, [], verbose)


def _compile(srcs, LIB_PATH, extra, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # the host-only mesh packer is plain C++: g++ builds the object (function multi-versioning for its
    # narrowing loop), hipcc links it into the one library
    pack_o = LIB_PATH + ".pack.o"
    cmds = [["g++", "-O3", "-std=c++17", "-fPIC", "-fvisibility-inlines-hidden", "-c", "-o", pack_o, srcs[3]],
            [hipcc] + HIPCC_FLAGS + extra + ["-o", LIB_PATH, srcs[0], srcs[4], "-Wl," + pack_o]]
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
    # python mvtn_amd/build.py                      -> the product library
    # python mvtn_amd/build.py OUT.so -DFOO -DBAR=1 -> a developer variant (see tools/)
    import sys
    if len(sys.argv) > 1:
        print(build_hip(out=sys.argv[1], extra_flags=sys.argv[2:], verbose=True))
    else:
        print(build_hip(force=True, verbose=True))
