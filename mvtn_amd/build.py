"""Builds libmvr_hip.so (HIP kernels + C ABI, gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
import fcntl
import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libmvr_hip.so")
HOSTMATH_PATH = os.path.join(_HERE, "libmvr_hostmath.so")
TORCH_OP_PATH = os.path.join(_HERE, "_mvr_torch_op.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               # coverage / depth arithmetic must be the exact IEEE sequence written in mvr_math.h
               "-ffp-contract=off", "-fno-fast-math",
               # the SLP vectoriser pairs scalar f32 operations into v_pk_*_f32 and pays for it in register moves: measured
               # slower on every kernel it touched (k_mesh_backward_tab 82.8 -> 73.3 us, k_mesh_raster_fast 91.0 -> 88.4 us);
               # the hand-written two-pixel arithmetic (explicit float2 types) is not affected
               "-fno-slp-vectorize"]


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


def _src_hash(srcs, flags):
    """Content hash of everything the library is built from (+ the flags): staleness must not depend on file times — a
    snapshot copied to another machine keeps the bytes, not the mtimes."""
    h = hashlib.sha256(" ".join(flags).encode())
    for p in sorted(set(srcs)):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _hash_path(lib):
    return lib + ".srchash"


def source_hash():
    """The hash the product library carries (mvr_source_hash()) when it was built from the sources on disk."""
    return _src_hash(_sources(), HIPCC_FLAGS)


def is_current(lib=None):
    """True when `lib` (default: the product library) was built from exactly the sources on disk."""
    lib = lib or LIB_PATH
    try:
        if not os.path.exists(lib):
            return False
        with open(_hash_path(lib)) as f:
            return f.read().strip() == source_hash()
    except OSError:
        return False


def build_hip(force=False, verbose=False, out=None, extra_flags=()):
    """hipcc --offload-arch=gfx950 ... -> mvtn_amd/libmvr_hip.so.  `out` + `extra_flags` (-D switches): developer
    variants for tools/ (ablation / diagnostic builds loaded through MVR_LIB), never the product library.

    Safe under a multi-rank launch: the build happens under an exclusive file lock, into a per-process temporary that is
    renamed into place, so a rank never loads a half-written library and never deletes another rank's object file; the
    ranks that waited for the lock find the library current and do not build again."""
    srcs = _sources()
    if out is not None:
        return _compile(srcs, out, list(extra_flags), verbose)
    if not force and is_current():
        return LIB_PATH
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_current():   # another rank built it while this one waited
                return LIB_PATH
            _compile(srcs, LIB_PATH, [], verbose)
            tmp = "%s.tmp.%d" % (_hash_path(LIB_PATH), os.getpid())
            with open(tmp, "w") as f:
                f.write(_src_hash(srcs, HIPCC_FLAGS) + "\n")
            os.replace(tmp, _hash_path(LIB_PATH))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def _compile(srcs, LIB_PATH, extra, verbose):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # the host-only mesh packer is plain C++: g++ builds the object (function multi-versioning for its
    # narrowing loop), hipcc links it into the one library.  Everything is written under per-process names and the
    # finished library is renamed into place (atomic on POSIX).
    tmp_lib = "%s.tmp.%d" % (LIB_PATH, os.getpid())
    pack_o = "%s.pack.%d.o" % (LIB_PATH, os.getpid())
    cmds = [["g++", "-O3", "-std=c++17", "-fPIC", "-fvisibility-inlines-hidden", "-c", "-o", pack_o, srcs[3]],
            [hipcc] + HIPCC_FLAGS + extra + ['-DMVR_SRC_HASH="%s"' % _src_hash(srcs, HIPCC_FLAGS), "-o", tmp_lib, srcs[0], srcs[4],
                                               "-Wl," + pack_o]]
    try:
        for cmd in cmds:
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("%s failed:\n%s\n%s" % (os.path.basename(cmd[0]), " ".join(cmd), r.stderr))
            if verbose:
                print(" ".join(cmd))
        os.replace(tmp_lib, LIB_PATH)
    finally:
        for f in (pack_o, tmp_lib):
            if os.path.exists(f):
                os.remove(f)
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


def _torch_op_sources():
    return [os.path.join(CSRC, "mvr_torch_op.cpp"), os.path.join(os.path.dirname(_HERE), "include", "mvr.h")]


def torch_op_hash():
    """Content hash of the torch operator's sources + the torch build it is compiled against (its C++ ABI is torch's)."""
    import torch
    # (the version string alone does not pin the C++ ABI: a rebuild of the same version with another libstdc++ ABI or from
    # another commit would otherwise reuse a stale operator)
    return _src_hash(_torch_op_sources(), ["torch-" + torch.__version__, "cxx11abi-%s" % int(bool(torch._C._GLIBCXX_USE_CXX11_ABI)),
                                           "git-%s" % getattr(torch.version, "git_version", ""), "hip-%s" % getattr(torch.version, "hip", "")])


def torch_op_is_current():
    try:
        with open(_hash_path(TORCH_OP_PATH)) as f:
            return os.path.exists(TORCH_OP_PATH) and f.read().strip() == torch_op_hash()
    except OSError:
        return False


def build_torch_op(force=False):
    """g++ -> mvtn_amd/_mvr_torch_op.so: host-side plumbing (csrc/mvr_torch_op.cpp: the mesh step as one torch operator with a
    C++ autograd node).  No kernels, no link against libmvr_hip.so (the entry points are handed over as addresses); links
    against the torch of this interpreter.  Same lock / rename discipline as build_hip."""
    if not force and torch_op_is_current():
        return TORCH_OP_PATH
    from torch.utils import cpp_extension as ce
    import torch
    with open(TORCH_OP_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and torch_op_is_current():
                return TORCH_OP_PATH
            tmp = "%s.tmp.%d" % (TORCH_OP_PATH, os.getpid())
            cmd = (["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)] +
                   ["-I" + d for d in ce.include_paths()] + [_torch_op_sources()[0], "-o", tmp] +
                   ["-L" + d for d in ce.library_paths()] + ["-ltorch", "-ltorch_cpu", "-lc10"])
            try:
                r = subprocess.run(cmd, capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("g++ failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
                os.replace(tmp, TORCH_OP_PATH)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
            htmp = "%s.tmp.%d" % (_hash_path(TORCH_OP_PATH), os.getpid())
            with open(htmp, "w") as f:
                f.write(torch_op_hash() + "\n")
            os.replace(htmp, _hash_path(TORCH_OP_PATH))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return TORCH_OP_PATH


if __name__ == "__main__":
    # python mvtn_amd/build.py                      -> the product library
    # python mvtn_amd/build.py OUT.so -DFOO -DBAR=1 -> a developer variant (see tools/)
    import sys
    if len(sys.argv) > 1:
        print(build_hip(out=sys.argv[1], extra_flags=sys.argv[2:], verbose=True))
    else:
        print(build_hip(force=True, verbose=True))
