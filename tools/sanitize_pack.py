"""Part of tools/sanitize_cpu.sh: the host-side mesh packer (csrc/mvr_pack.cpp) built with ASan + UBSan into a library of its own
(argv[1]) and driven like tests/test_host_logic.py::test_pack_meshes_host_c_abi — ragged meshes whose staging block is allocated
EXACTLY (any byte written past a section is a heap overflow the sanitizer reports), legal and illegal indices."""
import ctypes
import sys
import numpy as np


class Layout(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int32) for k in ("n_meshes", "total_verts", "total_faces", "max_verts", "max_faces", "reserved_")] + \
               [(k, ctypes.c_uint64) for k in ("verts_at", "faces_at", "vert_off_at", "face_off_at", "bytes")]


lib = ctypes.CDLL(sys.argv[1])
rng = np.random.default_rng(0)
libc = ctypes.CDLL(None)
libc.malloc.restype = ctypes.c_void_p
libc.malloc.argtypes = [ctypes.c_size_t]
libc.free.argtypes = [ctypes.c_void_p]
for trial in range(200):
    n = int(rng.integers(1, 9))
    sizes = [(int(rng.integers(1, 400)), int(rng.integers(1, 900))) for _ in range(n)]
    vs = [rng.normal(size=(nv, 3)).astype(np.float32) for nv, _ in sizes]
    fs = [rng.integers(0, nv, size=(nf, 3)).astype(np.int64) for nv, nf in sizes]
    nv = (ctypes.c_int32 * n)(*[s[0] for s in sizes])
    nf = (ctypes.c_int32 * n)(*[s[1] for s in sizes])
    lay = Layout()
    assert lib.mvr_pack_meshes_layout(n, nv, nf, ctypes.byref(lay)) == 0
    block = libc.malloc(lay.bytes)            # exactly the bytes the layout asks for, from the sanitizer's allocator
    vp = (ctypes.c_void_p * n)(*[v.ctypes.data for v in vs])
    fp = (ctypes.c_void_p * n)(*[f.ctypes.data for f in fs])
    assert lib.mvr_pack_meshes_host(ctypes.byref(lay), vp, nv, fp, nf, ctypes.c_void_p(block)) == 0
    got = np.ctypeslib.as_array((ctypes.c_uint8 * lay.bytes).from_address(block)).copy()
    assert np.array_equal(got[lay.faces_at:lay.faces_at + lay.total_faces * 12].view(np.int32).reshape(-1, 3), np.concatenate(fs).astype(np.int32))
    assert np.array_equal(got[lay.verts_at:lay.verts_at + lay.total_verts * 12].view(np.float32).reshape(-1, 3), np.concatenate(vs))
    b = int(rng.integers(0, n))
    g = fs[b].copy()
    g[int(rng.integers(0, len(g))), int(rng.integers(0, 3))] = int(rng.choice([-1, sizes[b][0], 2 ** 31, 2 ** 40 + 3, -2 ** 33]))
    fp2 = (ctypes.c_void_p * n)(*[(g if k == b else fs[k]).ctypes.data for k in range(n)])
    assert lib.mvr_pack_meshes_host(ctypes.byref(lay), vp, nv, fp2, nf, ctypes.c_void_p(block)) == 1
    libc.free(block)
print("packer under ASan + UBSan: 200 ragged batches, legal and illegal indices: ok")
