"""Host cost of handing CPU meshes to every forward (models/renderer.py:66-77): PackedMeshes.from_meshes alone,
its C packer alone, and a full forward with CPU meshes vs HBM-resident ones.  Run on the GPU box."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mvtn_amd import MVRenderer, Meshes, _lib, synthetic
from mvtn_amd.structures import PackedMeshes

dev = torch.device("cuda:0")
B, V = 16, 12
vl, fl = zip(*[synthetic.synthetic_mesh("M", s) for s in range(B)])
meshes = [Meshes([torch.from_numpy(v)], [torch.from_numpy(f)]) for v, f in zip(vl, fl)]
az, el = synthetic.unit_spherical_grid(V)
azim = torch.tensor(az, dtype=torch.float32, device=dev)[None].expand(B, V)
elev = torch.tensor(el, dtype=torch.float32, device=dev)[None].expand(B, V)
dist = torch.full((B, V), 2.2, device=dev)


def t(f, n=300):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("from_meshes (pack + H2D, pipelined)  %.1f us" % t(lambda: PackedMeshes.from_meshes(meshes, dev)))
lib = _lib.load()
n = B
vs = [m.verts_list()[0] for m in meshes]
fs = [m.faces_list()[0] for m in meshes]
nv = (ctypes.c_int32 * n)(*[v.shape[0] for v in vs])
nf = (ctypes.c_int32 * n)(*[f.shape[0] for f in fs])
lay = _lib.PackLayout()
lib.mvr_pack_meshes_layout(n, nv, nf, ctypes.byref(lay))
vp = (ctypes.c_void_p * n)(*[v.data_ptr() for v in vs])
fp = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fs])
blk = torch.empty(int(lay.bytes), dtype=torch.uint8, pin_memory=True)
print("mvr_pack_meshes_host alone           %.1f us" % t(lambda: lib.mvr_pack_meshes_host(ctypes.byref(lay), vp, nv, fp, nf, blk.data_ptr())))
print("H2D of the block alone               %.1f us" % t(lambda: blk.to(dev, non_blocking=True)))
r = MVRenderer(V, 224, pc_rendering=False, object_color="white", background_color="white", light_direction="fixed").eval()
packed = PackedMeshes.from_meshes(meshes, dev)
with torch.no_grad():
    print("forward, resident meshes             %.1f us" % t(lambda: r(packed, None, azim, elev, dist)))
    print("forward, CPU meshes every call       %.1f us" % t(lambda: r(meshes, None, azim, elev, dist)))
