"""Developer tool: phase times of k_mesh_backward_tab (-DMVR_DIAG build: per-workgroup s_memtime stamps of thread 0), C3
workload.     python3 tools/diag_bwd.py [-D...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
diag = os.path.join(ROOT, "mvtn_amd", "libmvr_hip_diag.so")
from mvtn_amd import build as b  # noqa: E402
b.build_hip(out=diag, extra_flags=["-DMVR_DIAG"] + sys.argv[1:])
os.environ["MVR_LIB"] = diag
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, _lib, synthetic as syn  # noqa: E402
lib = _lib.load()
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev).requires_grad_(True) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="fixed", rotation_check="off").train()
def step():
    img, _ = r(packed, None, ta, te, td)
    img.square().sum().backward()
for _ in range(3):
    step()
nblocks = 8 * ((B * V + 7) // 8) * 28
buf = torch.zeros((1 << 20) + 8 * 4096, dtype=torch.int64, device=dev)
lib.mvr_diag_set_buffer.argtypes = [ctypes.c_void_p]
lib.mvr_diag_set_buffer(buf.data_ptr())
step()
torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(-1, 8)[(1 << 18) // 8:(1 << 18) // 8 + nblocks]
d = d[d[:, 7] > 0]
ncov, nwin = (d[:, 1] & 0xffffffff).astype(np.float64), (d[:, 1] >> 32).astype(np.float64)
d = d.astype(np.float64)
tot = d[:, 7] - d[:, 0]
print("workgroups with covered pixels: %d   cycles/workgroup mean %.0f max %.0f; covered px mean %.0f max %.0f; winners mean %.0f max %.0f" % (
    len(d), tot.mean(), tot.max(), ncov.mean(), ncov.max(), nwin.mean(), nwin.max()))
for nme, a_, b_ in [("init + compaction + mask", 0, 2), ("mask prefix", 2, 3), ("slots", 3, 4), ("constants + first table", 4, 5),
                    ("pixel loop (+ later tables)", 5, 6), ("reduce + store", 6, 7)]:
    sp = d[:, b_] - d[:, a_]
    print("  %-32s %5.1f %%   mean %7.0f cycles" % (nme, 100 * sp.sum() / tot.sum(), sp.mean()))
