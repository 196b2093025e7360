"""Developer tool: phase times of the per-view pre-pass workgroups (diagnostic build, s_memtime stamps)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvtn_amd import build as b
diag = os.path.join(ROOT, "mvtn_amd", "libmvr_hip_diag.so")
b.build_hip(out=diag, extra_flags=["-DMVR_DIAG"] + sys.argv[1:])
os.environ["MVR_LIB"] = diag
import numpy as np, torch
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, _lib, synthetic as syn
lib = _lib.load()
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
TIER = os.environ.get("MVR_TIER", "M")
B = 8 if TIER == "L" else B
V = 20 if TIER == "L" else V
vl, fl = zip(*[syn.synthetic_mesh(TIER, i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="fixed", rotation_check="off").eval()
for _ in range(3): r(packed, None, ta, te, td)
buf = torch.zeros((1 << 20) + 8 * 4096, dtype=torch.int64, device=dev)
lib.mvr_diag_set_buffer.argtypes = [ctypes.c_void_p]
lib.mvr_diag_set_buffer(buf.data_ptr())
r(packed, None, ta, te, td)
torch.cuda.synchronize()
dall = buf.cpu().numpy()[(1 << 20):].reshape(-1, 8)
d = dall[:B * V]
if (d[:, 6] > 0).all():   # finer stamps inside the fused look-at
    x = dall[2048:2048 + B * V, 0]
    print("look-at: start -> angles loaded %.0f, sincos %.0f, look-at + validity %.0f, publish + barriers %.0f cycles" % (
        (d[:, 6] - d[:, 0]).mean(), (d[:, 7] - d[:, 6]).mean(), (x - d[:, 7]).mean(), (d[:, 5] - x).mean()))
names = ["project verts", "faces: bins + histogram", "scan", "fill lists"]
for k, nm in enumerate(names):
    dt = (d[:, k + 1] - d[:, k]).astype(np.float64)
    print("%-26s mean %7.0f cycles  max %7.0f" % (nm, dt.mean(), dt.max()))
print("total mean %.0f cycles" % (d[:, 4] - d[:, 0]).mean())
if (d[:, 5] > 0).all():
    print("  of which fused look-at (start -> R, T ready): mean %.0f cycles" % (d[:, 5] - d[:, 0]).mean())
