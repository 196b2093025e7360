import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn
dev = torch.device("cuda:0")
B, V, H = 2, 12, 224
vl, fl = zip(*[syn.synthetic_mesh(6, i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="fixed").eval()
with torch.no_grad():
    for _ in range(10): r(packed, None, ta, te, td)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): r(packed, None, ta, te, td)
    torch.cuda.synchronize(); print("level-6 meshes (%d verts, %d faces) x %d views: %.1f us/step" % (len(vl[0]), len(fl[0]), B * V, (time.perf_counter() - t0) / 50 * 1e6))
