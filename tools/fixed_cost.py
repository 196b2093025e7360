"""Where a 20-step timed region's fixed cost goes (developer tool): repeated K-step regions of the C2 forward with the time
at which the first step returned to Python, the time all K were issued and the time the closing synchronize returned."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn
dev = torch.device("cuda", 0)
B, V, H = 16, 12, 224
az, el, di = syn.spherical_views(V)
azim = torch.from_numpy(np.tile(az, (B, 1))).to(dev); elev = torch.from_numpy(np.tile(el, (B, 1))).to(dev); dist = torch.from_numpy(np.tile(di, (B, 1))).to(dev)
vl, fl = [], []
for i in range(B):
    v, f = syn.synthetic_mesh("M", i); vl.append(torch.from_numpy(v)); fl.append(torch.from_numpy(f))
packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)
r = MVRenderer(V, H, pc_rendering=False, object_color="white", background_color="white", light_direction="fixed").eval()
def step():
    with torch.no_grad():
        return r(packed, None, azim, elev, dist)[0]
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50): step()
for K in (20, 20, 20, 1000, 20):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    for _ in range(K - 1): step()
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("K %4d  per-step %.1f us | first step returned %.1f us, all issued %.1f us (%.1f/step), sync returned +%.1f us" % (
        K, (t3 - t0) / K * 1e6, (t1 - t0) * 1e6, (t2 - t0) * 1e6, (t2 - t0) / K * 1e6, (t3 - t2) * 1e6))
