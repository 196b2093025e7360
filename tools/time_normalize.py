"""Developer tool: the C2 forward followed by the trainer's Normalize — as a separate fused pass over the frames
(mvtn_amd.regularize.normalize_views) and inside the rasteriser's stores (MVRenderer(normalize=...))."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn
from mvtn_amd.regularize import normalize_views, IMAGENET_MEAN, IMAGENET_STD
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
kw = dict(pc_rendering=False, object_color="white", background_color="white", light_direction="fixed")
plain, fused = MVRenderer(V, H, **kw).eval(), MVRenderer(V, H, normalize=(IMAGENET_MEAN, IMAGENET_STD), **kw).eval()
def run(f, n=600, w=100):
    with torch.no_grad():
        for _ in range(w): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
a = run(lambda: plain(packed, None, ta, te, td)[0])
b = run(lambda: normalize_views(plain(packed, None, ta, te, td)[0]))
c = run(lambda: fused(packed, None, ta, te, td)[0])
with torch.no_grad():
    d = (normalize_views(plain(packed, None, ta, te, td)[0]) - fused(packed, None, ta, te, td)[0]).abs().max().item()
print("C2 forward %.1f us; + Normalize as a pass over the frames %.1f us; Normalize in the rasteriser's stores %.1f us; max difference %.1e" % (a, b, c, d))
