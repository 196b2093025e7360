"""Developer tool: host-side cost split of one training step (forward call, backward call as seen by the autograd
engine, and the Python body of the fused node's backward)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, functional as F, synthetic as syn
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev).requires_grad_(True) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="relative").eval()
up = torch.randn(B, V, 3, H, H, device=dev)
tb = [0.0]
orig = F._RenderMeshViews.backward
def timed(ctx, *g):
    t0 = time.perf_counter(); out = orig(ctx, *g); tb[0] += time.perf_counter() - t0; return out
F._RenderMeshViews.backward = staticmethod(timed)
tf = tbw = 0.0
for it in range(330):
    if it == 30: tf = tbw = 0.0; tb[0] = 0.0
    ta.grad = te.grad = td.grad = None
    t0 = time.perf_counter()
    img, _ = r(packed, None, ta, te, td)
    t1 = time.perf_counter()
    img.backward(up)
    t2 = time.perf_counter()
    tf += t1 - t0; tbw += t2 - t1
    if it % 10 == 9: torch.cuda.synchronize()
print("forward call %.1f us | backward call %.1f us (python body of the node's backward %.1f us)" % (tf / 300 * 1e6, tbw / 300 * 1e6, tb[0] / 300 * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for it in range(300):
    ta.grad = te.grad = td.grad = None
    img, _ = r(packed, None, ta, te, td)
    if it % 10 == 9: torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
