"""Developer tool: host enqueue time of one C3 step with the autograd engine's worker threads on (default) and off
(torch.autograd.set_multithreading_enabled(False): the backward then runs on the calling thread)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev).clone().requires_grad_(True) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="relative").eval()
up = torch.randn(B, V, 3, H, H, device=dev)
def step():
    ta.grad = te.grad = td.grad = None
    img, _ = r(packed, None, ta, te, td)
    img.backward(up)
def measure(tag):
    for _ in range(30): step()
    torch.cuda.synchronize()
    t = 0.0
    for _ in range(30):
        t0 = time.perf_counter()
        for _ in range(10): step()
        t += time.perf_counter() - t0
        torch.cuda.synchronize()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-28s host enqueue %.1f us / step, wall %.1f us / step" % (tag, t / 300 * 1e6, dt / 500 * 1e6))
measure("default")
with torch.autograd.set_multithreading_enabled(False):
    measure("multithreading off")
measure("default again")
