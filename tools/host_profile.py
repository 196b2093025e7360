"""Developer tool: where does the HOST spend its time in one C3 step (forward + backward through autograd)?
cProfile over 300 steps with the GPU queue kept short (one sync per step would hide nothing: we want the
CPU-side cost of enqueueing a step).  Usage: python tools/host_profile.py [fwd|fwdbwd]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fwdbwd"
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="relative" if mode == "fwdbwd" else "fixed").eval()
if mode == "fwdbwd":
    ta, te, td = (x.clone().requires_grad_(True) for x in (ta, te, td))
    up = torch.randn(B, V, 3, H, H, device=dev)


def step():
    if mode == "fwdbwd":
        ta.grad = te.grad = td.grad = None
        img, _ = r(packed, None, ta, te, td)
        img.backward(up)
    else:
        with torch.no_grad():
            r(packed, None, ta, te, td)


for _ in range(20):
    step()
torch.cuda.synchronize()
# pure host cost: enqueue-only timing with a sync every 10 steps so the queue never blocks
t = 0.0
for _ in range(30):
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    t += time.perf_counter() - t0
    torch.cuda.synchronize()
print("host enqueue time per step: %.1f us" % (t / 300 * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(30):
    for _ in range(10):
        step()
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
