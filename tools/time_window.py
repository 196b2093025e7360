"""Developer tool: the C2 forward followed by the reference's view regulariser (view dropout 0.3 + flip + replicate-pad + random
crop) and the trainer's Normalize — as the fused pass over the frames (regularize.apply_regularization with mean / std) and as
the rasteriser's own sampling window + store epilogue (MVRenderer(normalize=...)(..., window=draws)).  Same draws in both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn, regularize as rg
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
kw = dict(pc_rendering=False, object_color="white", background_color="white", light_direction="fixed")
plain, fused = MVRenderer(V, H, **kw).eval(), MVRenderer(V, H, normalize=(rg.IMAGENET_MEAN, rg.IMAGENET_STD), **kw).eval()
torch.manual_seed(1)
draws = rg.draw_regularization(B, V, H, dev, dropout_p=0.3, augment_training=True, crop_ratio=0.3)
def run(f, n=600, w=100):
    with torch.no_grad():
        for _ in range(w): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
two = lambda: rg.apply_regularization(plain(packed, None, ta, te, td)[0], *draws, mean=rg.IMAGENET_MEAN, std=rg.IMAGENET_STD)
one = lambda: fused(packed, None, ta, te, td, window=draws)[0]
a = run(lambda: plain(packed, None, ta, te, td)[0])
b = run(two)
c = run(one)
with torch.no_grad():
    d = (two() - one()).abs().max().item()
print("C2 forward %.1f us; + regulariser and Normalize as one pass over the frames %.1f us; both inside the rasteriser %.1f us; max difference %.1e" % (a, b, c, d))
