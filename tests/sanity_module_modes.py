"""Manual sanity run (not collected by pytest): MVRenderer in train mode across light / colour / culling / point modes,
forward + backward, list-of-single-mesh input as collate_fn yields it.  python tests/sanity_module_modes.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from mvtn_amd import MVRenderer, Meshes, synthetic as syn
from mvtn_amd.views import view_selector
dev = torch.device("cuda:0")
B, V = 3, 6
vl, fl = zip(*[syn.synthetic_mesh("S", i) for i in range(B)])
meshes = [Meshes([torch.from_numpy(v)], [torch.from_numpy(f)]) for v, f in zip(vl, fl)]   # as collate_fn yields
pts = torch.from_numpy(np.stack([syn.surface_points(v, f, 512, seed=i) for i, (v, f) in enumerate(zip(vl, fl))]))
az = torch.linspace(-180, 150, V).repeat(B, 1).to(dev).requires_grad_(True)
el = torch.full((B, V), 25.0, device=dev, requires_grad=True)
di = torch.full((B, V), 2.2, device=dev, requires_grad=True)
for kw in (dict(light_direction="random"), dict(light_direction="relative"), dict(light_direction="fixed", object_color="random"),
           dict(object_color="custom"), dict(cull_backfaces=True, perspective_correct=False)):
    r = MVRenderer(V, 96, pc_rendering=False, **kw).train()
    color = torch.rand(B, 642, 3) if kw.get("object_color") == "custom" else None
    img, cams = r(meshes, None, az, el, di, color=color)
    img.mean().backward()
    assert img.shape == (B, V, 3, 96, 96) and torch.isfinite(img).all() and torch.isfinite(az.grad).all()
    print("mesh", kw, float(img.mean()), float(az.grad.abs().sum()))
    az.grad = el.grad = di.grad = None
for kw in (dict(points_per_pixel=1), dict(points_per_pixel=3, points_radius=0.03), dict(object_color="custom", points_radius=0.02)):
    r = MVRenderer(V, 96, pc_rendering=True, **kw).train()
    color = torch.rand(B, 512, 3) if kw.get("object_color") == "custom" else None
    d1 = torch.full((B, V), 1.0, device=dev, requires_grad=True)
    img, cams = r(None, pts, az, el, d1, color=color)
    img.mean().backward()
    assert img.shape == (B, V, 3, 96, 96) and torch.isfinite(img).all()
    print("points", kw, float(img.mean()))
    az.grad = el.grad = None
r.check_status()
print("ok")
