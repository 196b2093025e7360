"""Writes visual artefacts: the 12 spherical views of a synthetic tier-M mesh, of its 2048-point cloud and
the camera plot, through MVRenderer.render_and_save — for human inspection.  Run on the GPU box:
    python tools/render_demo.py gpurun_out/demo"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtn_amd import MVRenderer, Meshes, synthetic as syn   # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/demo"
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda:0")
V, size = 12, 224
v, f = syn.synthetic_mesh("M", 0)
az, el, di = syn.spherical_views(V)
azim, elev, dist = (torch.from_numpy(x[None]).to(dev) for x in (az, el, di))
mesh = [Meshes([torch.from_numpy(v)], [torch.from_numpy(f)])]
for light in ("fixed", "relative"):
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction=light).eval()
    r.render_and_save(mesh, None, azim, elev, dist, os.path.join(out, "mesh_views_%s_light.png" % light),
                      os.path.join(out, "cameras.png"))
pts = torch.from_numpy(syn.surface_points(v, f, 2048, seed=0)[None])
rp = MVRenderer(V, size, pc_rendering=True, object_color="black", background_color="white", points_radius=0.006).eval()
rp.render_and_save(None, pts, azim, elev, torch.ones_like(dist), os.path.join(out, "point_views.png"), os.path.join(out, "cameras_points.png"))
rp8 = MVRenderer(V, size, pc_rendering=True, object_color="black", background_color="white", points_radius=0.02, points_per_pixel=4).eval()
rp8.render_and_save(None, pts, azim, elev, torch.ones_like(dist), os.path.join(out, "point_views_r0.02_k4.png"), os.path.join(out, "cameras_points.png"))
for fn in sorted(os.listdir(out)):
    print(fn, os.path.getsize(os.path.join(out, fn)))
