"""Manual stress run (not collected by pytest): the regulariser's window rendered by the rasterisers (functional.with_window)
against the plain render followed by regularize.apply_regularization with the same draws — random meshes / views / image
sizes / strip heights / crop ratios / dropout, both rasterisers, z-clip on and off: identical pixel -> face maps seen
through the window, identical frames from the exact kernel (an ulp from the fast one), gradients to round-off.
    python tests/stress_window.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvtn_amd import functional as F, regularize as rg, synthetic as syn  # noqa: E402
from mvtn_amd.structures import Meshes, PackedMeshes  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
torch.manual_seed(seed)
dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
bad = 0
for case in range(n_cases):
    B, V = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    size = int(rng.choice([32, 45, 64, 96, 100, 128, 224]))
    strip = int(rng.choice([0, 0, 0, 3, 5, 8]))
    exact = bool(rng.integers(0, 2))
    cull = bool(rng.integers(0, 4) == 0)
    crop = float(rng.choice([0.0, 0.1, 0.3, 0.6]))
    drop = float(rng.choice([0.0, 0.3]))
    zc = 0.5 if rng.integers(0, 2) else 0.0
    vl, fl = [], []
    for b in range(B):
        kind = rng.integers(0, 4)
        if kind == 0:
            v, f = syn.cad_soup(int(rng.integers(0, 1000)))
        else:
            v, f = syn.synthetic_mesh(str(rng.choice(["S", "M", "M", "L"])), int(rng.integers(0, 1000)))
        v = v * rng.uniform(0.5, 1.3, 3).astype(np.float32) + rng.uniform(-0.2, 0.2, 3).astype(np.float32)
        vl.append(torch.from_numpy(v.astype(np.float32))); fl.append(torch.from_numpy(f))
    az = rng.uniform(-180, 180, (B, V)).astype(np.float32)
    el = rng.uniform(-80, 80, (B, V)).astype(np.float32)
    di = rng.uniform(0.9, 3.0, (B, V)).astype(np.float32)   # close cameras: objects reach the border, faces get cut
    obj, bg = (0.9, 0.8, 0.7), (0.1, 0.2, 0.3)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)
    draws = rg.draw_regularization(B, V, size, dev, dropout_p=drop, augment_training=True, crop_ratio=crop)
    scale, flip, ci, cj, pad = draws
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(case)).to(dev)
    res = {}
    for fused in (False, True):
        params = F.mesh_params(V, size, 1, obj, bg, cull_backfaces=cull, strip_rows=strip, z_clip=zc, exact_raster=exact)
        if fused:
            params = F.with_window(params, *draws)
        R, T = F.look_at_view_transform(t(di.reshape(-1)), t(el.reshape(-1)), t(az.reshape(-1)))
        R, T = R.detach().requires_grad_(True), T.detach().requires_grad_(True)
        img = F.render_meshes(packed, R, T, t([[0, 1, 0]]), params)
        if not fused:
            img = rg.apply_regularization(img, *draws)
        img.backward(G)
        _, p2f = F.render_meshes_fragments(packed, R.detach(), T.detach(), t([[0, 1, 0]]), params)
        res[fused] = (img.detach(), p2f, R.grad.double(), T.grad.double())
    n_idx = torch.arange(B * V, device=dev).view(-1, 1, 1)
    rows = (torch.arange(size, device=dev).view(1, -1, 1) + (ci.view(-1, 1, 1) - pad)).clamp(0, size - 1)
    cols = (torch.arange(size, device=dev).view(1, 1, -1) + (cj.view(-1, 1, 1) - pad)).clamp(0, size - 1)
    cols = torch.where(flip.view(-1, 1, 1) != 0, size - 1 - cols, cols)
    nd = int((res[True][1] != res[False][1][n_idx, rows, cols]).sum())
    err = float((res[True][0] - res[False][0]).abs().max())
    gerr = max(float((a - b).abs().max() / (b.abs().max() + 1.0)) for a, b in ((res[True][2], res[False][2]), (res[True][3], res[False][3])))
    ok = nd == 0 and (err == 0.0 if exact else err <= 1e-6) and gerr <= 3e-4   # (sums of up to (pad+1)^2 replicated pixels in another order)
    bad += not ok
    print("case %2d: B=%d V=%d size=%3d strip=%d exact=%d cull=%d zclip=%.1f crop=%.1f pad=%d drop=%.1f faces=%s covered=%.2f  map diffs %d  max|d| %.1e  grads %.1e%s"
          % (case, B, V, size, strip, exact, cull, zc, crop, pad, drop, [len(f) for f in fl], float((res[True][1] >= 0).float().mean()), nd, err, gerr,
             "" if ok else "  <-- MISMATCH"))
print("cases with mismatches:", bad)
sys.exit(1 if bad else 0)
