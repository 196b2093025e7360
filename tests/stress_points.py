"""Manual stress run (not collected by pytest): random point clouds / radii / points_per_pixel / views / image sizes /
strip heights through the C ABI against the oracle — slot lists bit-exact, colours to 2e-6.
    python tests/stress_points.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from mvtn_amd import functional as F  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
oracle.set_threads(8)
t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)   # noqa: E731
bad = 0
for case in range(n_cases):
    B, V = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    size = int(rng.choice([32, 45, 64, 100, 128, 224]))
    P = int(rng.choice([1, 7, 300, 1024, 2048, 2500, 5000]))
    K = int(rng.choice([1, 1, 1, 2, 3, 8]))
    strip = int(rng.choice([0, 0, 0, 3, 5, 8]))
    radius = float(rng.choice([0.006, 0.01, 0.03, 0.08]))
    kind = rng.integers(0, 3)
    if kind == 0:      # a shell
        pts = rng.normal(size=(B, P, 3))
        pts /= np.linalg.norm(pts, axis=2, keepdims=True)
    elif kind == 1:    # a filled box (many points per pixel column)
        pts = rng.uniform(-0.7, 0.7, (B, P, 3))
    else:              # duplicated points: exact depth ties
        half = rng.uniform(-0.6, 0.6, (B, (P + 1) // 2, 3))
        pts = np.concatenate([half, half], 1)[:, :P]
    pts = pts.astype(np.float32)
    colored = bool(rng.integers(0, 2))
    cols = rng.uniform(0, 1, (B, P, 3)).astype(np.float32) if colored else None
    az = rng.uniform(-180, 180, (B, V)).astype(np.float32)
    el = rng.uniform(-80, 80, (B, V)).astype(np.float32)
    di = rng.uniform(0.8, 2.5, (B, V)).astype(np.float32)
    obj, bg = (0.9, 0.8, 0.7), (0.1, 0.2, 0.3)
    want = oracle.render_points(pts, az, el, di, size, radius=radius, obj_color=obj, bg_color=bg, point_colors=cols, points_per_pixel=K)
    R, T = F.look_at_view_transform(t(di.reshape(-1)), t(el.reshape(-1)), t(az.reshape(-1)))
    params = F.points_params(B, P, V, size, radius, obj, bg, strip_rows=strip, points_per_pixel=K)
    img, idx = F.render_points_fragments(t(pts), None if cols is None else t(cols), R, T, t(di.reshape(-1)), params)
    idx, img = idx.cpu().numpy(), img.cpu().numpy()
    nd = int((idx != want["idx"]).sum())
    err = float(np.abs(img - want["images"]).max())
    fail = nd != 0 or err > 2e-6
    bad += fail
    print("case %3d: B=%d V=%d size=%3d P=%4d K=%d strip=%d r=%.3f kind=%d colored=%d covered=%.3f  slot diffs %d  max|d| %.2e%s"
          % (case, B, V, size, P, K, strip, radius, kind, colored, (idx >= 0).mean(), nd, err, "  <-- MISMATCH" if fail else ""))
print("cases with mismatches:", bad)
sys.exit(1 if bad else 0)
