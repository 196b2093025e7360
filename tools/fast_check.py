"""developer scratch (GPU): the fast rasteriser against the exact kernel — bit-equal pixel->face maps and images,
how many strips were handed back, and the time of each mode.   python3 tools/fast_check.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtn_amd import functional as F, synthetic as syn
from mvtn_amd.structures import Meshes, PackedMeshes

dev = torch.device("cuda:0")
t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32).to(dev)


def run(name, specs, V, size, jitter, dist=2.2, reps=50):
    vl, fl = [], []
    for tier, seed in specs:
        v, f = syn.cad_soup(seed) if tier == "soup" else syn.synthetic_mesh(tier, seed)
        vl.append(v); fl.append(f)
    B = len(vl)
    az, el, di = syn.spherical_views(V)
    az, el, di = np.tile(az, (B, 1)), np.tile(el, (B, 1)), np.full((B, V), dist, np.float32)
    if jitter:
        rng = np.random.default_rng(1)
        az = az + rng.uniform(-5, 5, az.shape); el = el + rng.uniform(-5, 5, el.shape); di = di + rng.uniform(-0.1, 0.1, di.shape)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)
    R, T = F.look_at_view_transform(t(di.reshape(-1)), t(el.reshape(-1)), t(az.reshape(-1)))
    light = t(np.array([[0.0, 1.0, 0.0]], np.float32))
    obj = bg = (0.99999,) * 3
    out = {}
    for mode in ("fast", "exact"):
        params = F.mesh_params(V, size, 1, obj, bg, exact_raster=(mode == "exact"))
        img, p2f, cnt = F.render_meshes_fragments(packed, R, T, light, params, return_redo=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            F.render_meshes_fragments(packed, R, T, light, params)
        torch.cuda.synchronize()
        out[mode] = (img.cpu().numpy(), p2f.cpu().numpy(), int(cnt.item()), (time.perf_counter() - t0) / reps * 1e6)
    n_strips = B * V * ((size + 7) // 8)
    nd = int((out["fast"][1] != out["exact"][1]).sum())
    dimg = float(np.abs(out["fast"][0] - out["exact"][0]).max())
    print("%-28s p2f mismatches %d  max|dimg| %.3g  redo %d / %d strips (%.2f%%)  fast %.1f us  exact %.1f us" % (
        name, nd, dimg, out["fast"][2], n_strips, 100.0 * out["fast"][2] / n_strips, out["fast"][3], out["exact"][3]), flush=True)
    return nd


bad = 0
bad += run("C2 16xM 12v 224", [("M", s) for s in range(16)], 12, 224, False, reps=200)
bad += run("C2 jittered", [("M", s) for s in range(16)], 12, 224, True)
bad += run("C5-slice 8xL 20v", [("L", s) for s in range(8)], 20, 224, False)
bad += run("S tier 4 views 128", [("S", 0)], 4, 128, False)
bad += run("soup + M ragged", [("S", 3), ("L", 4), ("soup", 5), ("M", 6)], 6, 224, True)
bad += run("close-up dist 1.2", [("M", 0), ("M", 1)], 12, 224, True, dist=1.2)
bad += run("inside dist 0.8", [("M", 0), ("M", 1)], 12, 224, True, dist=0.8)
bad += run("odd size 37", [("S", 10), ("soup", 11)], 2, 37, True)
print("TOTAL mismatching pixels:", bad)
sys.exit(1 if bad else 0)
