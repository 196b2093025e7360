"""Manual stress run (not collected by pytest): random meshes / views / image sizes / K / blur radii / z-clip through the soft
path (mvr_render_meshes_soft_forward) against the oracle — fragment lists bit-exact.  (The images are reported, not judged: the
shading is ill-conditioned where a vertex normal nearly vanishes — the CAD-soup meshes' coincident double-sided faces: ambient-only on
one side, ambient + diffuse on the other — and the GPU tests compare colours on the well-conditioned pixels only; here the share of
pixels that differ by more than 1e-4 is printed.)
    python tests/stress_soft.py [n_cases] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from mvtn_amd import functional as F, synthetic as syn  # noqa: E402
from mvtn_amd.structures import Meshes, PackedMeshes  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
oracle.set_threads(8)
t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)   # noqa: E731
bad = 0
for case in range(n_cases):
    B, V = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    size = int(rng.choice([32, 45, 64, 100, 128]))
    K = int(rng.choice([1, 2, 4, 4, 4, 6]))
    sigma = float(rng.choice([1e-4, 1e-4, 3e-4, 1e-5]))
    blur = float(np.log(1.0 / 1e-4 - 1.0) * sigma)
    shader = str(rng.choice(["soft_phong", "soft_phong", "soft_silhouette"]))
    persp, clipb = bool(rng.integers(0, 2)) or True, bool(rng.integers(0, 2))
    zc = float(rng.choice([0.0, 0.5])) if persp else 0.0
    tiers = [str(rng.choice(["S", "M", "soup"])) for _ in range(B)]
    vl, fl = [], []
    for i, tr in enumerate(tiers):
        v, f = (syn.cad_soup(int(rng.integers(0, 100))) if tr == "soup" and hasattr(syn, "cad_soup") else syn.synthetic_mesh("S" if tr == "soup" else tr, int(rng.integers(0, 1000))))
        vl.append(v); fl.append(f)
    az = rng.uniform(-180, 180, (B, V)).astype(np.float32)
    el = rng.uniform(-80, 80, (B, V)).astype(np.float32)
    di = rng.uniform(0.9 if zc > 0 else 1.6, 2.6, (B, V)).astype(np.float32)
    want = oracle.render_meshes_soft(vl, fl, az, el, di, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, clip_barycentric_coords=clipb,
                                     obj_color=(0.9, 0.8, 0.7), perspective_correct=persp, z_clip=zc)
    packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
    R, T = t(want["R"]), t(want["T"])
    light = t(np.array([[0.0, 1.0, 0.0]], np.float32))
    params = F.mesh_params(V, size, 1, (0.9, 0.8, 0.7), (1.0, 1.0, 1.0), perspective_correct=persp, z_clip=zc)
    soft = F.soft_params(blur, K, shader, sigma, 1e-4, clipb)
    with torch.no_grad():
        img, p2f = F.render_meshes_soft(packed, R, T, light, params, soft, return_fragments=True)
    torch.cuda.synchronize()
    nd = int((p2f.cpu().numpy() != want["pix_to_face"]).sum())
    d = np.abs(img.cpu().numpy() - want["images"])
    print("case %d: B=%d V=%d size=%3d K=%d sigma=%g %s clipb=%d zclip=%.1f  slot diffs %d  pixels off by > 1e-4: %.4f %%" % (
        case, B, V, size, K, sigma, shader, clipb, zc, nd, 100.0 * float((d.max(axis=2) > 1e-4).mean())))
    if nd:
        bad += 1
print("cases with mismatches: %d" % bad)
