"""Generates tests/golden/torch_restatement_grad32.npz: the torch.autograd pass of make_torch_restatement_grad.py /
make_torch_restatement_r5.py run in FLOAT32 — the arithmetic the reference itself differentiates in (PyTorch3D's kernels and
torch.autograd on fp32 tensors, run_mvtn.py:195).  The float64 fixtures are the derivative of the restated function; the library's
fp32 kernels differ from them by up to 3e-3 of the largest derivative on views with pixels on nearly edge-on faces
(tests/test_gpu_mesh_backward.py).  This file holds what a float32 autograd pass through the same torch ops gives for the same
losses, so that the tests can compare the library's error with the reference arithmetic's own error instead of explaining it:
|library - f64| <= 2 |torch f32 - f64| + 2e-4 of the largest derivative, per derivative.

Same cases, same seeds, same pixel -> face maps (the fp32 rasteriser pass, held fixed):
  c3   two tier-M objects (seeds 0, 1), 12 jittered spherical views each, 224 x 224, relative light, G seeds 200, 201;
  cad  cad_mesh seed 7 (1 280 faces), 4 jittered circular views, 128 x 128, relative light, G seed 300.

    python tests/golden/make_torch_restatement_grad32.py        (CPU, ~10 min)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from mvtn_amd import synthetic as syn  # noqa: E402
import make_torch_restatement as r1  # noqa: E402
import make_torch_restatement_grad as rg  # noqa: E402


def grads_in(dtype, verts, faces, az, el, di, S, relative_light, g_seed):
    """d sum(G images) / d (azim, elev, dist) by torch.autograd with every tensor of the differentiable pass in `dtype`; the
    pixel -> face maps from the fp32 rasteriser pass, held fixed (as make_torch_restatement_grad.render_views_diff)."""
    V = len(az)
    with torch.no_grad():
        ndc32, _ = rg._project(verts, faces, *(torch.from_numpy(np.asarray(x, np.float32)) for x in (az, el, di)), V)
        p2f_all = [r1.rasterize(ndc32[n], faces, S)[0] for n in range(V)]
    torch.set_default_dtype(dtype)
    try:
        vd = verts.to(dtype)
        a, e, d = (torch.from_numpy(np.asarray(x, np.float32)).to(dtype).requires_grad_(True) for x in (az, el, di))
        ndc, W2V = rg._project(vd, faces, a, e, d, V)
        nrm = r1.verts_normals(vd, faces)
        centers = torch.inverse(W2V)[:, 3, :3]
        lights = r1.camera_position_from_spherical_angles(d, e, a) if relative_light else None
        obj = bg = (float(np.float32(0.99999)),) * 3
        G = torch.from_numpy(np.random.default_rng(g_seed).standard_normal((V, 3, S, S)).astype(np.float32)).to(dtype)
        loss = torch.zeros(())
        for n in range(V):
            bary = rg.bary_of_winners(ndc[n], faces, p2f_all[n])
            light = lights[n] if relative_light else torch.tensor([0.0, 1.0, 0.0])
            img, _ = rg.phong_diff(p2f_all[n], bary, vd, faces, nrm, centers[n], light, obj, bg)
            loss = loss + (img * G[n]).sum()
            print("  view", n, flush=True)
        loss.backward()
        out = [x.grad.detach().to(torch.float64).numpy().copy() for x in (a, e, d)]
    finally:
        torch.set_default_dtype(torch.float32)
    return out


def main():
    torch.set_num_threads(8)
    out = {}
    # ---- c3 (make_torch_restatement_grad.main)
    B, V, S = 2, 12, 224
    az, el, di = rg.c3_views(B, V, 1)
    g32 = []
    for b in range(B):
        v, f = syn.synthetic_mesh("M", b)
        g32.append(grads_in(torch.float32, torch.from_numpy(v), torch.from_numpy(f), az[b], el[b], di[b], S, True, 200 + b))
    out.update(c3_grad_azim_f32=np.stack([g[0] for g in g32]), c3_grad_elev_f32=np.stack([g[1] for g in g32]), c3_grad_dist_f32=np.stack([g[2] for g in g32]))
    # ---- cad (make_torch_restatement_r5.main)
    v, f = syn.cad_mesh(7, n_faces=1280)
    azc, elc, dic = syn.circular_views(4, 30.0, 2.2)
    rng = np.random.default_rng(3)
    azj = (azc + rng.uniform(-5, 5, 4)).astype(np.float32)
    elj = (elc + rng.uniform(-5, 5, 4)).astype(np.float32)
    dij = (dic + rng.uniform(-0.1, 0.1, 4)).astype(np.float32)
    ga, ge, gd = grads_in(torch.float32, torch.from_numpy(v), torch.from_numpy(f), azj, elj, dij, 128, True, 300)
    out.update(cad_grad_azim_f32=ga, cad_grad_elev_f32=ge, cad_grad_dist_f32=gd)
    # self-check of this script against the committed float64 fixtures: the same function in float64 must reproduce them
    g64 = grads_in(torch.float64, torch.from_numpy(v), torch.from_numpy(f), azj, elj, dij, 128, True, 300)
    ref = np.load(os.path.join(HERE, "torch_restatement_r5.npz"))
    for got, key in zip(g64, ("cad_grad_azim", "cad_grad_elev", "cad_grad_dist")):
        assert np.allclose(got, ref[key], rtol=1e-9, atol=1e-9), key
    path = os.path.join(HERE, "torch_restatement_grad32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
