"""Generates tests/golden/torch_restatement_grad.npz: a SECOND implementation of the mesh path's BACKWARD (missing 1a).

Every gradient test of rounds 1-3 differentiated the builder's own C oracle by finite differences.  Here the forward of
make_torch_restatement.py (real fp32 torch ops in PyTorch3D's op order: look-at, 4 x 4 bmm transforms + homogeneous divide,
three-cross vertex normals, phong_shading, hard blend) is re-run DIFFERENTIABLY: the pixel -> face map comes from the naive
rasteriser under no_grad and is then held fixed, exactly as PyTorch3D's rasteriser treats it (no gradient through indices,
run_mvtn.py:195 only ever sees d colour / d barycentrics); the perspective-corrected barycentrics of each pixel's winning
face are recomputed from the projected vertices with the forward's operations as differentiable torch ops, the Phong
shading and the camera centre (torch.inverse of the world-to-view matrix) likewise, and `loss.backward()` through
torch.autograd gives d loss / d (azim, elev, dist).  loss = sum(G * images) with G = standard normal, seeded.

Two cases:
  c1   BASELINE configs[0]: one tier-M mesh, 4 circular views, 128 x 128, light_direction "fixed" (0, 1, 0);
  c3   two objects of BASELINE configs[2]: tier M (seeds 0, 1), 12 jittered spherical views each, 224 x 224,
       light_direction "relative" (the light is the camera position, models/renderer.py:167-171: a third gradient path).

What it pins: the analytic backward kernels (k_mesh_backward_tab, k_mesh_backward) and the oracle's finite differences
against torch.autograd through a second reading of the same semantics.  What it cannot pin: PyTorch3D itself (not
installable here) — "parity unpinned" stays.

    python tests/golden/make_torch_restatement_grad.py [c1] [c3]        (CPU; c1 ~1 min, c3 ~15 min)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from mvtn_amd import synthetic as syn  # noqa: E402
import make_torch_restatement as r1  # noqa: E402

KEPS = r1.KEPS


def pix_centres(S):
    idx = torch.arange(S, dtype=torch.get_default_dtype())
    c = (-1.0 + (2.0 * idx.float() + 1.0) / S).to(idx.dtype)   # the fp32 pixel centres, as the rasteriser sees them
    py = c.flip(0)[:, None].expand(S, S)
    px = c.flip(0)[None, :].expand(S, S)
    return px, py


def bary_of_winners(ndc, faces, p2f):
    """BarycentricCoordsForward + BarycentricPerspectiveCorrectionForward of the winning face of every pixel, as
    differentiable torch ops on the projected vertices (S, S, 3); background pixels get the face 0 (masked later)."""
    S = p2f.shape[0]
    px, py = pix_centres(S)
    fv = ndc[faces][p2f.clamp(min=0)]                                 # (S, S, 3 corners, 3)
    v0, v1, v2 = fv[..., 0, :], fv[..., 1, :], fv[..., 2, :]
    area = r1.edge(v2[..., 0], v2[..., 1], v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1]) + KEPS
    w0 = r1.edge(px, py, v1[..., 0], v1[..., 1], v2[..., 0], v2[..., 1]) / area
    w1 = r1.edge(px, py, v2[..., 0], v2[..., 1], v0[..., 0], v0[..., 1]) / area
    w2 = r1.edge(px, py, v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1]) / area
    t0 = w0 * v1[..., 2] * v2[..., 2]
    t1 = v0[..., 2] * w1 * v2[..., 2]
    t2 = v0[..., 2] * v1[..., 2] * w2
    den = torch.clamp(t0 + t1 + t2, min=KEPS)
    return torch.stack([t0 / den, t1 / den, t2 / den], dim=-1)


def phong_diff(p2f, bary, verts, faces, normals, cam_center, light_dir, obj, bg):
    """r1.phong with a light direction that may be a tensor carrying a gradient (the "relative" light)."""
    S = p2f.shape[0]
    mask = p2f >= 0
    f = p2f.clamp(min=0)
    fverts, fnorm = verts[faces][f], normals[faces][f]
    fcol = torch.tensor(obj)[None, None, None, :].expand(S, S, 3, 3)
    pixel_coords = (bary[..., None] * fverts).sum(dim=-2)
    pixel_normals = (bary[..., None] * fnorm).sum(dim=-2)
    texels = (bary[..., None] * fcol).sum(dim=-2)
    nrm = F.normalize(pixel_normals, p=2, dim=-1, eps=1e-6)
    direction = F.normalize(light_dir[None, None, :].expand(S, S, 3), p=2, dim=-1, eps=1e-6)
    cos_angle = torch.sum(nrm * direction, dim=-1)
    diffuse = torch.tensor([0.3, 0.3, 0.3]) * F.relu(cos_angle)[..., None]
    lit = (cos_angle > 0).to(cos_angle.dtype)
    view_direction = F.normalize(cam_center[None, None, :] - pixel_coords, p=2, dim=-1, eps=1e-6)
    reflect = -direction + 2 * (cos_angle[..., None] * nrm)
    alpha = F.relu(torch.sum(view_direction * reflect, dim=-1)) * lit
    specular = torch.tensor([0.2, 0.2, 0.2]) * torch.pow(alpha, 64.0)[..., None]
    ambient = torch.tensor([1.0, 1.0, 1.0]) * torch.tensor([0.5, 0.5, 0.5])
    colors = (ambient + diffuse) * texels + specular
    out = torch.where(mask[..., None], colors, torch.tensor(bg)[None, None, :])
    # (second value: the pixels on a KINK of the diffuse term — covered, |n . l| < KINK —, where an fp32 evaluation may take the
    # other branch of relu(n . l) than this float64 one)
    return out.permute(2, 0, 1), (mask & (cos_angle.detach().abs() < KINK))


KINK = 1e-5
# Round 5: the pixels of nearly EDGE-ON winning faces.  cond = (larger extent of the face's NDC box)^2 / |2 x NDC area| is ~1-2 for a
# well-shaped face and grows like 1 / foreshortening; a pixel's term of d loss / d angle contains d barycentrics / d vertex ~ 1 / area,
# so a handful of pixels on such faces carry terms 100-1000 x the typical one, and an fp32 evaluation of THEIR edge functions is what
# the fp32 kernels differ by from this float64 pass (not relu kinks: no covered pixel of the c3 case has |n . l| < 1e-5).  The
# fixture therefore also holds, per threshold, the bit mask of the pixels with cond > threshold and the gradients of the loss
# without them.
COND_THRESHOLDS = (30.0, 100.0, 300.0, 1000.0)


def face_condition(ndc32, faces, p2f):
    fv = ndc32[faces][p2f.clamp(min=0)]                                 # (S, S, 3 corners, 3)
    x, y = fv[..., 0], fv[..., 1]
    ext = torch.maximum(x.max(-1).values - x.min(-1).values, y.max(-1).values - y.min(-1).values)
    area2 = ((x[..., 1] - x[..., 0]) * (y[..., 2] - y[..., 0]) - (x[..., 2] - x[..., 0]) * (y[..., 1] - y[..., 0])).abs()
    cond = ext * ext / area2.clamp(min=1e-30)
    return torch.where(p2f >= 0, cond, torch.zeros_like(cond))


def _project(verts, faces, a, e, d, V):
    R, T = r1.look_at_view_transform(d, e, a)
    W2V = r1.world_to_view_matrix(R, T)
    K = r1.fov_projection_matrix(V)
    view = r1.transform_points(verts[None].expand(V, -1, -1), W2V)
    ndc_xy = r1.transform_points(view, K)
    return torch.cat([ndc_xy[..., :2], view[..., 2:3]], dim=-1), W2V      # the rasteriser keeps the view depth


def render_views_diff(verts, faces, az, el, di, S, relative_light, g_seed):
    """One mesh, V views.  Pass 1 (fp32, no grad): the pixel -> face maps of the fp32 forward.  Pass 2 (the same torch ops in
    float64, maps held fixed): loss = sum(G * images), gradients by torch.autograd — in double so that the fixture is the
    derivative of the restated function itself, not that plus the rounding noise of an fp32 backward (with the random G the
    sums cancel heavily: an fp32 autograd pass differs from fp64 central differences by ~1e-3 of the largest derivative)."""
    V = len(az)
    with torch.no_grad():
        ndc32, _ = _project(verts, faces, *(torch.from_numpy(np.asarray(x, np.float32)) for x in (az, el, di)), V)
        p2f_all = [r1.rasterize(ndc32[n], faces, S)[0] for n in range(V)]
        cond_all = [face_condition(ndc32[n], faces, p2f_all[n]).double() for n in range(V)]
    torch.set_default_dtype(torch.float64)
    try:
        vd = verts.double()
        a, e, d = (torch.from_numpy(np.asarray(x, np.float32)).double().requires_grad_(True) for x in (az, el, di))
        ndc, W2V = _project(vd, faces, a, e, d, V)
        nrm = r1.verts_normals(vd, faces)
        centers = torch.inverse(W2V)[:, 3, :3]
        lights = r1.camera_position_from_spherical_angles(d, e, a) if relative_light else None
        obj = bg = (float(np.float32(0.99999)),) * 3
        G = torch.from_numpy(np.random.default_rng(g_seed).standard_normal((V, 3, S, S)).astype(np.float32)).double()
        loss = torch.zeros(())
        loss_m = torch.zeros(())   # the same loss without the pixels on a kink of relu(n . l) (round 5)
        loss_c = [torch.zeros(()) for _ in COND_THRESHOLDS]   # ... and without the pixels of badly conditioned (edge-on) faces
        img_all, kink_all = [], []
        for n in range(V):
            bary = bary_of_winners(ndc[n], faces, p2f_all[n])
            light = lights[n] if relative_light else torch.tensor([0.0, 1.0, 0.0])
            img, kink = phong_diff(p2f_all[n], bary, vd, faces, nrm, centers[n], light, obj, bg)
            loss = loss + (img * G[n]).sum()
            loss_m = loss_m + (img * G[n] * (~kink)[None].to(img.dtype)).sum()
            for k, thr in enumerate(COND_THRESHOLDS):
                loss_c[k] = loss_c[k] + (img * G[n] * (cond_all[n] <= thr)[None].to(img.dtype)).sum()
            img_all.append(img.detach())
            kink_all.append(kink)
            print("  view", n, "covered", float((p2f_all[n] >= 0).double().mean()), "pixels on a kink", int(kink.sum()), flush=True)
        gm = torch.autograd.grad(loss_m, (a, e, d), retain_graph=True)
        gc = [[g.numpy().copy() for g in torch.autograd.grad(lc, (a, e, d), retain_graph=True)] for lc in loss_c]
        cond_bits = [np.packbits((torch.stack(cond_all) > thr).numpy(), axis=-1) for thr in COND_THRESHOLDS]
        loss.backward()
        out = (torch.stack(p2f_all).numpy().astype(np.int16), torch.stack(img_all).numpy(),
               a.grad.numpy().copy(), e.grad.numpy().copy(), d.grad.numpy().copy(),
               np.packbits(torch.stack(kink_all).numpy(), axis=-1), [g.numpy().copy() for g in gm], cond_bits, gc)
    finally:
        torch.set_default_dtype(torch.float32)
    return out


def c3_views(B, V, seed):
    """The jittered views of bench.py / the GPU tests for configs[2]: grid + U(-5, 5) degrees, U(-0.1, 0.1) distance."""
    az, el, di = syn.spherical_views(V)
    rng = np.random.default_rng(seed)
    az = np.tile(az, (B, 1)) + rng.uniform(-5, 5, (B, V))
    el = np.tile(el, (B, 1)) + rng.uniform(-5, 5, (B, V))
    di = np.tile(di, (B, 1)) + rng.uniform(-0.1, 0.1, (B, V))
    return az.astype(np.float32), el.astype(np.float32), di.astype(np.float32)


def main(which):
    torch.set_num_threads(8)
    out = {}
    if "c1" in which:
        v, f = syn.synthetic_mesh("M", 0)
        az, el, di = syn.circular_views(4, 30.0, 2.2)
        p2f, img, ga, ge, gd = render_views_diff(torch.from_numpy(v), torch.from_numpy(f), az, el, di, 128, False, 101)[:5]
        out.update(c1_azim=az, c1_elev=el, c1_dist=di, c1_pix_to_face=p2f, c1_grad_azim=ga, c1_grad_elev=ge, c1_grad_dist=gd,
                   c1_g_seed=np.array(101), c1_image_size=np.array(128), c1_image_sum=np.array(img.astype(np.float64).sum()))
        print("c1 grads", ga, ge, gd)
    if "c3" in which:
        B, V, S = 2, 12, 224
        az, el, di = c3_views(B, V, 1)
        p2fs, gas, ges, gds, sums, kinks, gms, cbits, gcs = [], [], [], [], [], [], [], [], []
        for b in range(B):
            v, f = syn.synthetic_mesh("M", b)
            p2f, img, ga, ge, gd, kink, gm, cb, gc = render_views_diff(torch.from_numpy(v), torch.from_numpy(f), az[b], el[b], di[b], S, True, 200 + b)
            cbits.append(cb); gcs.append(gc)
            p2fs.append(p2f); gas.append(ga); ges.append(ge); gds.append(gd); sums.append(img.astype(np.float64).sum())
            kinks.append(kink); gms.append(gm)
        out.update(c3_azim=az, c3_elev=el, c3_dist=di, c3_pix_to_face=np.stack(p2fs), c3_grad_azim=np.stack(gas),
                   c3_grad_elev=np.stack(ges), c3_grad_dist=np.stack(gds), c3_g_seeds=np.array([200, 201]), c3_image_size=np.array(S),
                   c3_mesh_seeds=np.array([0, 1]), c3_image_sum=np.array(sums),
                   # the pixels with |n . l| < KINK (bit-packed along the columns) and the gradients of the loss WITHOUT them
                   c3_kink_bits=np.stack(kinks), c3_kink_threshold=np.array(KINK),
                   c3_grad_azim_nokink=np.stack([g[0] for g in gms]), c3_grad_elev_nokink=np.stack([g[1] for g in gms]),
                   c3_grad_dist_nokink=np.stack([g[2] for g in gms]),
                   # per threshold t: the pixels whose winning face has cond > t (face_condition; bit-packed), the gradients without them
                   c3_cond_thresholds=np.array(COND_THRESHOLDS),
                   c3_cond_bits=np.stack([np.stack([cbits[b][k] for b in range(B)]) for k in range(len(COND_THRESHOLDS))]),
                   c3_grad_azim_cond=np.stack([np.stack([gcs[b][k][0] for b in range(B)]) for k in range(len(COND_THRESHOLDS))]),
                   c3_grad_elev_cond=np.stack([np.stack([gcs[b][k][1] for b in range(B)]) for k in range(len(COND_THRESHOLDS))]),
                   c3_grad_dist_cond=np.stack([np.stack([gcs[b][k][2] for b in range(B)]) for k in range(len(COND_THRESHOLDS))]))
        print("c3 grads", np.stack(gas), np.stack(ges), np.stack(gds))
    path = os.path.join(HERE, "torch_restatement_grad.npz")
    if os.path.exists(path):   # a partial run keeps the other case
        old = dict(np.load(path))
        old.update(out)
        out = old
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main(sys.argv[1:] or ["c1", "c3"])
