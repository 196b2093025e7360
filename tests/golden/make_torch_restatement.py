"""Generates tests/golden/torch_restatement_c1.npz: BASELINE configs[0] (one tier-M mesh, 4 circular views, 128 x 128)
rendered by a SECOND, independent implementation written with real fp32 torch ops in PyTorch3D 0.7.x's own op order
(recalled) — torch.cos / torch.sin / F.normalize / torch.cross look-at, 4 x 4 bmm transforms with
the homogeneous divide, index_add vertex normals, interpolate_face_attributes as (bary[..., None] * vals).sum(-2),
F.relu / torch.pow Phong, hard blend.  tests/test_oracle.py compares the C oracle with it.

What it pins: the oracle against torch's actual numerics (bmm accumulation, fp32 sin/cos, torch.inverse camera centre,
torch.pow) and against a second reading of the same semantics.  What it cannot pin: PyTorch3D itself (not installable
here) — "parity unpinned" stays.  The two implementations are NOT expected to agree bit for bit: the oracle evaluates
sin / cos correctly rounded and accumulates X R + T without FMA in a fixed order, torch's kernels are free to differ in
the last bit, and a last-bit change of a projected vertex moves a handful of pixels on triangle edges.  The test
states how many.

    python tests/golden/make_torch_restatement.py            (CPU, about a minute)
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvtn_amd import synthetic as syn  # noqa: E402

torch.set_num_threads(8)
KEPS = 1e-8


def camera_position_from_spherical_angles(distance, elevation, azimuth):
    elev = math.pi / 180.0 * elevation
    azim = math.pi / 180.0 * azimuth
    x = distance * torch.cos(elev) * torch.sin(azim)
    y = distance * torch.sin(elev)
    z = distance * torch.cos(elev) * torch.cos(azim)
    return torch.stack([x, y, z], dim=1)


def look_at_view_transform(dist, elev, azim):
    C = camera_position_from_spherical_angles(dist, elev, azim)
    at = torch.zeros(1, 3).expand_as(C)
    up = torch.tensor([[0.0, 1.0, 0.0]]).expand_as(C)
    z_axis = F.normalize(at - C, eps=1e-5)
    x_axis = F.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = F.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        x_axis = torch.where(is_close, F.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5), x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1).transpose(1, 2)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T


def transform_points(points, M):
    """Transform3d.transform_points: [p 1] @ M, then divide by the homogeneous coordinate."""
    ones = torch.ones(points.shape[0], points.shape[1], 1)
    out = torch.bmm(torch.cat([points, ones], dim=2), M)
    return out[..., :3] / out[..., 3:]


def world_to_view_matrix(R, T):
    N = R.shape[0]
    Rm = torch.zeros(N, 4, 4)
    Rm[:, :3, :3] = R
    Rm[:, 3, 3] = 1.0
    Tm = torch.eye(4).repeat(N, 1, 1)
    Tm[:, 3, :3] = T
    return torch.bmm(Rm, Tm)            # Rotate(R).compose(Translate(T))


def fov_projection_matrix(N, fov_deg=60.0, znear=1.0, zfar=100.0, aspect=1.0):
    K = torch.zeros(N, 4, 4)
    fov = torch.tensor((math.pi / 180) * fov_deg)
    tan_half = torch.tan(fov / 2)
    max_y = tan_half * znear
    min_y = -max_y
    max_x = max_y * aspect
    min_x = -max_x
    K[:, 0, 0] = 2.0 * znear / (max_x - min_x)
    K[:, 1, 1] = 2.0 * znear / (max_y - min_y)
    K[:, 0, 2] = (max_x + min_x) / (max_x - min_x)
    K[:, 1, 2] = (max_y + min_y) / (max_y - min_y)
    K[:, 3, 2] = 1.0
    K[:, 2, 2] = zfar / (zfar - znear)
    K[:, 2, 3] = -(zfar * znear) / (zfar - znear)
    return K.transpose(1, 2)


def verts_normals(verts, faces):
    """Meshes._compute_vertex_normals as in the PyTorch3D releases of the reference's era (its README pins pytorch 1.7.1, i.e.
    PyTorch3D 0.3 / 0.4; the form recalled for 0.2 - 0.7): THREE cross products per face, index_add'ed to vertices 1, 2, 0 in
    that order.  (Rounds 2-3 generated the fixtures with the later one-cross form; colours differ by ~2e-6, coverage never.)"""
    vf = verts[faces]
    n = torch.zeros_like(verts)
    n = n.index_add(0, faces[:, 1], torch.cross(vf[:, 2] - vf[:, 1], vf[:, 0] - vf[:, 1], dim=1))
    n = n.index_add(0, faces[:, 2], torch.cross(vf[:, 0] - vf[:, 2], vf[:, 1] - vf[:, 2], dim=1))
    n = n.index_add(0, faces[:, 0], torch.cross(vf[:, 1] - vf[:, 0], vf[:, 2] - vf[:, 0], dim=1))
    return F.normalize(n, eps=1e-6, dim=1)


def edge(px, py, ax, ay, bx, by):
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def rasterize(ndc, faces, S, chunk=256):
    """RasterizeMeshesNaive with blur 0, faces_per_pixel 1, perspective_correct: (pix_to_face, zbuf, bary)."""
    idx = torch.arange(S, dtype=torch.float32)
    c = -1.0 + (2.0 * idx + 1.0) / S                     # PixToNdc(i, S)
    py = c.flip(0)[:, None].expand(S, S).reshape(-1, 1)  # image row yi <-> NDC index S-1-yi
    px = c.flip(0)[None, :].expand(S, S).reshape(-1, 1)
    P = S * S
    best_z = torch.full((P,), float("inf"))
    best_f = torch.full((P,), -1, dtype=torch.int64)
    best_b = torch.full((P, 3), -1.0)
    fv = ndc[faces]                                      # (F, 3, 3)
    for f0 in range(0, faces.shape[0], chunk):
        v = fv[f0:f0 + chunk]
        v0, v1, v2 = v[:, 0][None], v[:, 1][None], v[:, 2][None]          # (1, C, 3)
        xmin = torch.minimum(torch.minimum(v0[..., 0], v1[..., 0]), v2[..., 0]); xmax = torch.maximum(torch.maximum(v0[..., 0], v1[..., 0]), v2[..., 0])
        ymin = torch.minimum(torch.minimum(v0[..., 1], v1[..., 1]), v2[..., 1]); ymax = torch.maximum(torch.maximum(v0[..., 1], v1[..., 1]), v2[..., 1])
        zmin = torch.minimum(torch.minimum(v0[..., 2], v1[..., 2]), v2[..., 2]); zmax = torch.maximum(torch.maximum(v0[..., 2], v1[..., 2]), v2[..., 2])
        face_area = edge(v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1], v2[..., 0], v2[..., 1])
        zero_area = (face_area <= KEPS) & (face_area >= -KEPS)
        outside = (px > xmax) | (px < xmin) | (py > ymax) | (py < ymin) | (zmin < KEPS)
        skip = (zmax < 0) | outside | zero_area
        area = edge(v2[..., 0], v2[..., 1], v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1]) + KEPS
        w0 = edge(px, py, v1[..., 0], v1[..., 1], v2[..., 0], v2[..., 1]) / area
        w1 = edge(px, py, v2[..., 0], v2[..., 1], v0[..., 0], v0[..., 1]) / area
        w2 = edge(px, py, v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1]) / area
        t0 = w0 * v1[..., 2] * v2[..., 2]
        t1 = v0[..., 2] * w1 * v2[..., 2]
        t2 = v0[..., 2] * v1[..., 2] * w2
        den = torch.clamp(t0 + t1 + t2, min=KEPS)
        w0, w1, w2 = t0 / den, t1 / den, t2 / den
        pz = w0 * v0[..., 2] + w1 * v1[..., 2] + w2 * v2[..., 2]
        hit = ~skip & (pz >= 0) & (w0 > 0) & (w1 > 0) & (w2 > 0)
        pz = torch.where(hit, pz, torch.tensor(float("inf")))
        zc, fc = pz.min(dim=1)                           # first minimum = lowest face index among ties
        upd = zc < best_z
        sel = fc[upd]
        rows = torch.nonzero(upd)[:, 0]
        best_z[upd] = zc[upd]
        best_f[upd] = sel + f0
        best_b[upd] = torch.stack([w0[rows, sel], w1[rows, sel], w2[rows, sel]], 1)
    return best_f.view(S, S), best_z.view(S, S), best_b.view(S, S, 3)


def phong(p2f, bary, verts, faces, normals, cam_center, light_dir, obj, bg):
    S = p2f.shape[0]
    mask = p2f >= 0
    f = p2f.clamp(min=0)
    fverts, fnorm = verts[faces][f], normals[faces][f]                  # (S, S, 3, 3)
    fcol = torch.tensor(obj)[None, None, None, :].expand(S, S, 3, 3)
    pixel_coords = (bary[..., None] * fverts).sum(dim=-2)
    pixel_normals = (bary[..., None] * fnorm).sum(dim=-2)
    texels = (bary[..., None] * fcol).sum(dim=-2)
    nrm = F.normalize(pixel_normals, p=2, dim=-1, eps=1e-6)
    direction = F.normalize(torch.tensor(light_dir)[None, None, :].expand(S, S, 3), p=2, dim=-1, eps=1e-6)
    cos_angle = torch.sum(nrm * direction, dim=-1)
    diffuse = torch.tensor([0.3, 0.3, 0.3]) * F.relu(cos_angle)[..., None]
    lit = (cos_angle > 0).to(torch.float32)
    view_direction = F.normalize(cam_center[None, None, :] - pixel_coords, p=2, dim=-1, eps=1e-6)
    reflect = -direction + 2 * (cos_angle[..., None] * nrm)
    alpha = F.relu(torch.sum(view_direction * reflect, dim=-1)) * lit
    specular = torch.tensor([0.2, 0.2, 0.2]) * torch.pow(alpha, 64.0)[..., None]
    ambient = torch.tensor([1.0, 1.0, 1.0]) * torch.tensor([0.5, 0.5, 0.5])
    colors = (ambient + diffuse) * texels + specular
    out = torch.where(mask[..., None], colors, torch.tensor(bg)[None, None, :])
    return out.permute(2, 0, 1)


def main():
    S, V = 128, 4
    v_np, f_np = syn.synthetic_mesh("M", 0)
    az, el, di = syn.circular_views(V, 30.0, 2.2)
    verts, faces = torch.from_numpy(v_np), torch.from_numpy(f_np)
    R, T = look_at_view_transform(torch.from_numpy(di), torch.from_numpy(el), torch.from_numpy(az))
    W2V = world_to_view_matrix(R, T)
    K = fov_projection_matrix(V)
    pts = verts[None].expand(V, -1, -1)
    view = transform_points(pts, W2V)
    ndc = transform_points(view, K)
    ndc[..., 2] = view[..., 2]
    nrm = verts_normals(verts, faces)
    centers = torch.inverse(W2V)[:, 3, :3]
    obj = bg = (0.99999, 0.99999, 0.99999)
    p2f_all, img_all, z_all = [], [], []
    for n in range(V):
        p2f, zb, bary = rasterize(ndc[n], faces, S)
        img_all.append(phong(p2f, bary, verts, faces, nrm, centers[n], (0.0, 1.0, 0.0), obj, bg))
        p2f_all.append(p2f)
        z_all.append(torch.where(p2f >= 0, zb, torch.tensor(-1.0)))
        print("view", n, "covered", float((p2f >= 0).float().mean()))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "torch_restatement_c1.npz")
    np.savez_compressed(path, pix_to_face=torch.stack(p2f_all).numpy().astype(np.int16), images=torch.stack(img_all).numpy().astype(np.float32),
                        zbuf=torch.stack(z_all).numpy().astype(np.float32), R=R.numpy(), T=T.numpy(),
                        azim=az, elev=el, dist=di, image_size=np.array(S), mesh=np.array(["M", "0"]))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
