"""Generates tests/golden/torch_restatement_{points,zclip,soft}.npz: three more paths rendered by the SECOND implementation
(real fp32 torch ops in PyTorch3D 0.7.x's own op order, SURVEY Appendix A, RECALLED) that make_torch_restatement.py started:

  points  models/renderer.py:116-151 — 2 clouds x 20 spherical views of BASELINE configs[3] (2048 points, radius 0.006,
          224 x 224, dist 1): Pointclouds.scale_(1 / dist) -> world-to-view 4 x 4 bmm -> FoVOrthographic projection matrix ->
          RasterizePointsNaive (dx^2 + dy^2 < r^2, z >= 0, K nearest) -> weights 1 - d2 / (r r) -> NormWeightedCompositor,
          for points_per_pixel = 1 (the reference's setting) and 4;
  zclip   models/renderer.py:84-107 with the camera at dist 1.1 (models/mvtn.py:33,68,109 `transform_distance` lets it get
          there): clip_faces at z = znear / 2 (cull / 1 vertex behind -> 2 triangles / 2 behind -> 1, new vertices
          interpolated in (x z, y z, z) and projected back by their own z) -> RasterizeMeshesNaive on the clipped faces ->
          convert_clipped_rasterization_to_original_faces (barycentric conversion matrices, bmm) -> HardPhongShader;
  soft    the shaders the reference imports and leaves off (models/renderer.py:3-12): blur_radius > 0, faces_per_pixel = 4,
          clip_barycentric_coords, SoftPhongShader = phong_shading of every fragment + softmax_rgb_blend (sigma, gamma 1e-4).

tests/test_oracle.py compares the C oracle with them and states how many pixels differ; tests/test_gpu_round3.py compares
the HIP path.  Like the first fixture this pins the oracle against torch's numerics and a second reading of the semantics —
not against PyTorch3D itself (not installable here): "parity unpinned" stays.

    python tests/golden/make_torch_restatement_r3.py            (CPU, a few minutes)
"""
import math
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
import make_torch_restatement as r1  # noqa: E402  (look-at, transforms, vertex normals, Phong of the first fixture)

torch.set_num_threads(8)
KEPS = 1e-8


def pix_centres(S):
    idx = torch.arange(S, dtype=torch.float32)
    c = -1.0 + (2.0 * idx + 1.0) / S                     # PixToNdc(i, S); image row / column k <-> NDC index S - 1 - k
    return c.flip(0)


# ---- points ---------------------------------------------------------------------------------------------------------

def ortho_projection_matrix(N, znear=0.01, zfar=100.0, max_y=1.0, min_y=-1.0, max_x=1.0, min_x=-1.0):
    """FoVOrthographicCameras.compute_projection_matrix (scale_xyz = 1), as the transposed matrix Transform3d multiplies by."""
    K = torch.zeros(N, 4, 4)
    K[:, 0, 0] = (2.0 / (max_x - min_x)) * 1.0
    K[:, 1, 1] = (2.0 / (max_y - min_y)) * 1.0
    K[:, 0, 3] = -(max_x + min_x) / (max_x - min_x)
    K[:, 1, 3] = -(max_y + min_y) / (max_y - min_y)
    K[:, 3, 3] = 1.0
    K[:, 2, 2] = 1.0 * (1.0 / (zfar - znear)) * 1.0
    K[:, 2, 3] = -znear / (zfar - znear)
    return K.transpose(1, 2)


def rasterize_points(ndc, S, radius, K, rows_per_chunk=16):
    """RasterizePointsNaive: per pixel the K points with the smallest z among those with z >= 0 and dx^2 + dy^2 < r^2
    (r^2 = radius * radius in float, as the kernel squares its float argument).  Returns idx, zbuf, dist2 (S, S, K)."""
    c = pix_centres(S)
    r2 = torch.tensor(radius, dtype=torch.float32) * torch.tensor(radius, dtype=torch.float32)
    px_all = c[None, :].expand(S, S)
    py_all = c[:, None].expand(S, S)
    x, y, z = ndc[:, 0][None], ndc[:, 1][None], ndc[:, 2][None]          # (1, P)
    idx = torch.full((S, S, K), -1, dtype=torch.int64)
    zb = torch.full((S, S, K), -1.0)
    d2 = torch.full((S, S, K), -1.0)
    inf = torch.tensor(float("inf"))
    for r0 in range(0, S, rows_per_chunk):
        px = px_all[r0:r0 + rows_per_chunk].reshape(-1, 1)
        py = py_all[r0:r0 + rows_per_chunk].reshape(-1, 1)
        dx = px - x
        dy = py - y
        dist2 = dx * dx + dy * dy
        hit = (dist2 < r2) & ~(z < 0)
        zm = torch.where(hit, z.expand_as(dist2), inf)
        zs, order = torch.topk(zm, K, dim=1, largest=False, sorted=True)   # random clouds: no depth ties inside a pixel
        ok = torch.isfinite(zs)
        rows = min(rows_per_chunk, S - r0)
        idx[r0:r0 + rows] = torch.where(ok, order, torch.tensor(-1)).view(rows, S, K)
        zb[r0:r0 + rows] = torch.where(ok, zs, torch.tensor(-1.0)).view(rows, S, K)
        d2[r0:r0 + rows] = torch.where(ok, torch.gather(dist2, 1, order), torch.tensor(-1.0)).view(rows, S, K)
    return idx, zb, d2


def norm_weighted_composite(idx, d2, feats, radius, bg):
    """PointsRenderer.forward + NormWeightedCompositor: weights = 1 - dists2 / (r * r) (a tensor divided by the python float
    r * r), out = sum_k alpha_k f / max(sum_k alpha_k, 1e-4) accumulated slot by slot, background where slot 0 is empty."""
    weights = 1 - d2 / (radius * radius)
    valid = idx >= 0
    cum = torch.where(valid, weights, torch.tensor(0.0)).sum(dim=-1) if idx.shape[-1] == 1 else None
    if cum is None:                                   # sequential accumulation over the slots, like the kernel's loop
        cum = torch.zeros(idx.shape[:2])
        for k in range(idx.shape[-1]):
            cum = cum + torch.where(valid[..., k], weights[..., k], torch.tensor(0.0))
    cum = torch.where(cum < 1e-4, torch.tensor(1e-4), cum)
    out = torch.zeros(idx.shape[0], idx.shape[1], 3)
    for k in range(idx.shape[-1]):
        f = feats[idx[..., k].clamp(min=0)]                                          # (S, S, 3)
        out = out + torch.where(valid[..., k, None], weights[..., k, None] * f / cum[..., None], torch.tensor(0.0))
    out = torch.where((idx[..., 0] < 0)[..., None], torch.tensor(bg)[None, None, :], out)
    return out.permute(2, 0, 1)


def make_points():
    S, V, P, radius = 224, 20, 2048, 0.006
    az, el, _ = syn.spherical_views(V)
    di = np.full(V, 1.0, np.float32)
    clouds = []
    for seed in (0, 1):
        v, f = syn.synthetic_mesh("M", seed)
        clouds.append(syn.surface_points(v, f, P, seed=seed))
    R, T = r1.look_at_view_transform(torch.from_numpy(di), torch.from_numpy(el), torch.from_numpy(az))
    W2V = r1.world_to_view_matrix(R, T)
    Kp = ortho_projection_matrix(V)
    obj = bg = (0.99999, 0.99999, 0.99999)
    out = {}
    for K in (1, 4):
        idx_all, img_all = [], []
        for b, pts in enumerate(clouds):
            p = torch.from_numpy(pts)[None].expand(V, -1, -1) * (1.0 / torch.from_numpy(di))[:, None, None]   # Pointclouds.scale_
            view = r1.transform_points(p, W2V)
            ndc = r1.transform_points(view, Kp)
            ndc[..., 2] = view[..., 2]
            feats = torch.tensor(obj)[None, :] * torch.ones(P, 3)
            for n in range(V):
                idx, zb, d2 = rasterize_points(ndc[n], S, radius, K)
                img_all.append(norm_weighted_composite(idx, d2, feats, radius, bg))
                idx_all.append(idx)
            print("points K=%d cloud %d: covered %.4f" % (K, b, float((torch.stack(idx_all[-V:])[..., 0] >= 0).float().mean())))
        out["idx_k%d" % K] = torch.stack(idx_all).numpy().astype(np.int16)
        out["images_k%d" % K] = torch.stack(img_all).numpy().astype(np.float32).reshape(2, V, 3, S, S)
    path = os.path.join(HERE, "torch_restatement_points.npz")
    np.savez_compressed(path, R=R.numpy(), T=T.numpy(), azim=az, elev=el, dist=di, image_size=np.array(S), radius=np.array(radius),
                        seeds=np.array([0, 1]), n_points=np.array(P), **out)
    print("wrote", path, os.path.getsize(path), "bytes")


# ---- mesh rasteriser with blur / K / clipped barycentrics (RasterizeMeshesNaive, CheckPixelInsideFace) ---------------------

def point_line_dist2(px, py, ax, ay, bx, by):
    """PointLineDistanceForward: squared distance from p to the SEGMENT a b."""
    bax, bay = bx - ax, by - ay
    l2 = bax * bax + bay * bay
    t = (bax * (px - ax) + bay * (py - ay)) / l2
    t = torch.clamp(t, 0.0, 1.0)
    qx, qy = ax + t * bax, ay + t * bay
    d = (px - qx) * (px - qx) + (py - qy) * (py - qy)
    dd = (px - bx) * (px - bx) + (py - by) * (py - by)
    return torch.where(l2 <= KEPS, dd, d)


def rasterize_faces(fv, S, blur_radius, K, clip_bary, cull_backfaces=False, chunk=64, rows_per_chunk=32):
    """fv: (F, 3, 3) face vertices (NDC x, y, view z).  Per pixel the K nearest faces passing CheckPixelInsideFace
    (perspective-correct barycentrics).  Returns pix_to_face (S,S,K), zbuf, bary (S,S,K,3), dists."""
    c = pix_centres(S)
    blur = math.sqrt(blur_radius)
    Fn = fv.shape[0]
    p2f = torch.full((S, S, K), -1, dtype=torch.int64)
    zbuf = torch.full((S, S, K), -1.0)
    bary = torch.full((S, S, K, 3), -1.0)
    dists = torch.full((S, S, K), -1.0)
    inf = torch.tensor(float("inf"))
    v0, v1, v2 = fv[:, 0][None], fv[:, 1][None], fv[:, 2][None]          # (1, F, 3)
    xmin = torch.minimum(torch.minimum(v0[..., 0], v1[..., 0]), v2[..., 0]) - blur
    xmax = torch.maximum(torch.maximum(v0[..., 0], v1[..., 0]), v2[..., 0]) + blur
    ymin = torch.minimum(torch.minimum(v0[..., 1], v1[..., 1]), v2[..., 1]) - blur
    ymax = torch.maximum(torch.maximum(v0[..., 1], v1[..., 1]), v2[..., 1]) + blur
    zmin = torch.minimum(torch.minimum(v0[..., 2], v1[..., 2]), v2[..., 2])
    zmax = torch.maximum(torch.maximum(v0[..., 2], v1[..., 2]), v2[..., 2])
    face_area = r1.edge(v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1], v2[..., 0], v2[..., 1])
    zero_area = (face_area <= KEPS) & (face_area >= -KEPS)
    back = face_area < 0
    area = r1.edge(v2[..., 0], v2[..., 1], v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1]) + KEPS
    for r0 in range(0, S, rows_per_chunk):
        rows = min(rows_per_chunk, S - r0)
        py = c[r0:r0 + rows, None].expand(rows, S).reshape(-1, 1)
        px = c[None, :].expand(rows, S).reshape(-1, 1)
        outside = (px > xmax) | (px < xmin) | (py > ymax) | (py < ymin) | (zmin < KEPS)
        skip = (zmax < 0) | outside | zero_area | (back if cull_backfaces else torch.tensor(False))
        w0 = r1.edge(px, py, v1[..., 0], v1[..., 1], v2[..., 0], v2[..., 1]) / area
        w1 = r1.edge(px, py, v2[..., 0], v2[..., 1], v0[..., 0], v0[..., 1]) / area
        w2 = r1.edge(px, py, v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1]) / area
        t0 = w0 * v1[..., 2] * v2[..., 2]
        t1 = v0[..., 2] * w1 * v2[..., 2]
        t2 = v0[..., 2] * v1[..., 2] * w2
        den = torch.clamp(t0 + t1 + t2, min=KEPS)
        w0, w1, w2 = t0 / den, t1 / den, t2 / den
        if clip_bary:                                                   # BarycentricClipForward
            c0, c1, c2 = torch.clamp(w0, 0.0, 1.0), torch.clamp(w1, 0.0, 1.0), torch.clamp(w2, 0.0, 1.0)
            sm = torch.clamp(c0 + c1 + c2, min=1e-5)
            c0, c1, c2 = c0 / sm, c1 / sm, c2 / sm
        else:
            c0, c1, c2 = w0, w1, w2
        pz = c0 * v0[..., 2] + c1 * v1[..., 2] + c2 * v2[..., 2]
        e01 = point_line_dist2(px, py, v0[..., 0], v0[..., 1], v1[..., 0], v1[..., 1])
        e02 = point_line_dist2(px, py, v0[..., 0], v0[..., 1], v2[..., 0], v2[..., 1])
        e12 = point_line_dist2(px, py, v1[..., 0], v1[..., 1], v2[..., 0], v2[..., 1])
        dist = torch.minimum(torch.minimum(e01, e02), e12)
        inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
        sdist = torch.where(inside, -dist, dist)
        hit = ~skip & ~(pz < 0) & (inside | (dist < blur_radius))
        zm = torch.where(hit, pz, inf)
        # K smallest by (z, face index): a STABLE sort keeps the lower face index first among equal depths — with clipped
        # barycentrics neighbouring faces give bit-equal z at a shared vertex / edge, and the kernel (faces visited in index
        # order, a later face replaces the current farthest only when strictly nearer) keeps the lower indices at the K-th place
        zs_all, order_all = torch.sort(zm, dim=1, stable=True)
        zs, order = zs_all[:, :K], order_all[:, :K]
        ok = torch.isfinite(zs)
        g = lambda t: torch.where(ok, torch.gather(t, 1, order), torch.tensor(-1.0))   # noqa: E731
        p2f[r0:r0 + rows] = torch.where(ok, order, torch.tensor(-1)).view(rows, S, K)
        zbuf[r0:r0 + rows] = g(pz).view(rows, S, K)
        dists[r0:r0 + rows] = g(sdist).view(rows, S, K)
        bary[r0:r0 + rows] = torch.stack([g(c0), g(c1), g(c2)], -1).view(rows, S, K, 3)
    return p2f, zbuf, bary, dists


# ---- clip_faces (pytorch3d/renderer/mesh/clip.py) ---------------------------------------------------------------------

def find_verts_intersecting_clipping_plane(fv, p1_face_ind, clip_value):
    Tn = fv.shape[0]
    p2_face_ind = torch.remainder(p1_face_ind + 1, 3)
    p3_face_ind = torch.remainder(p1_face_ind + 2, 3)
    ar = torch.arange(Tn)
    p1, p2, p3 = fv[ar, p1_face_ind], fv[ar, p2_face_ind], fv[ar, p3_face_ind]
    w2 = (p1[:, [2]] - clip_value) / (p1[:, [2]] - p2[:, [2]])
    w3 = (p1[:, [2]] - clip_value) / (p1[:, [2]] - p3[:, [2]])
    # perspective_correct: interpolate in "world" coordinates (x z, y z, z), project back by the interpolated z
    p1w = torch.cat((p1[:, :2] * p1[:, [2]], p1[:, [2]]), dim=1)
    p2w = torch.cat((p2[:, :2] * p2[:, [2]], p2[:, [2]]), dim=1)
    p3w = torch.cat((p3[:, :2] * p3[:, [2]], p3[:, [2]]), dim=1)
    p4 = p1w * (1 - w2) + p2w * w2
    p5 = p1w * (1 - w3) + p3w * w3
    p4 = torch.cat((p4[:, :2] / p4[:, [2]], p4[:, [2]]), dim=1)
    p5 = torch.cat((p5[:, :2] / p5[:, [2]], p5[:, [2]]), dim=1)
    p4b = torch.zeros(Tn, 3)
    p4b[ar, p1_face_ind] = (1 - w2)[:, 0]
    p4b[ar, p2_face_ind] = w2[:, 0]
    p5b = torch.zeros(Tn, 3)
    p5b[ar, p1_face_ind] = (1 - w3)[:, 0]
    p5b[ar, p3_face_ind] = w3[:, 0]
    return p4, p5, p4b, p5b, p2, p3, p1, p2_face_ind, p3_face_ind


def clip_faces(fv, z_clip):
    """-> clipped face vertices (Fc, 3, 3), their original face (Fc,), barycentric conversion (Fc, 3, 3): row k = barycentrics
    of clipped vertex k in the original face."""
    behind = fv[:, :, 2] < z_clip
    nb = behind.sum(1)
    eye = torch.eye(3)
    out_v, out_f, out_b = [], [], []
    keep = torch.nonzero(nb == 0)[:, 0]
    out_v.append(fv[keep]); out_f.append(keep); out_b.append(eye[None].expand(len(keep), 3, 3))
    one = torch.nonzero(nb == 1)[:, 0]           # one vertex behind: quad -> (p4, p2, p5), (p5, p2, p3)
    if len(one):
        p1i = behind[one].to(torch.int64).argmax(1)
        p4, p5, p4b, p5b, p2, p3, _, p2i, p3i = find_verts_intersecting_clipping_plane(fv[one], p1i, z_clip)
        e2, e3 = eye[p2i], eye[p3i]
        out_v += [torch.stack((p4, p2, p5), 1), torch.stack((p5, p2, p3), 1)]
        out_f += [one, one]
        out_b += [torch.stack((p4b, e2, p5b), 1), torch.stack((p5b, e2, e3), 1)]
    two = torch.nonzero(nb == 2)[:, 0]           # two behind: (p1, p4, p5), p1 the vertex in front
    if len(two):
        p1i = (~behind[two]).to(torch.int64).argmax(1)
        p4, p5, p4b, p5b, _, _, p1, _, _ = find_verts_intersecting_clipping_plane(fv[two], p1i, z_clip)
        out_v.append(torch.stack((p1, p4, p5), 1)); out_f.append(two); out_b.append(torch.stack((eye[p1i], p4b, p5b), 1))
    return torch.cat(out_v), torch.cat(out_f), torch.cat(out_b), int(len(one) + len(two))


def mesh_setup(tier, seed, az, el, di):
    v_np, f_np = syn.synthetic_mesh(tier, seed)
    verts, faces = torch.from_numpy(v_np), torch.from_numpy(f_np)
    V = len(az)
    R, T = r1.look_at_view_transform(torch.from_numpy(di), torch.from_numpy(el), torch.from_numpy(az))
    W2V = r1.world_to_view_matrix(R, T)
    Kp = r1.fov_projection_matrix(V)
    view = r1.transform_points(verts[None].expand(V, -1, -1), W2V)
    ndc = r1.transform_points(view, Kp)
    ndc[..., 2] = view[..., 2]
    return verts, faces, R, T, W2V, ndc, r1.verts_normals(verts, faces), torch.inverse(W2V)[:, 3, :3]


def make_zclip():
    S = 128
    az = np.array([20.0, 200.0], np.float32)
    el = np.array([25.0, -40.0], np.float32)
    di = np.array([1.1, 1.1], np.float32)
    verts, faces, R, T, W2V, ndc, nrm, centers = mesh_setup("S", 5, az, el, di)
    obj = bg = (0.99999, 0.99999, 0.99999)
    p2f_all, img_all, cut_all = [], [], []
    for n in range(len(az)):
        fv, orig, conv, n_cut = clip_faces(ndc[n][faces], 0.5)
        p2f_c, zb, bary_c, _ = rasterize_faces(fv, S, 0.0, 1, False)
        p2f_c, bary_c = p2f_c[..., 0], bary_c[..., 0, :]
        mask = p2f_c >= 0
        fc = p2f_c.clamp(min=0)
        # convert_clipped_rasterization_to_original_faces: original face index, barycentrics through the conversion matrices
        bary_o = torch.bmm(bary_c.reshape(-1, 1, 3), conv[fc.reshape(-1)]).reshape(S, S, 3)
        p2f = torch.where(mask, orig[fc], torch.tensor(-1))
        img_all.append(r1.phong(p2f, bary_o, verts, faces, nrm, centers[n], (0.0, 1.0, 0.0), obj, bg))
        p2f_all.append(p2f)
        cut_all.append(n_cut)
        print("zclip view", n, "covered", float(mask.float().mean()), "faces cut", n_cut)
    path = os.path.join(HERE, "torch_restatement_zclip.npz")
    np.savez_compressed(path, pix_to_face=torch.stack(p2f_all).numpy().astype(np.int16), images=torch.stack(img_all).numpy().astype(np.float32),
                        straddlers=np.array(cut_all), R=R.numpy(), T=T.numpy(), azim=az, elev=el, dist=di, image_size=np.array(S),
                        mesh=np.array(["S", "5"]), z_clip=np.array(0.5))
    print("wrote", path, os.path.getsize(path), "bytes")


def phong_fragments(p2f, bary, verts, faces, normals, cam_center, light_dir, obj):
    """phong_shading for every fragment: (S, S, K, 3) colours (garbage where p2f < 0: the blend masks them)."""
    S, _, K = p2f.shape
    f = p2f.clamp(min=0)
    fverts, fnorm = verts[faces][f], normals[faces][f]                  # (S, S, K, 3, 3)
    fcol = torch.tensor(obj)[None, None, None, None, :].expand(S, S, K, 3, 3)
    pixel_coords = (bary[..., None] * fverts).sum(dim=-2)
    pixel_normals = (bary[..., None] * fnorm).sum(dim=-2)
    texels = (bary[..., None] * fcol).sum(dim=-2)
    nrm = F.normalize(pixel_normals, p=2, dim=-1, eps=1e-6)
    direction = F.normalize(torch.tensor(light_dir)[None, None, None, :].expand(S, S, K, 3), p=2, dim=-1, eps=1e-6)
    cos_angle = torch.sum(nrm * direction, dim=-1)
    diffuse = torch.tensor([0.3, 0.3, 0.3]) * F.relu(cos_angle)[..., None]
    lit = (cos_angle > 0).to(torch.float32)
    view_direction = F.normalize(cam_center[None, None, None, :] - pixel_coords, p=2, dim=-1, eps=1e-6)
    reflect = -direction + 2 * (cos_angle[..., None] * nrm)
    alpha = F.relu(torch.sum(view_direction * reflect, dim=-1)) * lit
    specular = torch.tensor([0.2, 0.2, 0.2]) * torch.pow(alpha, 64.0)[..., None]
    ambient = torch.tensor([1.0, 1.0, 1.0]) * torch.tensor([0.5, 0.5, 0.5])
    return (ambient + diffuse) * texels + specular


def softmax_rgb_blend(colors, p2f, dists, zbuf, bg, sigma, gamma, znear=1.0, zfar=100.0):
    eps = 1e-10
    mask = p2f >= 0
    prob_map = torch.sigmoid(-dists / sigma) * mask
    alpha = torch.prod((1.0 - prob_map), dim=-1)
    z_inv = (zfar - zbuf) / (zfar - znear) * mask
    z_inv_max = torch.max(z_inv, dim=-1).values[..., None].clamp(min=eps)
    weights_num = prob_map * torch.exp((z_inv - z_inv_max) / gamma)
    delta = torch.exp((eps - z_inv_max) / gamma).clamp(min=eps)
    denom = weights_num.sum(dim=-1)[..., None] + delta
    weighted_colors = (weights_num[..., None] * colors).sum(dim=-2)
    weighted_background = delta * torch.tensor(bg)
    rgb = (weighted_colors + weighted_background) / denom
    return torch.cat([rgb, (1.0 - alpha)[..., None]], -1).permute(2, 0, 1)


def make_soft():
    S, K, sigma, gamma = 64, 4, 1e-4, 1e-4
    blur_radius = float(np.log(1.0 / 1e-4 - 1.0) * sigma)
    az = np.array([35.0, 250.0], np.float32)
    el = np.array([20.0, -30.0], np.float32)
    di = np.array([2.2, 2.0], np.float32)
    verts, faces, R, T, W2V, ndc, nrm, centers = mesh_setup("S", 2, az, el, di)
    obj, bg = (0.99999, 0.99999, 0.99999), (0.99999, 0.99999, 0.99999)
    p2f_all, img_all, d_all = [], [], []
    for n in range(len(az)):
        p2f, zb, bary, dists = rasterize_faces(ndc[n][faces], S, blur_radius, K, True)
        colors = phong_fragments(p2f, bary, verts, faces, nrm, centers[n], (0.0, 1.0, 0.0), obj)
        img_all.append(softmax_rgb_blend(colors, p2f, dists, zb, bg, sigma, gamma))
        p2f_all.append(p2f)
        d_all.append(dists)
        print("soft view", n, "pixels with a fragment", float((p2f[..., 0] >= 0).float().mean()), "with 4", float((p2f[..., 3] >= 0).float().mean()))
    path = os.path.join(HERE, "torch_restatement_soft.npz")
    np.savez_compressed(path, pix_to_face=torch.stack(p2f_all).numpy().astype(np.int16), images=torch.stack(img_all).numpy().astype(np.float32),
                        dists=torch.stack(d_all).numpy().astype(np.float32), R=R.numpy(), T=T.numpy(), azim=az, elev=el, dist=di,
                        image_size=np.array(S), mesh=np.array(["S", "2"]), blur_radius=np.array(blur_radius), faces_per_pixel=np.array(K),
                        sigma=np.array(sigma), gamma=np.array(gamma))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    which = sys.argv[1:] or ["points", "zclip", "soft"]
    if "zclip" in which:
        make_zclip()
    if "soft" in which:
        make_soft()
    if "points" in which:
        make_points()
