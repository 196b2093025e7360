"""Generates tests/golden/torch_restatement_r5.npz: the torch restatements of rounds 2-4 — real torch ops in
PyTorch3D's op order, forward in fp32 (make_torch_restatement.py) and torch.autograd in float64 with the pixel -> face map held
fixed (make_torch_restatement_grad.py) — on the geometry classes round 5 added:

  cad     one CAD-like object (synthetic.cad_mesh seed 7, 1 280 faces: two-triangle flat sides, slivers, coincident reversed shells
          on shared and on duplicated vertices, interior parts), 4 circular views, 128 x 128: pixel -> face map, fp32 images, and
          d sum(G images) / d (azim, elev, dist) with the RELATIVE light;
  ragged  a tier-S and a tier-M mesh (seeds 3, 4) — a ragged pair through the packer's offsets —, 3 views each, 96 x 96: maps + images.

The torch rasteriser keeps the smallest (z, face index) per pixel: on the coincident shells this is the tie rule of SURVEY A.4 as a
second implementation.  Vertex normals: a float32 index_add here; at vertices whose face normals cancel (shared-vertex shells) that
is rounding noise (see oracle/mvr_oracle_impl.inc orc_vertex_normals): the tests mask the pixels of faces touching such a vertex
when they compare fp32 IMAGES; the float64 autograd pass sums them to ~1e-17 and needs no mask.

    python tests/golden/make_torch_restatement_r5.py        (CPU, ~5 min)
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


def forward_fp32(v_np, f_np, az, el, di, S):
    V = len(az)
    verts, faces = torch.from_numpy(v_np), torch.from_numpy(f_np)
    R, T = r1.look_at_view_transform(torch.from_numpy(di), torch.from_numpy(el), torch.from_numpy(az))
    W2V = r1.world_to_view_matrix(R, T)
    K = r1.fov_projection_matrix(V)
    view = r1.transform_points(verts[None].expand(V, -1, -1), W2V)
    ndc = r1.transform_points(view, K)
    ndc[..., 2] = view[..., 2]
    nrm = r1.verts_normals(verts, faces)
    centers = torch.inverse(W2V)[:, 3, :3]
    obj = bg = (0.99999, 0.99999, 0.99999)
    p2fs, imgs = [], []
    for n in range(V):
        p2f, _, bary = r1.rasterize(ndc[n], faces, S)
        imgs.append(r1.phong(p2f, bary, verts, faces, nrm, centers[n], (0.0, 1.0, 0.0), obj, bg))
        p2fs.append(p2f)
        print("  view", n, "covered", float((p2f >= 0).float().mean()), flush=True)
    return torch.stack(p2fs).numpy().astype(np.int16), torch.stack(imgs).numpy().astype(np.float32)


def main():
    torch.set_num_threads(8)
    out = {}
    # ---- cad
    v, f = syn.cad_mesh(7, n_faces=1280)
    az, el, di = syn.circular_views(4, 30.0, 2.2)
    p2f, img = forward_fp32(v, f, az, el, di, 128)
    rng = np.random.default_rng(3)
    azj = (az + rng.uniform(-5, 5, 4)).astype(np.float32)
    elj = (el + rng.uniform(-5, 5, 4)).astype(np.float32)
    dij = (di + rng.uniform(-0.1, 0.1, 4)).astype(np.float32)
    p2g, _, ga, ge, gd = rg.render_views_diff(torch.from_numpy(v), torch.from_numpy(f), azj, elj, dij, 128, True, 300)[:5]
    out.update(cad_seed=np.array(7), cad_faces=np.array(1280), cad_image_size=np.array(128), cad_azim=az, cad_elev=el, cad_dist=di,
               cad_pix_to_face=p2f, cad_images=img, cad_g_azim=azj, cad_g_elev=elj, cad_g_dist=dij, cad_g_pix_to_face=p2g,
               cad_grad_azim=ga, cad_grad_elev=ge, cad_grad_dist=gd, cad_g_seed=np.array(300))
    # ---- ragged pair
    azr, elr, dir_ = (x[:3] for x in syn.spherical_views(12))
    for name, tier, seed in (("rag_s", "S", 3), ("rag_m", "M", 4)):
        vr, fr = syn.synthetic_mesh(tier, seed)
        p, i = forward_fp32(vr, fr, azr, elr, dir_, 96)
        out[name + "_pix_to_face"] = p
        out[name + "_images"] = i
    out.update(rag_azim=azr, rag_elev=elr, rag_dist=dir_, rag_image_size=np.array(96), rag_seeds=np.array([3, 4]))
    path = os.path.join(HERE, "torch_restatement_r5.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
