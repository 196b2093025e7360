"""Shared helpers of the -m gpu tests (tests/test_gpu_*.py, filed by component in round 6): device tensors, seeded batches and views,
finite-difference gradients of the oracle, fixture loading.  Test infrastructure; imports the product only lazily inside the helpers."""
import os

import numpy as np
import pytest  # noqa: F401
import torch

from mvtn_amd import synthetic as syn

from reg_composition import regularize_composition as _regularize_composition  # noqa: E402,F401  (tests/reg_composition.py)

TOL_PIXEL = 1e-4  # north_star tolerance


def _dev():
    assert torch.cuda.is_available(), "GPU test needs a GPU"
    return torch.device("cuda:0")


def _t(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(_dev())


def _mesh_batch(specs):
    vl, fl = [], []
    for tier, seed in specs:
        v, f = syn.cad_soup(seed) if tier == "soup" else syn.synthetic_mesh(tier, seed)
        vl.append(v)
        fl.append(f)
    return vl, fl


def _well_conditioned_pixels(vl, fl, p2f, V):
    """Pixels whose winning face has three WELL-CONDITIONED vertex normals.  Where the area-weighted face
    normals of a vertex cancel (double-sided CAD quads: |sum n_f| << sum |n_f|) the normalised sum is
    rounding noise in ANY implementation (it depends on summation order; PyTorch3D's own CPU index_add and
    GPU atomics disagree there), so image parity is only asserted on the rest; coverage parity on all."""
    mask = np.ones(p2f.shape, bool)
    for b, (v, f) in enumerate(zip(vl, fl)):
        v64 = v.astype(np.float64)
        nf = np.cross(v64[f[:, 2]] - v64[f[:, 1]], v64[f[:, 0]] - v64[f[:, 1]])
        s, a = np.zeros_like(v64), np.zeros(len(v))
        for k in range(3):
            np.add.at(s, f[:, k], nf)
            np.add.at(a, f[:, k], np.linalg.norm(nf, axis=1))
        well_v = np.linalg.norm(s, axis=1) > 1e-3 * np.maximum(a, 1e-30)
        well_f = well_v[f].all(1)
        sl = p2f[b * V:(b + 1) * V]
        m = np.ones(sl.shape, bool)
        m[sl >= 0] = well_f[sl[sl >= 0]]
        mask[b * V:(b + 1) * V] = m
    return mask


def _views(B, V, seed, jitter=True):
    az, el, di = syn.spherical_views(V)
    az, el, di = np.tile(az, (B, 1)), np.tile(el, (B, 1)), np.tile(di, (B, 1))
    if jitter:
        rng = np.random.default_rng(seed)
        az = az + rng.uniform(-5, 5, az.shape)
        el = el + rng.uniform(-5, 5, el.shape)
        di = di + rng.uniform(-0.1, 0.1, di.shape)
    return az.astype(np.float32), el.astype(np.float32), di.astype(np.float32)


def _fd_grad_full(oracle, vl, fl, az, el, di, size, G, p2f, relative, persp, h=1e-5, z_clip=0.0):
    """fp64 central differences of L = sum(G * image) w.r.t. (azim, elev, dist), pix_to_face fixed."""
    vl64 = [v.astype(np.float64) for v in vl]

    def L(a, e, d):
        ld = oracle.camera_position(a, e, d, np.float64) if relative else np.array([[0.0, 1.0, 0.0]])
        o = oracle.render_meshes(vl64, fl, a, e, d, size, light_dir=ld, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3,
                                 perspective_correct=persp, dtype=np.float64, fixed_p2f=p2f, return_fragments=False, z_clip=z_clip)
        return float((o["images"] * G).sum())

    base = [az.astype(np.float64), el.astype(np.float64), di.astype(np.float64)]
    grads = []
    for k in range(3):
        g = np.zeros(base[k].shape)
        for idx in np.ndindex(*base[k].shape):
            p = [x.copy() for x in base]
            m = [x.copy() for x in base]
            p[k][idx] += h
            m[k][idx] -= h
            g[idx] = (L(*p) - L(*m)) / (2 * h)
        grads.append(g)
    return grads


def _points_grad_reference(oracle, pts, colors, obj, az, el, di, size, r, G, idx, dist2):
    """Reference camera gradients of the point path: PyTorch3D's compositor weight gradient (restated in
    oracle.norm_weighted_weight_grads) chained through fp64 central differences of the oracle's weights
    w_k = 1 - dist2_k / r^2 at a fixed pixel->points map.  The compositor coefficients use the fp32 fragments
    `dist2` (what the reference's backward would see: near the clamp 1e-4 - w is an fp32 cancellation)."""
    K = 1 if idx.ndim == 3 else idx.shape[-1]
    N = idx.shape[0]
    pts64 = pts.astype(np.float64)
    col64 = None if colors is None else colors.astype(np.float64)
    B, V = az.shape

    def frags(a, e, d):
        o = oracle.render_points(pts64, a, e, d, size, radius=r, obj_color=obj, bg_color=(0, 0, 0), point_colors=col64,
                                 dtype=np.float64, fixed_idx=idx, points_per_pixel=K)
        return o["dist2"].reshape(N, size, size, K)
    idx4 = idx.reshape(N, size, size, K)
    base = [az.astype(np.float64), el.astype(np.float64), di.astype(np.float64)]
    d2 = np.asarray(dist2, np.float64).reshape(N, size, size, K)
    if colors is None:
        feats = np.broadcast_to(np.asarray(obj, np.float64), idx4.shape + (3,))
    else:
        b_of_n = np.repeat(np.arange(B), V)
        feats = col64[b_of_n[:, None, None, None], np.maximum(idx4, 0)]
    gw = oracle.norm_weighted_weight_grads(idx4, d2, feats, np.asarray(G, np.float64).reshape(N, 3, size, size), r)
    h = 1e-6
    out = []
    for k in range(3):
        g = np.zeros((B, V))
        for b in range(B):
            for v in range(V):
                p = [x.copy() for x in base]
                m = [x.copy() for x in base]
                p[k][b, v] += h
                m[k][b, v] -= h
                dw = -(frags(*p)[b * V + v] - frags(*m)[b * V + v]) / (2 * h) / (r * r)
                g[b, v] = (np.where(idx4[b * V + v] >= 0, dw, 0.0) * gw[b * V + v]).sum()
        out.append(g)
    return out


def _render_pair(oracle, vl, fl, az, el, di, size, strip=0, persp=True):
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    B, V = az.shape
    obj, bg = (0.9, 0.8, 0.7), (0.1, 0.2, 0.3)
    want = oracle.render_meshes(vl, fl, az, el, di, size, obj_color=obj, bg_color=bg, perspective_correct=persp, mode=1)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.mesh_params(V, size, 1, obj, bg, perspective_correct=persp, strip_rows=strip)
    img, p2f = F.render_meshes_fragments(packed, R, T, _t([[0, 1, 0]]), params)
    return img.cpu().numpy(), p2f.cpu().numpy(), want


def _hip_render_mesh(v, f, az, el, di, size, light=(0.0, 1.0, 0.0), obj=(1, 1, 1), bg=(0, 0, 0), persp=True):
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(np.asarray(v, np.float32))], [torch.from_numpy(np.asarray(f, np.int64))]), _dev())
    R, T = F.look_at_view_transform(_t([di]), _t([el]), _t([az]))
    params = F.mesh_params(1, size, 1, obj, bg, perspective_correct=persp)
    img, p2f = F.render_meshes_fragments(packed, R, T, _t([list(light)]), params)
    return img[0, 0].cpu().numpy(), p2f[0].cpu().numpy()


def _tris(tris):
    verts = np.asarray(tris, np.float32).reshape(-1, 3)
    return verts, np.arange(len(verts)).reshape(-1, 3).astype(np.int64)


def _soft_blur(sigma):
    return float(np.log(1.0 / 1e-4 - 1.0) * sigma)      # the PyTorch3D tutorials' choice: prob 1e-4 at the rim


def _fd_grad_soft(oracle, vl, fl, az, el, di, size, G, p2f, relative, blur, K, shader, sigma, h=1e-5):
    vl64 = [v.astype(np.float64) for v in vl]

    def L(a, e, d):
        ld = oracle.camera_position(a, e, d, np.float64) if relative else np.array([[0.0, 1.0, 0.0]])
        o = oracle.render_meshes_soft(vl64, fl, a, e, d, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, light_dir=ld,
                                      obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3, dtype=np.float64, fixed_p2f=p2f)
        return float((o["images"] * G).sum())
    base = [az.astype(np.float64), el.astype(np.float64), di.astype(np.float64)]
    grads = []
    for k in range(3):
        g = np.zeros(base[k].shape)
        for idx in np.ndindex(*base[k].shape):
            p, m = [x.copy() for x in base], [x.copy() for x in base]
            p[k][idx] += h
            m[k][idx] -= h
            g[idx] = (L(*p) - L(*m)) / (2 * h)
        grads.append(g)
    return grads


def _write_off(path, v, f, glued=False):
    with open(path, "w") as fh:
        fh.write(("OFF%d %d 0\n" if glued else "OFF\n%d %d 0\n") % (len(v), len(f)))
        for p in v:
            fh.write("%.9g %.9g %.9g\n" % tuple(p))
        for t in f:
            fh.write("3 %d %d %d\n" % tuple(t))


def _write_obj(path, v, f):
    with open(path, "w") as fh:
        fh.write("# round-3 fixture\n")
        for p in v:
            fh.write("v %.9g %.9g %.9g\n" % tuple(p))
        for k, t in enumerate(f):
            if k % 2:
                fh.write("f %d/1/1 %d/2/2 %d/3/3\n" % tuple(int(i) + 1 for i in t))
            else:
                fh.write("f %d %d %d\n" % tuple(int(i) - len(v) for i in t))       # negative (relative) indices


def _fx(name):
    return np.load(os.path.join(os.path.dirname(__file__), "golden", name))


def _hip_grads(vl, fl, az, el, di, S, G, relative_light, gather):
    """d sum(G * images) / d (azim, elev, dist) from the HIP path (one autograd node per direction), (B, V) each; and the
    pixel -> face maps of the same forward."""
    from mvtn_amd import Meshes, functional as F
    from mvtn_amd.structures import PackedMeshes
    B, V = az.shape
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    a, e, d = (_t(x.reshape(-1)).requires_grad_(True) for x in (az, el, di))
    light = None if relative_light else _t([[0.0, 1.0, 0.0]])
    params = F.mesh_params(V, S, B * V if relative_light else 1, (0.99999,) * 3, (0.99999,) * 3, gather_backward=gather)
    images, R, T, _ = F.render_mesh_views(packed, a, e, d, light, params)
    (images * _t(G).reshape(images.shape)).sum().backward()
    with torch.no_grad():
        lt = F.camera_position_from_spherical_angles(d, e, a) if relative_light else light
        _, p2f = F.render_meshes_fragments(packed, R.detach(), T.detach(), lt.detach(), params)
    return [x.grad.cpu().numpy().reshape(B, V).astype(np.float64) for x in (a, e, d)], p2f.cpu().numpy()


def _oracle_both_normal_sums(oracle, vl, fl, az, el, di, S, **kw):
    """(oracle render with ORDER-INDEPENDENT vertex-normal sums — the library's semantics —, oracle render with the default
    face-order float sums, (N, H, W) mask of the pixels whose winning face touches a vertex whose face normals cancel).
    At those vertices (coincident reversed shells on shared vertices) a float sum is rounding noise of its order (PyTorch3D on the
    GPU: atomics in arrival order), the normalised normal ~1e-2 long in a random direction, the interpolated normal renormalised
    to unit length: the diffuse / specular terms of such pixels are not a function of the input.  Everywhere else the two
    sums agree to ~3e-6 in the colours."""
    oracle.set_normals_exact_sum(True)
    try:
        exact = oracle.render_meshes(vl, fl, az, el, di, S, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3, mode=1, **kw)
    finally:
        oracle.set_normals_exact_sum(False)
    default = oracle.render_meshes(vl, fl, az, el, di, S, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3, mode=1, **kw)
    assert np.array_equal(exact["pix_to_face"], default["pix_to_face"])
    V = az.shape[1]
    touch = np.zeros(default["pix_to_face"].shape, bool)
    for b, (v, f) in enumerate(zip(vl, fl)):
        face_c = oracle.cancelled_normal_vertices(v, f)[f].any(axis=1)
        p = default["pix_to_face"][b * V:(b + 1) * V]
        touch[b * V:(b + 1) * V] = np.where(p >= 0, face_c[np.maximum(p, 0)], False)
    return exact, default, touch


def _render(vl, fl, az, el, di, S, **kw):
    from mvtn_amd import Meshes, functional as F
    from mvtn_amd.structures import PackedMeshes
    B, V = az.shape
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.mesh_params(V, S, 1, (0.99999,) * 3, (0.99999,) * 3, **kw)
    img, p2f, st = F.render_meshes_fragments(packed, R, T, _t([[0.0, 1.0, 0.0]]), params, return_stats=True)
    return img.cpu().numpy(), p2f.cpu().numpy(), [int(x) for x in st.cpu()]
