"""oracle/oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy/ctypes front-end of the CPU oracle (``liboracle.so``, built from
``mvr_oracle.c`` by ``oracle/Makefile``).  It restates, on the CPU, the
PyTorch3D arithmetic that the reference's ``models/renderer.py:65-198`` calls
(look-at, projection, naive z-buffered rasterisation, Phong + hard blend, point
splat + norm-weighted compositing).  PyTorch3D is not vendored by the reference
and not installable here, so this oracle is **parity unpinned** (SURVEY.md
section 8c): it is checked against analytic known-answer tests and the golden
caller-side fixtures only.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product (``mvtn_amd``) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("mvr_oracle.c", "mvr_oracle_impl.inc", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def set_threads(n):
    """Threads of the literal pixel-major loops (mode 0).  1 (default) is the faithful stand-in for PyTorch3D's
    single-threaded CPU rasteriser; > 1 only for bench.py's "all host cores" line."""
    lib().orc_set_threads(ctypes.c_int(int(n)))


def set_options(normals_three_cross=True, fov_scale=0.0):
    """The two switches of VERDICT r2 (Weak 1): the vertex normals as three different cross products per face (slots 1, 2, 0)
    instead of one added three times, and the projection scale s given explicitly (e.g. K[0][0] of a PyTorch3D build, whose
    fp32 tan may differ in the last bit from RN(1 / tan(fov / 2))); no arguments = the defaults (three cross products: the form of
    the PyTorch3D releases of the reference's era; s from fov).  Process-wide."""
    lib().orc_set_options(ctypes.c_int(int(bool(normals_three_cross))), ctypes.c_double(float(fov_scale)))


def set_edge_fma(on=False):
    """Switch #3 (round 4): the per-pixel face test with its sums of products contracted into FMAs the way nvcc compiles
    PyTorch3D's CUDA kernels (orc_set_edge_fma in mvr_oracle.c); False = the plain IEEE sequence (default).  Process-wide."""
    lib().orc_set_edge_fma(ctypes.c_int(int(bool(on))))


def set_normals_exact_sum(on=False):
    """Switch #4 (round 5): the vertex-normal sums accumulated order-independently in 64-bit fixed point (what libmvr_hip.so
    does) instead of as a float index_add in face order (default; PyTorch3D on the GPU: float atomics in arrival order).  The
    forms differ visibly only at vertices whose face normals CANCEL (coincident reversed shells on shared vertices), where the
    float sums are rounding noise (orc_vertex_normals in mvr_oracle_impl.inc).  Process-wide."""
    lib().orc_set_normals_exact_sum(ctypes.c_int(int(bool(on))))


def set_projection_bmm(on=False):
    """Switch #5 (round 6): True = the world -> view transform accumulated like a K = 4 GPU matrix product (ascending k, fused
    multiply-adds: PyTorch3D's transform_points is a torch.bmm); False = the closed form with every operation rounded (default,
    what the library computes).  Moves only which of two COINCIDENT faces wins a pixel (orc_set_projection_bmm in mvr_oracle.c).
    Process-wide."""
    lib().orc_set_projection_bmm(ctypes.c_int(int(bool(on))))


def cancelled_normal_vertices(verts, faces, rel=1e-4):
    """Boolean (Vn,) mask of the vertices whose area-weighted face normals cancel: |sum_f n_f| < rel * sum_f |n_f| (float64)."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    n = np.cross(c - b, a - b)
    acc, mag = np.zeros_like(v), np.zeros(len(v))
    for k in range(3):
        np.add.at(acc, f[:, k], n)
        np.add.at(mag, f[:, k], np.linalg.norm(n, axis=1))
    return np.linalg.norm(acc, axis=1) < rel * np.maximum(mag, 1e-300)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", ctypes.c_float
    if dtype == np.float64:
        return "_f64", ctypes.c_double
    raise TypeError(dtype)


def _flat(x, dtype):
    return np.ascontiguousarray(np.asarray(x, dtype=dtype).reshape(-1))


def camera_position(azim, elev, dist, dtype=np.float32):
    sfx, _ = _sfx(dtype)
    a, e, d = _flat(azim, dtype), _flat(elev, dtype), _flat(dist, dtype)
    n = a.size
    C = np.empty((n, 3), dtype)
    getattr(lib(), "orc_camera_position" + sfx)(ctypes.c_int(n), _ptr(a), _ptr(e), _ptr(d), _ptr(C))
    return C


def look_at(azim, elev, dist, dtype=np.float32):
    """R (n,3,3), T (n,3) of look_at_view_transform(dist, elev, azim) in degrees."""
    sfx, _ = _sfx(dtype)
    a, e, d = _flat(azim, dtype), _flat(elev, dtype), _flat(dist, dtype)
    n = a.size
    R = np.empty((n, 3, 3), dtype)
    T = np.empty((n, 3), dtype)
    getattr(lib(), "orc_look_at" + sfx)(ctypes.c_int(n), _ptr(a), _ptr(e), _ptr(d), _ptr(R), _ptr(T))
    return R, T


def check_rotation(R, tol=1e-6):
    """(all_valid, per-view flags) of util.py:403-420 check_valid_rotation_matrix."""
    R = np.ascontiguousarray(R)
    sfx, cty = _sfx(R.dtype)
    n = R.shape[0]
    flags = np.empty(n, np.int32)
    fn = getattr(lib(), "orc_check_rotation" + sfx)
    fn.restype = ctypes.c_int
    ok = fn(ctypes.c_int(n), _ptr(R), cty(tol), _ptr(flags))
    return bool(ok), flags.astype(bool)


def vertex_normals(verts, faces):
    verts = np.ascontiguousarray(verts)
    sfx, _ = _sfx(verts.dtype)
    faces = np.ascontiguousarray(faces, dtype=np.int64)
    out = np.empty_like(verts)
    getattr(lib(), "orc_vertex_normals" + sfx)(
        ctypes.c_int(verts.shape[0]), ctypes.c_int(faces.shape[0]), _ptr(verts), _ptr(faces), _ptr(out))
    return out


def pack_meshes(verts_list, faces_list, dtype=np.float32):
    vert_off = np.zeros(len(verts_list) + 1, np.int32)
    face_off = np.zeros(len(verts_list) + 1, np.int32)
    for i, (v, f) in enumerate(zip(verts_list, faces_list)):
        vert_off[i + 1] = vert_off[i] + len(v)
        face_off[i + 1] = face_off[i] + len(f)
    verts = np.ascontiguousarray(np.concatenate([np.asarray(v, dtype).reshape(-1, 3) for v in verts_list], 0))
    faces = np.ascontiguousarray(np.concatenate([np.asarray(f, np.int64).reshape(-1, 3) for f in faces_list], 0))
    return verts, faces, vert_off, face_off


def render_meshes(verts_list, faces_list, azim, elev, dist, image_size, light_dir=((0.0, 1.0, 0.0),),
                  obj_color=(1.0, 1.0, 1.0), bg_color=(1.0, 1.0, 1.0), vert_colors=None,
                  perspective_correct=True, cull_backfaces=False, fov=60.0,
                  ambient=0.5, diffuse=0.3, specular=0.2, shininess=64.0,
                  mode=1, fixed_p2f=None, dtype=np.float32, R=None, T=None, return_fragments=True, z_clip=0.0):
    """Restates models/renderer.py:65-114.  azim/elev/dist: (B,V) degrees.  z_clip > 0: PyTorch3D >= 0.5's face clipping
    at that view depth (MeshRasterizer uses znear / 2 = 0.5 for the perspective cameras of the mesh path); 0 = off.
    Returns dict(images (B,V,3,H,W), pix_to_face (N,H,W) local idx, zbuf, bary (N,H,W,3), R, T, straddlers (N,))."""
    sfx, cty = _sfx(dtype)
    azim = np.asarray(azim, dtype)
    B, V = azim.shape
    N = B * V
    H = W = int(image_size)
    verts, faces, vert_off, face_off = pack_meshes(verts_list, faces_list, dtype)
    if R is None:
        R, T = look_at(azim, elev, dist, dtype)
    R = np.ascontiguousarray(R, dtype)
    T = np.ascontiguousarray(T, dtype)
    light = np.ascontiguousarray(np.asarray(light_dir, dtype).reshape(-1, 3))
    assert light.shape[0] in (1, N)
    vc = None if vert_colors is None else np.ascontiguousarray(np.asarray(vert_colors, dtype).reshape(-1, 3))
    oc = _flat(obj_color, dtype)
    bg = _flat(bg_color, dtype)
    images = np.empty((B, V, 3, H, W), dtype)
    p2f = np.empty((N, H, W), np.int32) if return_fragments else None
    zbuf = np.empty((N, H, W), dtype) if return_fragments else None
    bary = np.empty((N, H, W, 3), dtype) if return_fragments else None
    fixed = None if fixed_p2f is None else np.ascontiguousarray(fixed_p2f, np.int32)
    straddlers = np.zeros(N, np.int32)
    getattr(lib(), "orc_render_meshes" + sfx)(
        ctypes.c_int(B), ctypes.c_int(V), ctypes.c_int(H), ctypes.c_int(W),
        _ptr(verts), _ptr(faces), _ptr(vert_off), _ptr(face_off), _ptr(R), _ptr(T),
        _ptr(light), ctypes.c_int(light.shape[0]), _ptr(vc), _ptr(oc), _ptr(bg),
        ctypes.c_int(int(perspective_correct)), ctypes.c_int(int(cull_backfaces)),
        cty(fov), cty(ambient), cty(diffuse), cty(specular), cty(shininess),
        ctypes.c_int(mode), _ptr(fixed), cty(z_clip), _ptr(straddlers), _ptr(images), _ptr(p2f), _ptr(zbuf), _ptr(bary))
    return dict(images=images, pix_to_face=p2f, zbuf=zbuf, bary=bary, R=R, T=T, straddlers=straddlers)


def render_meshes_soft(verts_list, faces_list, azim, elev, dist, image_size, blur_radius, faces_per_pixel, shader="soft_phong",
                       sigma=1e-4, gamma=1e-4, clip_barycentric_coords=True, light_dir=((0.0, 1.0, 0.0),),
                       obj_color=(1.0, 1.0, 1.0), bg_color=(1.0, 1.0, 1.0), vert_colors=None, perspective_correct=True,
                       cull_backfaces=False, fov=60.0, ambient=0.5, diffuse=0.3, specular=0.2, shininess=64.0,
                       fixed_p2f=None, dtype=np.float32, R=None, T=None, z_clip=0.0, fixed_piece=None):
    """Soft rasterisation + SoftPhongShader ("soft_phong") / SoftSilhouetteShader ("soft_silhouette") — SURVEY 8 f4.
    Returns dict(images (B,V,4,H,W) RGBA, pix_to_face / zbuf / dists (N,H,W,K), bary (N,H,W,K,3), R, T)."""
    sfx, cty = _sfx(dtype)
    azim = np.asarray(azim, dtype)
    B, V = azim.shape
    N, K = B * V, int(faces_per_pixel)
    assert 1 <= K <= 64
    H = W = int(image_size)
    verts, faces, vert_off, face_off = pack_meshes(verts_list, faces_list, dtype)
    if R is None:
        R, T = look_at(azim, elev, dist, dtype)
    R, T = np.ascontiguousarray(R, dtype), np.ascontiguousarray(T, dtype)
    light = np.ascontiguousarray(np.asarray(light_dir, dtype).reshape(-1, 3))
    assert light.shape[0] in (1, N)
    vc = None if vert_colors is None else np.ascontiguousarray(np.asarray(vert_colors, dtype).reshape(-1, 3))
    oc, bg = _flat(obj_color, dtype), _flat(bg_color, dtype)
    images = np.empty((B, V, 4, H, W), dtype)
    p2f = np.empty((N, H, W, K), np.int32)
    zbuf, dists, bary = np.empty((N, H, W, K), dtype), np.empty((N, H, W, K), dtype), np.empty((N, H, W, K, 3), dtype)
    fixed = None if fixed_p2f is None else np.ascontiguousarray(fixed_p2f, np.int32)
    # pieces of cut faces (round 4): returned as "pix_to_piece" (-1: whole face); with fixed_p2f, `fixed_piece` (the array a
    # free render returned) makes the listed fragments be evaluated on the same pieces — without it fixed lists evaluate WHOLE faces
    piece_out = np.full((N, H, W, K), -1, np.int32)
    fpiece = None if fixed_piece is None else np.ascontiguousarray(fixed_piece, np.int32)
    lib().orc_set_soft_pieces(_ptr(piece_out), _ptr(fpiece))
    getattr(lib(), "orc_render_meshes_soft" + sfx)(
        ctypes.c_int(B), ctypes.c_int(V), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(K),
        _ptr(verts), _ptr(faces), _ptr(vert_off), _ptr(face_off), _ptr(R), _ptr(T),
        _ptr(light), ctypes.c_int(light.shape[0]), _ptr(vc), _ptr(oc), _ptr(bg),
        ctypes.c_int(int(perspective_correct)), ctypes.c_int(int(clip_barycentric_coords)), ctypes.c_int(int(cull_backfaces)),
        cty(fov), cty(ambient), cty(diffuse), cty(specular), cty(shininess), cty(blur_radius), cty(sigma), cty(gamma),
        ctypes.c_int({"soft_phong": 0, "soft_silhouette": 1}[shader]), _ptr(fixed), cty(z_clip),
        _ptr(images), _ptr(p2f), _ptr(zbuf), _ptr(dists), _ptr(bary))
    lib().orc_set_soft_pieces(None, None)
    return dict(images=images, pix_to_face=p2f, pix_to_piece=piece_out, zbuf=zbuf, dists=dists, bary=bary, R=R, T=T)


def render_points(points, azim, elev, dist, image_size, radius=0.006, obj_color=(1.0, 1.0, 1.0),
                  bg_color=(1.0, 1.0, 1.0), point_colors=None, mode=1, fixed_idx=None,
                  dtype=np.float32, R=None, T=None, return_fragments=True, points_per_pixel=1):
    """Restates models/renderer.py:116-151.  points (B,P,3); azim/elev/dist (B,V).  idx / zbuf / dist2 are
    (N,H,W) for points_per_pixel == 1 and (N,H,W,K) otherwise."""
    sfx, cty = _sfx(dtype)
    azim = np.asarray(azim, dtype)
    B, V = azim.shape
    N = B * V
    H = W = int(image_size)
    pts = np.ascontiguousarray(np.asarray(points, dtype))
    assert pts.shape[0] == B and pts.shape[2] == 3
    P = pts.shape[1]
    if R is None:
        R, T = look_at(azim, elev, dist, dtype)
    R = np.ascontiguousarray(R, dtype)
    T = np.ascontiguousarray(T, dtype)
    d = _flat(dist, dtype)
    assert d.size == N
    pc = None if point_colors is None else np.ascontiguousarray(np.asarray(point_colors, dtype))
    oc = _flat(obj_color, dtype)
    bg = _flat(bg_color, dtype)
    images = np.empty((B, V, 3, H, W), dtype)
    K = int(points_per_pixel)
    shp = (N, H, W) if K == 1 else (N, H, W, K)
    idx = np.empty(shp, np.int32) if return_fragments else None
    zbuf = np.empty(shp, dtype) if return_fragments else None
    d2 = np.empty(shp, dtype) if return_fragments else None
    fixed = None if fixed_idx is None else np.ascontiguousarray(fixed_idx, np.int32)
    getattr(lib(), "orc_render_points" + sfx)(
        ctypes.c_int(B), ctypes.c_int(V), ctypes.c_int(P), ctypes.c_int(H), ctypes.c_int(W), ctypes.c_int(K),
        _ptr(pts), _ptr(R), _ptr(T), _ptr(d), _ptr(pc), _ptr(oc), _ptr(bg),
        ctypes.c_double(radius), ctypes.c_int(mode), _ptr(fixed),
        _ptr(images), _ptr(idx), _ptr(zbuf), _ptr(d2))
    return dict(images=images, idx=idx, zbuf=zbuf, dist2=d2, R=R, T=T)


def norm_weighted_weight_grads(idx, dist2, feats, grad_images, radius):
    """d loss / d w_k of NormWeightedCompositor as PyTorch3D's norm_weighted_sum backward computes it
    (pytorch3d/csrc/compositing/norm_weighted_sum.cu|cpp, un-vendored: restated from the published algorithm):
    (f_k S - sum_j w_j f_j) / S^2 * grad_out, summed over channels, with S = max(sum_j w_j, 1e-4) — i.e. the
    unclamped quotient rule evaluated with the clamped S.  fp64.
      idx, dist2: (N,H,W,K) fragments; feats: (N,H,W,K,3) colour of each slot; grad_images: (N,3,H,W)."""
    idx = np.asarray(idx)
    valid = idx >= 0
    w = np.where(valid, 1.0 - np.asarray(dist2, np.float64) / (float(radius) * float(radius)), 0.0)
    f = np.where(valid[..., None], np.asarray(feats, np.float64), 0.0)
    S = np.maximum(w.sum(-1), 1e-4)                                   # (N,H,W)
    A = (w[..., None] * f).sum(-2)                                     # (N,H,W,3)
    G = np.moveaxis(np.asarray(grad_images, np.float64), 1, -1)        # (N,H,W,3)
    gw = (((f * S[..., None, None] - A[..., None, :]) / (S * S)[..., None, None]) * G[..., None, :]).sum(-1)
    return np.where(valid, gw, 0.0)
