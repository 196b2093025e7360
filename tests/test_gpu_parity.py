"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): pixel -> face / pixel -> point maps bit-exact; images within
1e-4 max abs error (fp32) — observed error is ~1e-6 (pow by squaring vs powf, atomic-order of the
normal accumulation); camera matrices bit-exact; gradients vs fp64 finite differences of the
oracle within 2e-3 relative (fp32 accumulation over thousands of pixels).
"""
import numpy as np
import pytest
import torch

from mvtn_amd import synthetic as syn

pytestmark = pytest.mark.gpu

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


def test_look_at_bit_exact(oracle):
    from mvtn_amd import functional as F
    rng = np.random.default_rng(0)
    az = rng.uniform(-360, 360, 500).astype(np.float32)
    el = rng.uniform(-89, 89, 500).astype(np.float32)
    di = rng.uniform(0.5, 4, 500).astype(np.float32)
    az[:4], el[:4] = [0, 90, 180, 270], [0, 0, 0, 0]
    el[4:8] = [90, -90, 89.9999, -90.0]  # degenerate: invalid R (KA10)
    R, T, status = F.look_at_view_transform(_t(di), _t(el), _t(az), return_status=True)
    Ro, To = oracle.look_at(az, el, di)
    assert np.array_equal(R.cpu().numpy(), Ro)
    assert np.array_equal(T.cpu().numpy(), To)
    ok, flags = oracle.check_rotation(Ro)
    assert np.array_equal(status[2:].cpu().numpy().astype(bool), flags)
    assert int(status[0]) == 0 and int(status[1]) == int((~flags).sum()) and int(status[1]) >= 1
    C = F.camera_position_from_spherical_angles(_t(di), _t(el), _t(az))
    assert np.array_equal(C.cpu().numpy(), oracle.camera_position(az, el, di))
    # a dense sweep: the device evaluates sin and cos of an angle with ONE f64 sincos; it must return what the oracle's
    # separate libm sin / cos return, for every angle (200 000 random ones + every multiple of 0.25 degree)
    az2 = np.concatenate([rng.uniform(-720, 720, 200000), np.arange(-360, 360.25, 0.25)]).astype(np.float32)
    el2 = np.concatenate([rng.uniform(-90, 90, 200000), np.resize(np.arange(-90, 90.25, 0.25), az2.size - 200000)]).astype(np.float32)
    di2 = np.full(az2.size, 2.2, np.float32)
    C2 = F.camera_position_from_spherical_angles(_t(di2), _t(el2), _t(az2))
    assert np.array_equal(C2.cpu().numpy(), oracle.camera_position(az2, el2, di2))


def test_vertex_normals(oracle):
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    vl, fl = _mesh_batch([("S", 0), ("M", 1), ("soup", 2)])
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    got = F.vertex_normals(packed).cpu().numpy()
    got2 = F.vertex_normals(packed).cpu().numpy()
    assert np.array_equal(got, got2), "fixed-point accumulation must be bit-reproducible"
    want = np.concatenate([oracle.vertex_normals(v, f) for v, f in zip(vl, fl)], 0)
    # summation order differs from the sequential index_add (exact fixed-point sum vs float adds);
    # the CAD soup's normals cancel to rounding noise and are excluded (see _well_conditioned_pixels)
    n_ok = len(vl[0]) + len(vl[1])
    assert np.abs(got[:n_ok] - want[:n_ok]).max() < 2e-5
    assert np.isfinite(got).all() and np.abs(np.linalg.norm(got[:n_ok], axis=1) - 1).max() < 1e-5


@pytest.mark.parametrize("size,specs,V,persp,cull,light,strip", [
    (128, [("M", 0)], 4, True, False, "fixed", 0),                                  # C1
    (224, [("M", 0), ("M", 1)], 12, True, False, "fixed", 0),                       # C2 slice
    (224, [("L", 2), ("L", 3)], 20, True, False, "fixed", 0),                       # C5 slice: ShapeNet-sized meshes, 20 views
    (224, [("S", 3), ("L", 4), ("soup", 5), ("M", 6)], 6, True, False, "relative", 0),  # ragged + CAD soup
    (224, [("M", 7), ("soup", 8)], 4, False, True, "fixed", 0),                     # 0.4-style switches
    (96, [("S", 9)], 3, True, False, "fixed", 5),                                   # odd strip height
    (37, [("S", 10), ("soup", 11)], 2, True, True, "relative", 0),                  # odd image size
])
def test_mesh_forward_parity(oracle, size, specs, V, persp, cull, light, strip):
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    vl, fl = _mesh_batch(specs)
    B = len(vl)
    az, el, di = _views(B, V, seed=len(specs) + V) if V in (4, 6, 12, 20) else _views(B, 4, 1)
    az, el, di = az[:, :V], el[:, :V], di[:, :V]
    if size == 128:  # C1: the reference's own circular configuration
        a, e, d = syn.circular_views(4, 30.0, 2.2)
        az, el, di = a[None], e[None], d[None]
    obj, bg = (0.99999, 0.99999, 0.99999), (0.99999, 0.99999, 0.99999)
    ldir = oracle.camera_position(az, el, di) if light == "relative" else np.array([[0.0, 1.0, 0.0]], np.float32)
    want = oracle.render_meshes(vl, fl, az, el, di, size, light_dir=ldir, obj_color=obj, bg_color=bg,
                                perspective_correct=persp, cull_backfaces=cull, mode=1)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.mesh_params(V, size, ldir.shape[0], obj, bg, cull_backfaces=cull, perspective_correct=persp, strip_rows=strip)
    img, p2f = F.render_meshes_fragments(packed, R, T, _t(ldir), params)
    torch.cuda.synchronize()
    p2f, img = p2f.cpu().numpy(), img.cpu().numpy()
    n_diff = int((p2f != want["pix_to_face"]).sum())
    well = _well_conditioned_pixels(vl, fl, want["pix_to_face"], V)
    diff = np.abs(img - want["images"]).reshape(B * V, 3, size, size).max(1)
    err = float(diff[well].max())
    print("coverage mismatches: %d  max|d|: %.3g  covered: %.3f  ill-conditioned px: %d" % (
        n_diff, err, (p2f >= 0).mean(), int((~well).sum())))
    assert n_diff == 0, "pixel->face map differs from the oracle at %d pixels" % n_diff
    assert err <= TOL_PIXEL
    assert img.shape == (B, V, 3, size, size) and np.isfinite(img).all()


def test_mesh_custom_vertex_colors(oracle):
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    vl, fl = _mesh_batch([("S", 0), ("M", 1)])
    rng = np.random.default_rng(0)
    mx = max(len(v) for v in vl)
    col = rng.uniform(0, 1, (2, mx, 3)).astype(np.float32)
    vc = np.concatenate([col[b, :len(v)] for b, v in enumerate(vl)], 0)
    az, el, di = _views(2, 4, 3)
    want = oracle.render_meshes(vl, fl, az, el, di, 64, vert_colors=vc, bg_color=(0, 0, 0))
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev(), vert_colors=torch.from_numpy(col))
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.mesh_params(4, 64, 1, (1, 1, 1), (0, 0, 0))
    img, p2f = F.render_meshes_fragments(packed, R, T, _t([[0, 1, 0]]), params)
    assert np.array_equal(p2f.cpu().numpy(), want["pix_to_face"])
    assert np.abs(img.cpu().numpy() - want["images"]).max() <= TOL_PIXEL


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


@pytest.mark.parametrize("light,persp", [("relative", True), ("fixed", True), ("relative", False)])
def test_mvrenderer_mesh_forward_backward(oracle, light, persp):
    """End-to-end through the nn.Module + autograd, against fp64 finite differences (KA9)."""
    from mvtn_amd import MVRenderer, Meshes
    vl, fl = _mesh_batch([("S", 0), ("M", 1)])
    B, V, size = 2, 3, 64
    az, el, di = _views(B, 4, 5)
    az, el, di = az[:, :V], el[:, :V], di[:, :V]
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white",
                   light_direction=light, perspective_correct=persp).eval()
    meshes = [Meshes([v], [f]) for v, f in zip(vl, fl)]
    ta, te, td = (_t(x).requires_grad_(True) for x in (az, el, di))
    images, cams = r(meshes, None, ta, te, td)
    assert images.shape == (B, V, 3, size, size) and cams.R.shape == (B * V, 3, 3)
    ldir = oracle.camera_position(az, el, di) if light == "relative" else np.array([[0.0, 1.0, 0.0]], np.float32)
    want = oracle.render_meshes(vl, fl, az, el, di, size, light_dir=ldir, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3,
                                perspective_correct=persp)
    assert np.abs(images.detach().cpu().numpy() - want["images"]).max() <= TOL_PIXEL
    rng = np.random.default_rng(2)
    G = rng.normal(size=images.shape)
    images.backward(_t(G))
    fd = _fd_grad_full(oracle, vl, fl, az, el, di, size, G, want["pix_to_face"], light == "relative", persp)
    for name, got, ref in zip(("azim", "elev", "dist"), (ta.grad, te.grad, td.grad), fd):
        got = got.cpu().numpy().astype(np.float64)
        scale = np.abs(ref).max() + 1e-12
        rel = np.abs(got - ref).max() / scale
        print(name, "rel err", rel, "scale", scale)
        assert rel < 2e-3, (name, got, ref)


def test_host_mesh_hand_over_staging_ring():
    """CPU meshes every forward (the reference's calling convention, models/renderer.py:66-77): the pinned staging
    blocks are reused round-robin, so queue many forwards with DIFFERENT batches back to back, with no host sync in
    between, and compare each with the render of the same batch from HBM-resident PackedMeshes."""
    from mvtn_amd import MVRenderer, Meshes
    from mvtn_amd.structures import PackedMeshes
    V, size = 4, 64
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="fixed").eval()
    batches = []
    for k in range(7):                       # more batches than ring slots, ragged and of different sizes
        specs = [("S", k), ("M", 10 + k)] if k % 2 else [("soup", k), ("S", 20 + k), ("S", 30 + k)]
        vl, fl = _mesh_batch(specs)
        batches.append([Meshes([v], [f]) for v, f in zip(vl, fl)])
    views = [tuple(_t(x[:, :V]) for x in _views(len(b), 4, 40 + k)) for k, b in enumerate(batches)]
    with torch.no_grad():
        want = [r(PackedMeshes.from_meshes(b, _dev()), None, *v)[0].clone() for b, v in zip(batches, views)]
        torch.cuda.synchronize()
        got = []
        for rep in range(3):
            for b, v in zip(batches, views):
                got.append(r(b, None, *v)[0])
        torch.cuda.synchronize()
    for i, g in enumerate(got):
        assert torch.equal(g, want[i % len(batches)]), i


def test_mesh_backward_is_bit_reproducible():
    """The covered-pixel list of k_mesh_backward has a fixed order, every reduction a fixed tree: gradients repeat
    bit for bit (PyTorch3D's backward accumulates with float atomics and does not)."""
    from mvtn_amd import MVRenderer, Meshes
    vl, fl = _mesh_batch([("M", 0), ("soup", 1), ("L", 2)])
    B, V, size = 3, 6, 224
    az, el, di = _views(B, 6, 3)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative").eval()
    meshes = [Meshes([v], [f]) for v, f in zip(vl, fl)]
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(4)).to(_dev())
    runs = []
    for _ in range(4):
        leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
        r(meshes, None, *leaves)[0].backward(G)
        runs.append([t.grad.clone() for t in leaves])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)
    assert all(t.abs().max() > 0 for t in runs[0])


@pytest.mark.parametrize("dist,gather", [(1.1, False), (1.1, True), (0.9, False), (0.9, True)])
def test_mesh_backward_through_cut_faces_against_finite_differences(oracle, dist, gather):
    """Gradients where the clip plane cuts faces (dist 1.1: vertices between the camera plane and z = 0.5; dist 0.9: the
    camera is inside the bounding sphere, vertices behind it): fp64 central differences of the CLIPPING oracle with the
    pixel -> face map held fixed, both backward kernels."""
    from mvtn_amd import Meshes, PackedMeshes, functional as F
    vl, fl = _mesh_batch([("M", 31)])
    B, V, size = 1, 3, 64
    az, el, _ = _views(B, 4, 33)
    az, el = az[:, :V], el[:, :V]
    di = np.full((B, V), dist, np.float32)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    obj = bg = (0.99999,) * 3
    params = F.mesh_params(V, size, B * V, obj, bg, gather_backward=gather)   # z_clip 0.5
    leaves = [_t(x.reshape(-1)).requires_grad_(True) for x in (az, el, di)]
    R, T = F.look_at_view_transform(leaves[2], leaves[1], leaves[0])
    light = F.camera_position_from_spherical_angles(leaves[2], leaves[1], leaves[0])
    images = F.render_meshes(packed, R, T, light, params)
    want = oracle.render_meshes(vl, fl, az, el, di, size, light_dir=oracle.camera_position(az, el, di), obj_color=obj, bg_color=bg, z_clip=0.5)
    assert int(want["straddlers"].sum()) > 0
    assert np.abs(images.detach().cpu().numpy() - want["images"]).max() <= TOL_PIXEL
    G = np.random.default_rng(34).normal(size=images.shape)
    images.backward(_t(G))
    fd = _fd_grad_full(oracle, vl, fl, az, el, di, size, G, want["pix_to_face"], True, True, z_clip=0.5)
    for name, leaf, ref in zip(("azim", "elev", "dist"), leaves, fd):
        got = leaf.grad.cpu().numpy().astype(np.float64).reshape(ref.shape)
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert rel < 2e-3, (name, rel, got, ref)


@pytest.mark.parametrize("exact", [False, True])
def test_normalize_fused_into_the_rasteriser_stores(exact):
    """MVRenderer(normalize=(mean, std)) / mesh_params(normalize=...): the trainer's Normalize
    (viewGCN/tools/Trainer_mvt.py:41-49) applied where the rasteriser stores its pixels — against the plain render followed
    by torch's (x - mean) / std, images (background included) and gradients, through both rasterisers and, for the module,
    the fused look-at node."""
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes, functional as F
    from mvtn_amd.regularize import IMAGENET_MEAN, IMAGENET_STD
    vl, fl = _mesh_batch([("M", 41), ("soup", 42), ("S", 43)])
    B, V, size = 3, 4, 96
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    az, el, di = _views(B, 4, 44)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    mt, st = _t(np.array(mean)).view(1, 1, 3, 1, 1), _t(np.array(std)).view(1, 1, 3, 1, 1)
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(45)).to(_dev())
    out = {}
    for fused in (False, True):
        params = F.mesh_params(V, size, B * V, (0.9, 0.7, 0.5), (0.2, 0.4, 0.6), exact_raster=exact, normalize=(mean, std) if fused else None)
        R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
        R, T = R.detach().requires_grad_(True), T.detach().requires_grad_(True)
        light = F.camera_position_from_spherical_angles(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1))).detach().requires_grad_(True)
        img = F.render_meshes(packed, R, T, light, params)
        if not fused:
            img = (img - mt) / st
        img.backward(G)
        out[fused] = (img.detach(), [x.grad.double() for x in (R, T, light)])
    assert (out[True][0] - out[False][0]).abs().max() <= 2e-6 / min(std)
    for a, b in zip(out[True][1], out[False][1]):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-9
    if not exact:   # the module: one autograd node, ImageNet constants of the reference's trainer
        r0 = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative").eval()
        r1 = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative",
                        normalize=(IMAGENET_MEAN, IMAGENET_STD)).eval()
        res = []
        for r in (r0, r1):
            leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
            img, _ = r(packed, None, *leaves)
            if r is r0:
                img = (img - IMAGENET_MEAN[0]) / IMAGENET_STD[0]
            img.backward(G)
            res.append((img.detach(), [t.grad.double() for t in leaves]))
        assert (res[1][0] - res[0][0]).abs().max() <= 1e-5
        for a, b in zip(res[1][1], res[0][1]):
            assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-9


@pytest.mark.parametrize("exact", [False, True])
def test_regulariser_window_rendered_by_the_rasteriser(exact):
    """MVRenderer(...)(..., window=draws) / functional.with_window: the reference's view regulariser (ops.py:138-176: view
    dropout, flip, replicate-pad 67, random 224-crop) as the rasteriser's grid of sample positions — against the plain render
    followed by `apply_regularization` with the same draws (itself bit-exact to the reference's composition): identical
    frames from the exact kernel (the same arithmetic at the same sample positions; an ulp apart from the fast kernel, in the
    strips it hands to the exact one), identical pixel -> face maps, gradients to round-off; with
    the Normalize in the stores on top; through both rasterisers; objects reaching the image border (close camera), so that
    the replicated edge rows / columns are not all background."""
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes, functional as F, regularize as rg
    vl, fl = _mesh_batch([("M", 51), ("S", 52), ("soup", 53)])
    B, V, size = 3, 4, 96
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    az, el, _ = _views(B, 4, 54)
    di = np.full((B, V), 1.6, np.float32)   # fills the frame: the border pixels are covered in several views
    torch.manual_seed(55)
    draws = rg.draw_regularization(B, V, size, _dev(), dropout_p=0.3, augment_training=True, crop_ratio=0.3)
    scale, flip, ci, cj, pad = draws
    assert pad == int(1.3 * size) - size and int(flip.sum()) not in (0, B * V) and float(scale.min()) == 0.0
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(56)).to(_dev())
    obj, bg = (0.9, 0.7, 0.5), (0.2, 0.4, 0.6)
    light = _t(np.array([[0.0, 1.0, 0.0]], np.float32))
    res = {}
    for fused in (False, True):
        params = F.mesh_params(V, size, 1, obj, bg, exact_raster=exact)
        if fused:
            params = F.with_window(params, *draws)
        R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
        R, T = R.detach().requires_grad_(True), T.detach().requires_grad_(True)
        img = F.render_meshes(packed, R, T, light, params)
        if not fused:
            img = rg.apply_regularization(img, *draws)
        img.backward(G)
        _, p2f = F.render_meshes_fragments(packed, R.detach(), T.detach(), light, params)
        res[fused] = (img.detach(), p2f, R.grad.double(), T.grad.double())
    if exact:
        assert torch.equal(res[True][0], res[False][0])
    else:
        # the fast rasteriser hands undecidable strips (the soup's coincident faces) to the exact kernel, whose shading rounds
        # differently (test_fast_rasteriser_equals_exact_kernel); WHICH output strips those are moves with the window
        assert (res[True][0] - res[False][0]).abs().max() <= 5e-7
    # the windowed pixel -> face map is the plain one seen through the window
    n_idx = torch.arange(B * V, device=_dev()).view(-1, 1, 1)
    rows = (torch.arange(size, device=_dev()).view(1, -1, 1) + (ci.view(-1, 1, 1) - pad)).clamp(0, size - 1)
    cols = (torch.arange(size, device=_dev()).view(1, 1, -1) + (cj.view(-1, 1, 1) - pad)).clamp(0, size - 1)
    cols = torch.where(flip.view(-1, 1, 1) != 0, size - 1 - cols, cols)
    assert torch.equal(res[True][1], res[False][1][n_idx, rows, cols])
    border = res[False][1][:, [0, -1], :]
    assert int((border >= 0).sum()) > 0, "the test wants covered border pixels"
    for a, b in ((res[True][2], res[False][2]), (res[True][3], res[False][3])):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-9
    if not exact:   # the module, with dropout + augmentation + Normalize in one render
        r0 = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative").train()
        r1 = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative",
                        normalize=(rg.IMAGENET_MEAN, rg.IMAGENET_STD)).train()
        out = []
        for r in (r0, r1):
            leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
            if r is r0:
                img = rg.apply_regularization(r(packed, None, *leaves)[0], *draws, mean=rg.IMAGENET_MEAN, std=rg.IMAGENET_STD)
            else:
                img = r(packed, None, *leaves, window=draws)[0]
            img.backward(G)
            out.append((img.detach(), [t.grad.double() for t in leaves]))
        assert (out[1][0] - out[0][0]).abs().max() <= 1e-5
        for a, b in zip(out[1][1], out[0][1]):
            assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-9


@pytest.mark.parametrize("K", [1, 3])
def test_normalize_fused_into_the_point_rasteriser_stores(K):
    """The same for the point path (K = 1: k_points_raster_shade, K = 3: the top-K kernel): MVRenderer(pc_rendering=True,
    normalize=...) against the plain render followed by (x - mean) / std; images and the gradient w.r.t. the distance (the
    only camera gradient of the point path that is not identically zero at K = 1 is in the clamp region; K = 3 has all)."""
    from mvtn_amd import MVRenderer
    B, P, V, size = 3, 1500, 4, 96
    rng = np.random.default_rng(46)
    pts = _t(rng.normal(size=(B, P, 3)) * 0.3)
    cols = _t(rng.uniform(0, 1, (B, P, 3)))
    az, el, _ = _views(B, 4, 47)
    di = np.full((B, V), 1.0, np.float32)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    mt, st = _t(np.array(mean)).view(1, 1, 3, 1, 1), _t(np.array(std)).view(1, 1, 3, 1, 1)
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(48)).to(_dev())
    kw = dict(pc_rendering=True, object_color="custom", background_color="black", points_radius=0.02, points_per_pixel=K)
    res = []
    for fused in (False, True):
        r = MVRenderer(V, size, normalize=(mean, std) if fused else None, **kw).eval()
        leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
        img, _ = r(None, pts, *leaves, color=cols)
        if not fused:
            img = (img - mt) / st
        img.backward(G)
        res.append((img.detach(), [t.grad.double() for t in leaves]))
    assert (res[1][0] - res[0][0]).abs().max() <= 2e-6 / min(std)
    for a, b in zip(res[1][1], res[0][1]):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-7
    if K > 1:
        assert all(g.abs().max() > 0 for g in res[0][1])


@pytest.mark.parametrize("K", [1, 3])
def test_regulariser_window_rendered_by_the_point_rasterisers(K):
    """The point path's version of test_regulariser_window_rendered_by_the_rasteriser (K = 1: k_points_raster_shade_win,
    K = 3: k_points_raster_topk_win): MVRenderer(pc_rendering=True)(..., window=draws) against the plain render followed by
    `apply_regularization` with the same draws — identical frames (same arithmetic at the same sample positions), the
    pixel -> point map seen through the window, gradients to round-off; a close camera, so that discs reach the border and
    the replicated rows / columns are not all background; with the Normalize in the stores on top."""
    from mvtn_amd import MVRenderer, functional as F, regularize as rg
    B, P, V, size = 3, 1500, 4, 96
    rng = np.random.default_rng(61)
    pts = _t(rng.normal(size=(B, P, 3)) * 0.45)
    cols = _t(rng.uniform(0, 1, (B, P, 3)))
    az, el, _ = _views(B, 4, 62)
    di = np.full((B, V), 1.0, np.float32)
    torch.manual_seed(63)
    draws = rg.draw_regularization(B, V, size, _dev(), dropout_p=0.3, augment_training=True, crop_ratio=0.3)
    scale, flip, ci, cj, pad = draws
    assert int(flip.sum()) not in (0, B * V) and float(scale.min()) == 0.0
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(64)).to(_dev())
    # the library calls: frames and pixel -> point maps
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    res = {}
    for fused in (False, True):
        params = F.points_params(B, P, V, size, 0.02, (1.0, 1.0, 1.0), (0.1, 0.2, 0.3), points_per_pixel=K)
        if fused:
            params = F.with_window(params, *draws)
        Rg, Tg, dg = R.detach().clone().requires_grad_(True), T.detach().clone().requires_grad_(True), _t(di.reshape(-1)).requires_grad_(True)
        img = F.render_points(pts, cols, Rg, Tg, dg, params)
        p2p = img.grad_fn.saved_tensors[4].clone()
        if not fused:
            img = rg.apply_regularization(img, *draws)
        img.backward(G)
        res[fused] = (img.detach(), p2p, [x.grad.double() for x in (Rg, Tg, dg)])
    assert torch.equal(res[True][0], res[False][0])
    n_idx = torch.arange(B * V, device=_dev()).view(-1, 1, 1)
    rows = (torch.arange(size, device=_dev()).view(1, -1, 1) + (ci.view(-1, 1, 1) - pad)).clamp(0, size - 1)
    cols_i = (torch.arange(size, device=_dev()).view(1, 1, -1) + (cj.view(-1, 1, 1) - pad)).clamp(0, size - 1)
    cols_i = torch.where(flip.view(-1, 1, 1) != 0, size - 1 - cols_i, cols_i)
    assert torch.equal(res[True][1], res[False][1][n_idx, rows, cols_i])
    plain_map = res[False][1] if K == 1 else res[False][1][..., 0]
    assert int((plain_map[:, [0, -1], :] >= 0).sum()) > 0 and int((plain_map[:, :, [0, -1]] >= 0).sum()) > 0, "the test wants covered border pixels"
    for a, b in zip(res[True][2], res[False][2]):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-7
    if K > 1:
        assert all(g.abs().max() > 0 for g in res[False][2])
    # the module, with the Normalize in the stores
    kw = dict(pc_rendering=True, object_color="custom", background_color="black", points_radius=0.02, points_per_pixel=K)
    out = []
    for fused in (False, True):
        r = MVRenderer(V, size, normalize=(rg.IMAGENET_MEAN, rg.IMAGENET_STD) if fused else None, **kw).train()
        leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
        if fused:
            img = r(None, pts, *leaves, color=cols, window=draws)[0]
        else:
            img = rg.apply_regularization(r(None, pts, *leaves, color=cols)[0], *draws, mean=rg.IMAGENET_MEAN, std=rg.IMAGENET_STD)
        img.backward(G)
        out.append((img.detach(), [t.grad.double() for t in leaves]))
    assert (out[1][0] - out[0][0]).abs().max() <= 1e-5
    for a, b in zip(out[1][1], out[0][1]):
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-7


def test_mesh_backward_at_the_c3_image_size_against_finite_differences(oracle):
    """C3's image size and mesh tier (224 px, 2 562 vertices / 5 120 faces: ~1 000 covered pixels and ~100 winning faces per
    8-row strip, the regime k_mesh_backward_tab is built for) against fp64 central differences of the oracle with the
    pixel -> face map held fixed; relative light, so the camera-position branch of the fused finalize runs too."""
    from mvtn_amd import MVRenderer, Meshes
    vl, fl = _mesh_batch([("M", 21), ("M", 22)])
    B, V, size = 2, 3, 224
    az, el, di = _views(B, 12, 23)
    az, el, di = az[:, 2:2 + V], el[:, 2:2 + V], di[:, 2:2 + V]
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative").eval()
    leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
    images, _ = r([Meshes([v], [f]) for v, f in zip(vl, fl)], None, *leaves)
    want = oracle.render_meshes(vl, fl, az, el, di, size, light_dir=oracle.camera_position(az, el, di), obj_color=(0.99999,) * 3,
                                bg_color=(0.99999,) * 3)
    assert np.abs(images.detach().cpu().numpy() - want["images"]).max() <= TOL_PIXEL
    G = np.random.default_rng(24).normal(size=images.shape)
    images.backward(_t(G))
    fd = _fd_grad_full(oracle, vl, fl, az, el, di, size, G, want["pix_to_face"], True, True)
    for name, leaf, ref in zip(("azim", "elev", "dist"), leaves, fd):
        got = leaf.grad.cpu().numpy().astype(np.float64)
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert rel < 2e-3, (name, rel, got, ref)


@pytest.mark.parametrize("persp", [True, False])
def test_mesh_backward_table_kernel_equals_gather_kernel(persp):
    """k_mesh_backward_tab (per-face data fetched once per strip into an LDS table; the default) against k_mesh_backward
    (MVR_BACKWARD_GATHER: per-pixel gathers): the same per-pixel arithmetic summed in a different order — gradients
    agree to float round-off.  Tier M (a single table pass per strip), tier L (~1 pixel per face: several passes, the
    per-wave pixel queue) and a soup of large faces; per-view light; 224 px and 100 px (a last strip of 4 rows)."""
    from mvtn_amd import Meshes, PackedMeshes, functional as F
    vl, fl = _mesh_batch([("M", 5), ("L", 6), ("soup", 7), ("S", 8)])
    B, V = 4, 5
    packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), _dev())
    az, el, di = _views(B, V, 9)
    for size in (224, 100):
        H = W = size
        G = torch.randn(B, V, 3, H, W, generator=torch.Generator().manual_seed(5)).to(_dev())
        grads = {}
        for mode in ("table", "gather"):
            params = F.mesh_params(V, size, B * V, (0.9, 0.8, 0.7), (1, 1, 1), perspective_correct=persp, gather_backward=(mode == "gather"))
            R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
            R, T = R.detach().requires_grad_(True), T.detach().requires_grad_(True)
            light = F.camera_position_from_spherical_angles(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1))).detach().requires_grad_(True)
            img = F.render_meshes(packed, R, T, light, params)
            img.backward(G)
            grads[mode] = [x.grad.double().cpu().numpy() for x in (R, T, light)]
        for name, a, b in zip(("R", "T", "light"), grads["table"], grads["gather"]):
            scale = np.abs(b).max() + 1e-12
            assert np.abs(a - b).max() / scale < 2e-5, (size, name, np.abs(a - b).max() / scale)
            assert scale > 1e-3


def test_refined_reciprocal_is_correctly_rounded_exhaustive():
    """The rasteriser's perspective-correct barycentrics divide by den through y = RN(1/den) (Markstein); y comes
    from v_rcp_f32 + two Newton steps.  Check y == 1.0f / den (IEEE) for EVERY float in [2^-30, 2^24): all 2^23
    mantissas of 54 binades — den = max(., 1e-8) is >= 2^-27 and a sum of three z_i z_j-scaled barycentrics."""
    from mvtn_amd import _lib
    lib = _lib.load()
    out = torch.zeros(2, dtype=torch.int32, device=_dev())
    lo = np.array([2.0 ** -30], np.float32).view(np.uint32)[0]
    hi = np.array([2.0 ** 24], np.float32).view(np.uint32)[0] - 1
    _lib.check(lib.mvr_selftest_rcp(int(lo), int(hi), out[0:1].data_ptr(), out[1:2].data_ptr(), None), "mvr_selftest_rcp")
    torch.cuda.synchronize()
    n_bad, first = int(out[0]), int(out[1]) & 0xFFFFFFFF
    assert n_bad == 0, "%d mismatches, e.g. bits 0x%08x" % (n_bad, first)


def test_fused_mesh_views_node_equals_chained_nodes():
    """MVRenderer's single autograd node (look-at -> relative light -> render) gives the same images and gradients
    as the three chained nodes, including a loss term that differentiates through the returned cameras."""
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes, functional as F
    B, V, size = 2, 3, 64
    vl, fl = _mesh_batch([("S", 0), ("M", 1)])
    packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), _dev())
    rng = np.random.default_rng(4)
    az = rng.uniform(-180, 180, (B, V)).astype(np.float32)
    el = rng.uniform(-60, 60, (B, V)).astype(np.float32)
    di = rng.uniform(1.8, 2.6, (B, V)).astype(np.float32)
    G = _t(rng.normal(size=(B, V, 3, size, size)))
    GR, GT = _t(rng.normal(size=(B * V, 3, 3))), _t(rng.normal(size=(B * V, 3)))
    rend = MVRenderer(V, size, pc_rendering=False, light_direction="relative", rotation_check="off").eval()
    a1, e1, d1 = (_t(x).requires_grad_(True) for x in (az, el, di))
    img1, cams = rend(packed, None, a1, e1, d1)
    ((img1 * G).sum() + (cams.R * GR).sum() + (cams.T * GT).sum()).backward()
    a2, e2, d2 = (_t(x).requires_grad_(True) for x in (az, el, di))
    R, T = F.look_at_view_transform(d2.reshape(-1), e2.reshape(-1), a2.reshape(-1))
    light = F.camera_position_from_spherical_angles(d2.reshape(-1), e2.reshape(-1), a2.reshape(-1))
    from mvtn_amd.helpers import torch_color
    params = F.mesh_params(V, size, B * V, [float(c) for c in rend.rendering_color()],
                           [float(c) for c in torch_color(rend.background_color, max_lightness=True)])
    img2 = F.render_meshes(packed, R, T, light, params)
    ((img2 * G).sum() + (R * GR).sum() + (T * GT).sum()).backward()
    assert torch.equal(img1, img2)
    for x, y in ((a1, a2), (e1, e2), (d1, d2)):
        assert torch.allclose(x.grad, y.grad, rtol=1e-5, atol=1e-6), (x.grad, y.grad)
    # forward-only (no graph) renders the same thing
    with torch.no_grad():
        img3, _ = rend(packed, None, a1, e1, d1)
    assert torch.equal(img3, img1)


@pytest.mark.parametrize("size,B,P,V,dist,strip", [(224, 3, 2048, 20, 1.0, 0), (224, 2, 2048, 4, 2.2, 0), (128, 2, 512, 4, 1.0, 7),
                                                   (45, 1, 300, 3, 0.8, 0)])
def test_points_forward_parity(oracle, size, B, P, V, dist, strip):
    from mvtn_amd import functional as F
    vl, fl = _mesh_batch([("M", s) for s in range(B)])
    pts = np.stack([syn.surface_points(v, f, P, seed=i) for i, (v, f) in enumerate(zip(vl, fl))])
    az, el, di = syn.spherical_views(V) if V in (4, 20) else syn.circular_views(V)
    az, el = np.tile(az, (B, 1)), np.tile(el, (B, 1))
    di = np.full((B, V), dist, np.float32)
    obj, bg = (0.99999,) * 3, (0.99999, 0.5, 0.25)
    want = oracle.render_points(pts, az, el, di, size, radius=0.006 if size >= 128 else 0.03, obj_color=obj, bg_color=bg)
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.points_params(B, P, V, size, 0.006 if size >= 128 else 0.03, obj, bg, strip_rows=strip)
    img, idx = F.render_points_fragments(_t(pts), None, R, T, _t(di.reshape(-1)), params)
    idx, img = idx.cpu().numpy(), img.cpu().numpy()
    n_diff = int((idx != want["idx"]).sum())
    err = float(np.abs(img - want["images"]).max())
    print("idx mismatches %d, max|d| %.3g, covered %.4f" % (n_diff, err, (idx >= 0).mean()))
    assert n_diff == 0 and err <= 1e-6


def test_points_colors_and_module(oracle):
    from mvtn_amd import MVRenderer
    B, P, V, size = 2, 1024, 4, 128
    vl, fl = _mesh_batch([("M", 0), ("M", 1)])
    pts = np.stack([syn.surface_points(v, f, P, seed=i) for i, (v, f) in enumerate(zip(vl, fl))])
    az, el, di = syn.spherical_views(V)
    az, el, di = np.tile(az, (B, 1)), np.tile(el, (B, 1)), np.full((B, V), 1.0, np.float32)
    r = MVRenderer(V, size, pc_rendering=True, object_color="red", background_color="black", points_radius=0.02).eval()
    ta, te, td = (_t(x).requires_grad_(True) for x in (az, el, di))
    images, cams = r(None, torch.from_numpy(pts), ta, te, td)
    want = oracle.render_points(pts, az, el, di, size, radius=0.02, obj_color=(1.0 / 1.00001, 0, 0), bg_color=(0, 0, 0))
    assert np.abs(images.detach().cpu().numpy() - want["images"]).max() <= 1e-6
    # KA8: K=1 norm-weighted compositing has exactly zero gradient away from the 1e-4 weight clamp
    w = 1 - want["dist2"] / np.float32(0.02 * 0.02)
    clamped = (want["idx"] >= 0) & (w < 1e-4)
    G = np.ones(images.shape, np.float32).reshape(B * V, 3, size, size)
    G[np.broadcast_to(clamped[:, None], G.shape)] = 0
    images.backward(_t(G.reshape(images.shape)))
    assert float(ta.grad.abs().max()) == 0 and float(te.grad.abs().max()) == 0 and float(td.grad.abs().max()) == 0


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


def test_points_backward_clamp_region(oracle):
    """A point whose weight falls below the compositor's 1e-4 clamp has a non-zero gradient; it must be the one
    PyTorch3D's backward returns, f (1e-4 - w) / 1e-8 per unit of d w, not the derivative of the clamped function."""
    from mvtn_amd import functional as F
    size, r = 32, 0.05
    az, el, di = np.array([[20.0]], np.float32), np.array([[10.0]], np.float32), np.array([[1.3]], np.float32)
    R64, T64 = oracle.look_at(az, el, di, np.float64)
    # want view-space (x, y) = pixel centre (i=10, j=12) + offset of length sqrt(r^2 (1 - 5e-5)) ; world = ((view - T) R^T) * dist
    cx, cy = -1 + (2 * 10 + 1) / size, -1 + (2 * 12 + 1) / size
    off = np.sqrt(r * r * (1 - 5e-5))
    view = np.array([cx + off * 0.6, cy + off * 0.8, 1.1])
    world = ((view - T64[0]) @ R64[0].T) * float(di[0, 0])
    pts = np.array([[world, [5.0, 5.0, 5.0]]], np.float32)  # second point far outside the frustum
    obj = (0.7, 0.8, 0.9)
    want = oracle.render_points(pts, az, el, di, size, radius=r, obj_color=obj, bg_color=(0, 0, 0))
    hit = np.argwhere(want["idx"][0] == 0)
    w = 1 - want["dist2"][0][want["idx"][0] == 0] / (r * r)
    assert len(hit) >= 1 and (w < 1e-4).any(), "test construction: expected a clamped pixel"
    ta, te, td = (_t(x.reshape(-1)).requires_grad_(True) for x in (az, el, di))
    Rt, Tt = F.look_at_view_transform(td, te, ta)
    params = F.points_params(1, 2, 1, size, r, obj, (0, 0, 0))
    img = F.render_points(_t(pts), None, Rt, Tt, td, params)
    assert np.abs(img.detach().cpu().numpy() - want["images"]).max() <= 1e-6
    rng = np.random.default_rng(0)
    G = rng.normal(size=img.shape)
    img.backward(_t(G))
    ref = _points_grad_reference(oracle, pts, None, obj, az, el, di, size, r, G, want["idx"], want["dist2"])
    for name, got, rf in zip(("azim", "elev", "dist"), (ta.grad, te.grad, td.grad), ref):
        got = float(got.cpu().numpy().reshape(-1)[0])
        print(name, got, rf[0, 0])
        # w is an fp32 cancellation (1 - d2/r2 with d2/r2 within 5e-5 of 1): 1e-4 - w carries ~1e-3 relative error
        assert abs(got - rf[0, 0]) <= 2e-2 * abs(rf[0, 0]) + 1e-3


@pytest.mark.parametrize("size,B,P,V,K,r,strip,colored", [(64, 2, 1500, 3, 4, 0.05, 0, True), (224, 2, 2048, 4, 2, 0.006, 0, False),
                                                          (96, 1, 4000, 2, 8, 0.04, 5, True), (45, 1, 300, 2, 3, 0.03, 0, True)])
def test_points_topk_forward_parity(oracle, size, B, P, V, K, r, strip, colored):
    """points_per_pixel = K > 1: the (N,H,W,K) slot lists are bit-identical to the oracle's, images within 1e-6."""
    from mvtn_amd import functional as F
    vl, fl = _mesh_batch([("M", s) for s in range(B)])
    pts = np.stack([syn.surface_points(v, f, P, seed=i) for i, (v, f) in enumerate(zip(vl, fl))])
    az, el, di = syn.spherical_views(V) if V == 4 else syn.circular_views(V)
    az, el, di = np.tile(az, (B, 1)), np.tile(el, (B, 1)), np.full((B, V), 1.0, np.float32)
    col = np.random.default_rng(3).uniform(0, 1, (B, P, 3)).astype(np.float32) if colored else None
    obj, bg = (0.9, 0.6, 0.3), (0.1, 0.2, 0.3)
    want = oracle.render_points(pts, az, el, di, size, radius=r, obj_color=obj, bg_color=bg, point_colors=col, points_per_pixel=K)
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.points_params(B, P, V, size, r, obj, bg, strip_rows=strip, points_per_pixel=K)
    img, idx = F.render_points_fragments(_t(pts), None if col is None else _t(col), R, T, _t(di.reshape(-1)), params)
    idx, img = idx.cpu().numpy(), img.cpu().numpy()
    assert idx.shape == want["idx"].shape
    n_diff = int((idx != want["idx"]).sum())
    err = float(np.abs(img - want["images"]).max())
    depth = [(want["idx"][..., k] >= 0).mean() for k in range(K)]
    print("slot mismatches %d, max|d| %.3g, slot fill %s" % (n_diff, err, np.round(depth, 4)))
    assert n_diff == 0 and err <= 1e-6
    if K > 2 and r > 0.035:
        assert depth[2] > 0.01, "test construction: expected real depth complexity"
    # the no-fragments call (slot lists in the workspace) renders the same image
    img2 = F.render_points(_t(pts), None if col is None else _t(col), R, T, _t(di.reshape(-1)), params).cpu().numpy()
    assert np.array_equal(img2, img)


def test_points_topk_backward_and_module(oracle):
    """MVRenderer(points_per_pixel=3): image parity and camera gradients vs PyTorch3D's compositor backward chained
    through fp64 finite differences of the oracle."""
    from mvtn_amd import MVRenderer
    B, P, V, size, K, r = 2, 600, 2, 40, 3, 0.06
    vl, fl = _mesh_batch([("M", 0), ("M", 1)])
    pts = np.stack([syn.surface_points(v, f, P, seed=i) for i, (v, f) in enumerate(zip(vl, fl))])
    az = np.array([[10.0, 200.0], [75.0, -40.0]], np.float32)
    el = np.array([[20.0, -30.0], [5.0, 60.0]], np.float32)
    di = np.array([[1.0, 1.3], [0.9, 1.1]], np.float32)
    col = np.random.default_rng(5).uniform(0, 1, (B, P, 3)).astype(np.float32)
    rend = MVRenderer(V, size, pc_rendering=True, object_color="custom", background_color="black", points_radius=r,
                      points_per_pixel=K).eval()
    ta, te, td = (_t(x).requires_grad_(True) for x in (az, el, di))
    images, _ = rend(None, torch.from_numpy(pts), ta, te, td, color=torch.from_numpy(col))
    want = oracle.render_points(pts, az, el, di, size, radius=r, point_colors=col, bg_color=(0, 0, 0), points_per_pixel=K)
    assert np.abs(images.detach().cpu().numpy() - want["images"]).max() <= 1e-6
    G = np.random.default_rng(1).normal(size=images.shape)
    images.backward(_t(G))
    ref = _points_grad_reference(oracle, pts, col, (1, 1, 1), az, el, di, size, r, G, want["idx"], want["dist2"])
    for name, got, rf in zip(("azim", "elev", "dist"), (ta.grad, te.grad, td.grad), ref):
        got = got.cpu().numpy().astype(np.float64)
        scale = np.abs(rf).max()
        rel = np.abs(got - rf).max() / scale
        print(name, "rel err", rel, "scale", scale)
        assert scale > 1e-3 and rel < 2e-3, (name, got, rf)


@pytest.mark.parametrize("mode", ["device", "sync"])
def test_rotation_remedy(mode):
    """ops.py:156-165: an invalid R (elev = 89.9999; exactly 90 is repaired inside look_at) triggers a
    random re-draw of ALL views; the render then uses valid cameras and gradients still flow."""
    from mvtn_amd import MVRenderer, Meshes
    from mvtn_amd.helpers import check_valid_rotation_matrix
    v, f = syn.synthetic_mesh("S", 0)
    az, el, di = _t([[0.0, 90.0]]).requires_grad_(True), _t([[89.9999, 30.0]]).requires_grad_(True), _t([[2.2, 2.2]])
    torch.manual_seed(0)
    r = MVRenderer(2, 32, pc_rendering=False, light_direction="fixed", rotation_check=mode).eval()
    images, cams = r([Meshes([v], [f])], None, az, el, di)
    assert check_valid_rotation_matrix(cams.R)
    assert torch.isfinite(images).all()
    images.sum().backward()
    assert torch.isfinite(az.grad).all() and torch.isfinite(el.grad).all()
    if mode == "device":
        r.check_status()  # no exhaustion -> no exit
    r_off = MVRenderer(2, 32, pc_rendering=False, light_direction="fixed", rotation_check="off").eval()
    images2, cams2 = r_off([Meshes([v], [f])], None, az.detach(), el.detach(), di)
    assert not check_valid_rotation_matrix(cams2.R)
    assert not torch.equal(images2, images)


def test_rotation_remedy_device_statistics():
    """The device-side re-draw follows ops.py:160-161: every view moves by azim + U[0,180), elev + U[0,90)."""
    from mvtn_amd import functional as F
    n = 512
    az, el, di = _t(np.zeros(n)), _t(np.concatenate([[89.9999], np.zeros(n - 1)])), _t(np.full(n, 2.2))
    lib_out = F._LookAt.apply(az, el, di, 20, 1234)
    R, T, status = lib_out
    assert int(status[0]) >= 1 and int(status[1]) == 0
    # recover the angles actually used from the camera centres: C = -R T
    C = -torch.bmm(R, T[:, :, None])[:, :, 0].cpu().numpy().astype(np.float64)
    elev_used = np.degrees(np.arcsin(C[:, 1] / 2.2))
    azim_used = np.degrees(np.arctan2(C[:, 0], C[:, 2])) % 360
    if int(status[0]) == 1:  # view 0 started at elev 89.9999 and may pass the pole: judge the others
        a, e = np.delete(azim_used, 0), np.delete(elev_used, 0)
        assert 0 <= a.min() and a.max() < 180.01 and 0 <= e.min() and e.max() < 90.01
        assert abs(a.mean() - 90) < 10 and abs(e.mean() - 45) < 5
    # a different seed gives a different draw; the same seed reproduces it
    R2, _, _ = F._LookAt.apply(az, el, di, 20, 1234)
    R3, _, _ = F._LookAt.apply(az, el, di, 20, 99)
    assert torch.equal(R, R2) and not torch.equal(R, R3)


def test_full_size_properties():
    """BASELINE config C2 at full size (B=16, V=12, 224^2): size-independent properties —
    determinism, background constant, silhouette fraction of a unit sphere (KA2), views of the
    same mesh from mirrored azimuths are mirror images in coverage count."""
    from mvtn_amd import MVRenderer, Meshes
    B, V = 16, 12
    meshes = [Meshes([v], [f]) for v, f in (syn.synthetic_mesh("M", s) for s in range(B))]
    az, el, di = syn.spherical_views(V)
    ta, te, td = (_t(np.tile(x, (B, 1))) for x in (az, el, di))
    r = MVRenderer(V, 224, pc_rendering=False, light_direction="fixed", rotation_check="off").eval()
    a, _ = r(meshes, None, ta, te, td)
    b, _ = r(meshes, None, ta, te, td)
    assert a.shape == (B, V, 3, 224, 224)
    assert torch.equal(a, b), "render is not deterministic"
    bg = np.float32(1.0) / np.float32(1.00001)
    assert float(a.max()) <= 1.0 + 1e-6 and float(a.min()) >= 0.49
    assert (a == float(bg)).float().mean() > 0.3  # background pixels are exactly 0.99999 (KA6)
    # KA2: unit sphere at d = 2.2, fov 60: silhouette radius 0.88388 NDC -> 61.36 % of the image
    vs, fs = syn.icosphere(5)
    sph = [Meshes([torch.from_numpy(vs.astype(np.float32))], [torch.from_numpy(fs)])]
    r1 = MVRenderer(1, 224, pc_rendering=False, light_direction="fixed", background_color="black", rotation_check="off").eval()
    img, _ = r1(sph, None, _t([[30.0]]), _t([[20.0]]), _t([[2.2]]))
    frac = float((img[0, 0].sum(0) > 0).float().mean())
    assert abs(frac - 0.6136) < 0.004, frac


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


def test_edge_case_flat_band_many_faces_per_strip(oracle):
    """20 480 faces squeezed into a few image rows: > 8192 faces (several scan passes) and far more than
    MVR_LIST_CAP faces touching one strip (list windows), thousands of coincident-depth candidates per pixel."""
    v, f = syn.synthetic_mesh("L", 3)
    v = v.copy()
    v[:, 1] *= 0.03
    az, el, di = np.array([[20.0, 200.0]], np.float32), np.array([[0.0, 3.0]], np.float32), np.full((1, 2), 2.2, np.float32)
    img, p2f, want = _render_pair(oracle, [v], [f], az, el, di, 224)
    assert int((p2f != want["pix_to_face"]).sum()) == 0
    assert (p2f >= 0).sum() > 1000
    well = _well_conditioned_pixels([v], [f], want["pix_to_face"], 2)
    assert np.abs(img - want["images"]).reshape(2, 3, 224, 224).max(1)[well].max() <= TOL_PIXEL


def test_edge_case_huge_and_tiny_faces(oracle):
    """Two triangles covering the whole frame (every strip, every lane of the balanced loop shares one face), a
    sub-pixel sliver, a degenerate (zero-area) face and a face behind the camera."""
    big = [[-4, -4, 0.1], [4, -4, 0.1], [0, 4, 0.1], [-4, 4, -0.2], [4, 4, -0.2], [0, -4, -0.2]]
    sliver = [[0.0, 0.0, 0.5], [0.004, 0.0, 0.5], [0.0, 0.004, 0.5]]
    degenerate = [[0.1, 0.1, 0.6], [0.2, 0.2, 0.6], [0.3, 0.3, 0.6]]
    behind = [[-1, -1, 5.0], [1, -1, 5.0], [0, 1, 5.0]]
    v = np.asarray(big + sliver + degenerate + behind, np.float32)
    f = np.arange(len(v)).reshape(-1, 3).astype(np.int64)
    az, el, di = np.array([[0.0, 10.0, 180.0]], np.float32), np.array([[0.0, 5.0, 0.0]], np.float32), np.full((1, 3), 2.2, np.float32)
    for size, strip in ((64, 0), (224, 0), (50, 3)):
        img, p2f, want = _render_pair(oracle, [v], [f], az, el, di, size, strip=strip)
        assert int((p2f != want["pix_to_face"]).sum()) == 0, (size, strip)
        assert np.abs(img - want["images"]).max() <= TOL_PIXEL
        assert (p2f[0] >= 0).mean() > 0.9


def test_edge_case_strip_list_overflow_falls_back_to_the_scan(oracle):
    """400 large overlapping triangles: every face reaches most strips, the per-strip lists of the pre-pass exceed their
    reserved capacity (3 entries per face + slack), the view is flagged on the device and the rasteriser scans the
    2-byte bins instead — in several windows of its 256-entry LDS list; the near/far split and the hierarchical-z
    rejection are off on that path.  A second mesh in the batch keeps its lists: both paths in one launch."""
    rng = np.random.default_rng(11)
    n = 400
    c = rng.uniform(-0.3, 0.3, (n, 1, 3))
    tri = c + rng.uniform(-0.9, 0.9, (n, 3, 3)) * np.array([1.0, 1.0, 0.15])
    v_big = tri.reshape(-1, 3).astype(np.float32)
    f_big = np.arange(3 * n).reshape(-1, 3).astype(np.int64)
    v_small, f_small = syn.synthetic_mesh("S", 3)
    az = np.array([[0.0, 77.0], [30.0, 200.0]], np.float32)
    el = np.array([[0.0, 25.0], [-20.0, 40.0]], np.float32)
    di = np.full((2, 2), 2.2, np.float32)
    for size in (224, 96):
        img, p2f, want = _render_pair(oracle, [v_big, v_small], [f_big, f_small], az, el, di, size)
        assert int((p2f != want["pix_to_face"]).sum()) == 0, size
        ok = _well_conditioned_pixels([v_big, v_small], [f_big, f_small], want["pix_to_face"], 2)
        err = np.abs(img - want["images"]).reshape(4, 3, size, size).max(1)
        assert err[ok].max() <= TOL_PIXEL
        assert (p2f[:2] >= 0).mean() > 0.5      # the triangle soup covers most of its views


def test_edge_case_mesh_too_large_for_the_fused_prepass(oracle):
    """40 962 vertices / 81 920 faces: the projected vertices no longer fit in LDS (limit 12 288), so the forward takes
    the separate launches (chunked vertex normals, k_project_verts, k_face_bins, no strip lists) and the rasteriser scans
    the bins in several passes of 8192 faces; batched with a small mesh (ragged batch, max_verts decides the path)."""
    v_big, f_big = syn.synthetic_mesh(6, 5)
    v_small, f_small = syn.synthetic_mesh("S", 1)
    assert len(v_big) > 12288
    az = np.array([[15.0, 250.0], [100.0, -60.0]], np.float32)
    el = np.array([[20.0, -35.0], [5.0, 50.0]], np.float32)
    di = np.full((2, 2), 2.2, np.float32)
    img, p2f, want = _render_pair(oracle, [v_big, v_small], [f_big, f_small], az, el, di, 96)
    assert int((p2f != want["pix_to_face"]).sum()) == 0
    assert np.abs(img - want["images"]).max() <= TOL_PIXEL
    assert (p2f >= 0).mean() > 0.3


def test_edge_case_nothing_visible_and_large_image(oracle):
    """Object behind / beside the camera -> pure background; a 300^2 image (more than 254 rows: row-bins of 2 strips)."""
    v, f = syn.synthetic_mesh("S", 0)
    far = v + np.array([0.0, 0.0, 50.0], np.float32)
    az, el, di = np.array([[0.0]], np.float32), np.array([[0.0]], np.float32), np.array([[2.2]], np.float32)
    img, p2f, want = _render_pair(oracle, [far], [f], az, el, di, 96)
    assert (p2f == -1).all() and np.array_equal(p2f, want["pix_to_face"])
    assert np.allclose(img[0, 0, 0], 0.1) and np.allclose(img[0, 0, 2], 0.3)
    img, p2f, want = _render_pair(oracle, [v], [f], az, el, di, 300, persp=False)
    assert int((p2f != want["pix_to_face"]).sum()) == 0 and np.abs(img - want["images"]).max() <= TOL_PIXEL
    img, p2f, want = _render_pair(oracle, [v], [f], az, el, di, 2200 // 4, strip=2)   # 550 rows / 2 = 275 strips > 254 bins
    assert int((p2f != want["pix_to_face"]).sum()) == 0


def test_edge_case_points_ragged_counts(oracle):
    """P not a multiple of the 4-point scan vector, points behind the camera, off-screen, and stacked at one pixel."""
    from mvtn_amd import functional as F
    rng = np.random.default_rng(0)
    pts = rng.uniform(-0.8, 0.8, (2, 1027, 3)).astype(np.float32)
    pts[0, :5] = [0.1, 0.1, 0.0]            # five identical points: smallest index wins
    pts[1, :50, 2] = 7.0                    # far off along +z (behind the orthographic camera after T)
    pts[1, 50:60, 0] = 9.0                  # off-screen
    az, el = np.array([[0.0, 77.0]] * 2, np.float32), np.array([[0.0, -20.0]] * 2, np.float32)
    di = np.array([[1.0, 1.4]] * 2, np.float32)
    want = oracle.render_points(pts, az, el, di, 100, radius=0.03, obj_color=(1, 1, 1), bg_color=(0, 0, 0))
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    for strip in (0, 1, 9):
        params = F.points_params(2, 1027, 2, 100, 0.03, (1, 1, 1), (0, 0, 0), strip_rows=strip)
        img, idx = F.render_points_fragments(_t(pts), None, R, T, _t(di.reshape(-1)), params)
        assert int((idx.cpu().numpy() != want["idx"]).sum()) == 0, strip
        assert np.abs(img.cpu().numpy() - want["images"]).max() <= 1e-6


from reg_composition import regularize_composition as _regularize_composition  # noqa: E402  (tests/reg_composition.py)


@pytest.mark.parametrize("B,V,C,H,p,augment,crop_ratio", [
    (3, 4, 3, 224, 0.3, True, 0.3),      # the reference's training configuration with both switches on
    (2, 5, 3, 64, 0.0, True, 0.3),       # augmentation only
    (4, 3, 3, 32, 0.5, False, 0.3),      # view dropout only
    (2, 2, 1, 37, 0.25, True, 0.6),      # odd width (scalar store path), one channel, wide padding
    (1, 3, 3, 8, 0.0, True, 2.0),        # padding wider than the image: windows that see only replicated border
])
def test_regularize_views_matches_the_torch_composition(B, V, C, H, p, augment, crop_ratio):
    """Row f3 (ops.py:168-176): the fused HIP pass against the reference's op-by-op composition under the same
    seeds.  Forward: bit-exact.  Backward: the composition's replication-pad / index backward sum in another
    (atomic) order, so 1e-5 relative on the border sums; interior pixels are bit-exact."""
    from mvtn_amd import regularize as rg
    dev = _dev()
    g = torch.Generator().manual_seed(B * 100 + V)
    x0 = torch.rand(B, V, C, H, H, generator=g).to(dev)
    G = torch.randn(B, V, C, H, H, generator=g).to(dev)
    for seed in (0, 1, 2):
        xr = x0.clone().requires_grad_(True)
        torch.manual_seed(seed)
        want = _regularize_composition(xr, p, augment, crop_ratio)
        want.backward(G)
        xm = x0.clone().requires_grad_(True)
        torch.manual_seed(seed)
        got = rg.regualarize_rendered_views(xm, p, augment, crop_ratio)
        assert got.shape == want.shape and torch.equal(got, want.detach()), (seed, (got - want).abs().max())
        got.backward(G)
        err = (xm.grad - xr.grad).abs().max().item()
        assert err <= 1e-5 * max(1.0, xr.grad.abs().max().item()), (seed, err)
        inner = (slice(None),) * 3 + (slice(1, H - 1), slice(1, H - 1))
        assert torch.equal(xm.grad[inner], xr.grad[inner])
        xm.grad = None
        torch.manual_seed(seed)
        rg.regualarize_rendered_views(xm, p, augment, crop_ratio).backward(G)
        g2 = xm.grad.clone()
        xm.grad = None
        torch.manual_seed(seed)
        rg.regualarize_rendered_views(xm, p, augment, crop_ratio).backward(G)
        assert torch.equal(g2, xm.grad), "fixed summation order: bit-reproducible backward"


def test_regularize_views_after_the_renderer_trains_end_to_end():
    """The reference's step (run_mvtn.py:182-195): render -> regularise -> loss.backward(); gradients reach the view
    parameters through both fused stages, and equal the chain rule applied by hand through the regulariser's draws."""
    from mvtn_amd import MVRenderer, Meshes, regularize as rg
    vl, fl = _mesh_batch([("S", 0), ("S", 1)])
    B, V, size = 2, 4, 64
    az, el, di = _views(B, 4, 9)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative").eval()
    meshes = [Meshes([v], [f]) for v, f in zip(vl, fl)]
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(4)).to(_dev())
    ta, te, td = (_t(x).requires_grad_(True) for x in (az, el, di))
    images, _ = r(meshes, None, ta, te, td)
    torch.manual_seed(5)
    draws = rg.draw_regularization(B, V, size, _dev(), 0.3, True, 0.3)
    out = rg.apply_regularization(images, *draws)
    out.backward(G)
    got = [t.grad.clone() for t in (ta, te, td)]
    # by hand: pull G back through the regulariser alone, then through the renderer alone
    x = images.detach().clone().requires_grad_(True)
    rg.apply_regularization(x, *draws).backward(G)
    tb = [_t(v).requires_grad_(True) for v in (az, el, di)]
    r(meshes, None, *tb)[0].backward(x.grad)
    for a, b in zip(got, tb):
        assert torch.equal(a, b.grad)
    dropped = (draws[0] == 0).reshape(B, V)
    assert dropped.any() and (out[dropped] == 0).all() and (got[0][dropped] == 0).all()


# ---- SURVEY section 8c known answers, straight through the HIP library: closed-form expectations, no oracle in the loop ----

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


def test_known_answers_mesh_hip():
    s = 1.0 / np.tan(np.radians(30.0))
    # KA2: unit sphere at d = 2.2 -> disc of NDC radius 1/(tan30 sqrt(d^2 - 1)) = 0.88388 -> 61.36 % of the pixels
    v, f = syn.icosphere(5)
    _, p2f = _hip_render_mesh(v, f, 30.0, 20.0, 2.2, 224)
    assert abs((p2f >= 0).mean() - np.pi * 0.88388 ** 2 / 4) < 0.004
    # KA3: lattice-aligned fronto-parallel right triangle: exactly the pixel centres with all barycentrics > 0
    S, d = 16, 2.0
    ndc = np.array([[-0.75, -0.75], [0.75, -0.75], [-0.75, 0.75]])
    world = np.stack([-ndc[:, 0] * d / s, ndc[:, 1] * d / s, np.zeros(3)], 1)
    _, p2f = _hip_render_mesh(*_tris([world]), 0.0, 0.0, d, S, persp=False)
    c = -1 + (2 * np.arange(S) + 1) / S
    X, Y = np.meshgrid(c[::-1], c[::-1])           # output (row i, col j) <-> (x, y) = (c[S-1-j], c[S-1-i])
    assert np.array_equal(p2f >= 0, (X > -0.75) & (Y > -0.75) & (X + Y < 0.0))
    # KA4: coincident faces -> the lower index wins; a nearer face beats a lower index
    tri = [[-0.5, -0.5, 0.0], [0.5, -0.5, 0.0], [0.0, 0.6, 0.0]]
    near = [[-0.5, -0.5, 0.3], [0.5, -0.5, 0.3], [0.0, 0.6, 0.3]]
    assert set(np.unique(_hip_render_mesh(*_tris([tri, tri, tri]), 0.0, 0.0, 2.2, 32)[1])) == {-1, 0}
    assert set(np.unique(_hip_render_mesh(*_tris([tri, near]), 0.0, 0.0, 2.2, 32)[1])) == {-1, 1}
    # KA5 + KA6: constant-normal plane, closed-form Phong; background pixels carry the background colour exactly
    quad_v = np.asarray([[-2, -2, 0], [2, -2, 0], [2, 2, 0], [-2, 2, 0]], np.float32)
    quad_f = np.asarray([[0, 1, 2], [0, 2, 3]], np.int64)
    col = 0.99999
    cen = -1 + (2 * np.arange(8) + 1) / 8
    for light, cosang in (((0.0, 1.0, 0.0), 0.0), ((0.0, 0.0, 1.0), 1.0), ((0.0, 1.0, 1.0), np.sqrt(0.5))):
        img, p2f = _hip_render_mesh(quad_v, quad_f, 0.0, 0.0, 2.2, 8, light=light, obj=(col,) * 3, bg=(0.25, 0.5, 0.75))
        assert (p2f >= 0).mean() > 0.8
        checked = 0
        for i in range(8):
            for j in range(8):
                if p2f[i, j] < 0:
                    assert tuple(img[:, i, j]) == (0.25, 0.5, 0.75)
                    continue
                p = np.array([-cen[7 - j] * 2.2 / s, cen[7 - i] * 2.2 / s, 0.0])
                vh = np.array([0, 0, 2.2]) - p
                vh /= np.linalg.norm(vh)
                l = np.asarray(light, float) / np.linalg.norm(light)
                r = -l + 2 * cosang * np.array([0, 0, 1.0])
                alpha = max(0.0, float(vh @ r)) * (1.0 if cosang > 0 else 0.0)
                want = (0.5 + 0.3 * max(0.0, cosang)) * col + 0.2 * alpha ** 64
                assert np.abs(img[:, i, j] - want).max() < 2e-5, (light, i, j)
                checked += 1
        assert checked > 50


def test_known_answers_points_hip(oracle):
    """KA7 through the HIP path: r = 0.006 is smaller than the pixel half-diagonal at 224^2, so a point on a pixel
    corner hits nothing and a point on a pixel centre exactly that pixel; nearer wins, ties go to the smaller index.
    (The oracle is used only for the look-at matrices that place the points.)"""
    from mvtn_amd import functional as F
    S, d = 224, 1.0
    R64, T64 = oracle.look_at([0.0], [0.0], [d], np.float64)

    def world_of(x, y, z=1.0):
        return ((np.array([x, y, z]) - T64[0]) @ R64[0].T) * d

    def render(pts, size, radius, K=1):
        R, T = F.look_at_view_transform(_t([d]), _t([0.0]), _t([0.0]))
        params = F.points_params(1, len(pts), 1, size, radius, (1, 0, 0), (0, 0, 0), points_per_pixel=K)
        img, idx = F.render_points_fragments(_t(np.asarray([pts], np.float32)), None, R, T, _t([d]), params)
        return img[0, 0].cpu().numpy(), idx[0].cpu().numpy()
    img, idx = render([world_of(0.0, 0.0)], S, 0.006)
    assert (idx < 0).all() and (img == 0).all()
    img, idx = render([world_of(-1 + (2 * 100 + 1) / S, -1 + (2 * 50 + 1) / S)], S, 0.006)
    assert np.argwhere(idx >= 0).tolist() == [[S - 1 - 50, S - 1 - 100]]
    assert np.allclose(img[:, S - 1 - 50, S - 1 - 100], [1, 0, 0], atol=1e-6)
    _, idx = render([world_of(0.1, 0.1, 1.5), world_of(0.1, 0.1, 0.7), world_of(0.1, 0.1, 0.7)], 64, 0.05)
    assert set(np.unique(idx)) == {-1, 1}


@pytest.mark.parametrize("rotation_check", ["off", "device"])
def test_training_step_is_capturable_in_a_hip_graph(rotation_check):
    """Render + backward issue no host synchronisation and allocate only through torch's caching allocator, so a whole
    step (forward, regulariser, loss, backward) can be captured in a HIP graph (torch.cuda.CUDAGraph) and replayed
    with new view parameters written into the captured input tensors: replays equal eager execution bit for bit."""
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes, regularize as rg
    dev = _dev()
    vl, fl = _mesh_batch([("S", 0), ("M", 1)])
    B, V, size = 2, 4, 64
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative",
                   rotation_check=rotation_check).eval()
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(4)).to(dev)
    draws = rg.draw_regularization(B, V, size, dev, 0.3, True, 0.3)
    sets = [tuple(_t(x) for x in _views(B, 4, seed)) for seed in (1, 2, 3)]

    def step(a, e, d):
        img, _ = r(packed, None, a, e, d)
        out = rg.apply_regularization(img, *draws)
        return out, torch.autograd.grad((out * G).sum(), (a, e, d))

    eager = []
    for a, e, d in sets:
        leaves = [x.clone().requires_grad_(True) for x in (a, e, d)]
        out, grads = step(*leaves)
        eager.append((out.detach().clone(), [g.clone() for g in grads]))
    sa, se, sd = (x.clone().requires_grad_(True) for x in sets[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):                          # warm-up outside the capture (workspace sizes, caches)
            step(sa, se, sd)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        g_out, g_grads = step(sa, se, sd)
    for (a, e, d), (want_out, want_grads) in zip(sets, eager):
        with torch.no_grad():
            sa.copy_(a); se.copy_(e); sd.copy_(d)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(g_out, want_out)
        for got, want in zip(g_grads, want_grads):
            assert torch.equal(got, want)


@pytest.mark.parametrize("seed", [101, 202, 303])
def test_random_configurations_slice(oracle, seed):
    """A seeded slice of tests/stress_parity.py inside the suite: random anisotropically scaled / shifted meshes,
    views anywhere on the sphere, cameras from inside the bounding sphere (dist 0.9: vertices behind the camera,
    faces crossing z = 0, NDC coordinates far outside the screen) to 3.0, odd image sizes and strip heights, both
    PyTorch3D switch settings.  Pixel->face maps bit-exact; colours within TOL_PIXEL on well-conditioned pixels."""
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    rng = np.random.default_rng(seed)
    for case in range(4):
        B, V = int(rng.integers(1, 4)), int(rng.integers(1, 5))
        size = int(rng.choice([32, 45, 64, 100, 128]))
        strip = int(rng.choice([0, 0, 3, 5, 8]))
        persp, cull = bool(rng.integers(0, 2)), bool(rng.integers(0, 4) == 0)
        vl, fl = [], []
        for b in range(B):
            v, f = syn.cad_soup(int(rng.integers(0, 1000))) if rng.integers(0, 4) == 0 else \
                syn.synthetic_mesh(str(rng.choice(["S", "M"])), int(rng.integers(0, 1000)))
            vl.append((v * rng.uniform(0.5, 1.3, 3) + rng.uniform(-0.2, 0.2, 3)).astype(np.float32))
            fl.append(f)
        az = rng.uniform(-180, 180, (B, V)).astype(np.float32)
        el = rng.uniform(-80, 80, (B, V)).astype(np.float32)
        di = rng.uniform(0.9, 3.0, (B, V)).astype(np.float32)
        if case == 0:
            di[:] = 0.9                      # always include the camera-inside-the-object case
        obj, bg = (0.9, 0.8, 0.7), (0.1, 0.2, 0.3)
        want = oracle.render_meshes(vl, fl, az, el, di, size, obj_color=obj, bg_color=bg, perspective_correct=persp,
                                    cull_backfaces=cull, mode=1)
        packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
        R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
        # z_clip = 0: the no-clipping rule on both sides, so that faces crossing the near region stay comparable bit for bit
        params = F.mesh_params(V, size, 1, obj, bg, perspective_correct=persp, cull_backfaces=cull, strip_rows=strip, z_clip=0.0)
        img, p2f = F.render_meshes_fragments(packed, R, T, _t([[0, 1, 0]]), params)
        p2f, img = p2f.cpu().numpy(), img.cpu().numpy()
        assert np.array_equal(p2f, want["pix_to_face"]), (seed, case, int((p2f != want["pix_to_face"]).sum()))
        well = _well_conditioned_pixels(vl, fl, want["pix_to_face"], V)
        diff = np.abs(img - want["images"]).reshape(B * V, 3, size, size).max(1)
        assert float(diff[well].max()) <= TOL_PIXEL, (seed, case, float(diff[well].max()))


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_mesh_backward_random_configurations(oracle, seed):
    """The analytic backward against fp64 central differences of the oracle (pix_to_face held fixed, as autograd
    through a hard rasteriser does) on random scaled / shifted meshes, views anywhere on the sphere, distances from 1.3
    to 3.0, both light modes and both perspective-correction settings."""
    from mvtn_amd import MVRenderer, Meshes
    rng = np.random.default_rng(seed)
    B, V, size = int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.choice([32, 48, 64]))
    persp, relative = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    vl, fl = [], []
    for b in range(B):
        v, f = syn.synthetic_mesh("S", int(rng.integers(0, 1000)))
        vl.append((v * rng.uniform(0.6, 1.2, 3) + rng.uniform(-0.15, 0.15, 3)).astype(np.float32))
        fl.append(f)
    az = rng.uniform(-180, 180, (B, V)).astype(np.float32)
    el = rng.uniform(-75, 75, (B, V)).astype(np.float32)
    di = rng.uniform(1.3, 3.0, (B, V)).astype(np.float32)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white",
                   light_direction="relative" if relative else "fixed", perspective_correct=persp, z_clip=0.0).eval()
    leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
    images, _ = r([Meshes([v], [f]) for v, f in zip(vl, fl)], None, *leaves)
    ldir = oracle.camera_position(az, el, di) if relative else np.array([[0.0, 1.0, 0.0]], np.float32)
    want = oracle.render_meshes(vl, fl, az, el, di, size, light_dir=ldir, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3,
                                perspective_correct=persp)
    assert np.abs(images.detach().cpu().numpy() - want["images"]).max() <= TOL_PIXEL
    G = rng.normal(size=images.shape)
    images.backward(_t(G))
    fd = _fd_grad_full(oracle, vl, fl, az, el, di, size, G, want["pix_to_face"], relative, persp)
    for name, leaf, ref in zip(("azim", "elev", "dist"), leaves, fd):
        got = leaf.grad.cpu().numpy().astype(np.float64)
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        assert rel < 2e-3, (seed, name, rel, got, ref)


@pytest.mark.parametrize("degenerate", [False, True])
def test_fused_point_views_forward_equals_chained_calls(degenerate):
    """mvr_render_point_views_forward (look-at inside the projection pre-pass) against mvr_look_at_checked +
    mvr_render_points_forward with the same remedy seed: identical R, T, status and images — also when a view at
    elev = 89.9999 makes the device-side remedy re-draw all views."""
    from mvtn_amd import MVRenderer, functional as F
    B, P, V, size = 3, 2048, 5, 128
    rng = np.random.default_rng(8)
    pts = _t(rng.normal(size=(B, P, 3)) * 0.35)
    cols = _t(rng.uniform(0, 1, (B, P, 3)))
    az = rng.uniform(-180, 180, (B, V)).astype(np.float32)
    el = rng.uniform(-70, 70, (B, V)).astype(np.float32)
    di = rng.uniform(0.9, 2.0, (B, V)).astype(np.float32)
    if degenerate:
        el[1, 2] = 89.9999      # (exactly 90 is repaired inside the look-at)
    params = F.points_params(B, P, V, size, 0.01, (1, 1, 1), (0.2, 0.3, 0.4))
    a, e, d = _t(az.reshape(-1)), _t(el.reshape(-1)), _t(di.reshape(-1))
    img1, R1, T1, st1 = F.render_point_views(pts, cols, a, e, d, params, max_trials=20, seed=1234)
    R2, T2, st2 = F.look_at_view_transform(d, e, a, return_status=True, max_trials=20, seed=1234)
    img2, _ = F.render_points_fragments(pts, cols, R2, T2, d, params)
    assert torch.equal(R1, R2) and torch.equal(T1, T2) and torch.equal(st1, st2)
    assert torch.equal(img1, img2)
    assert (int(st1[0]) > 0) == degenerate and int(st1[1]) == 0
    # the module takes the fused call under no_grad and the chained autograd nodes otherwise: same images
    r = MVRenderer(V, size, pc_rendering=True, object_color="custom", background_color="white", points_radius=0.01,
                   rotation_check="off" if not degenerate else "device").eval()
    ta, te, td = _t(az), _t(el), _t(di)
    with torch.no_grad():
        fused, cams_f = r(None, pts, ta, te, td, color=cols)
    if not degenerate:
        chained, cams_c = r(None, pts, ta.clone().requires_grad_(True), te, td, color=cols)
        assert torch.equal(fused, chained.detach()) and torch.equal(cams_f.R, cams_c.R.detach()) and torch.equal(cams_f.T, cams_c.T.detach())
    else:
        r.check_status()        # the re-draw succeeded: no "Remedy did not work"
        from mvtn_amd.helpers import check_valid_rotation_matrix
        assert check_valid_rotation_matrix(cams_f.R)


def test_render_and_save_writes_grid_and_camera_plot(tmp_path):
    """models/renderer.py:200-207: a no-grad forward, the image grid of the first object and the camera plot."""
    pytest.importorskip("matplotlib")
    from mvtn_amd import MVRenderer, Meshes
    v, f = syn.synthetic_mesh("S", 0)
    V, size = 4, 64
    r = MVRenderer(V, size, pc_rendering=False, light_direction="fixed").eval()
    az, el, di = syn.circular_views(V)
    ta, te, td = _t(az[None]).requires_grad_(True), _t(el[None]), _t(di[None])
    img_path, cam_path = str(tmp_path / "views.png"), str(tmp_path / "cams.png")
    r.render_and_save([Meshes([v], [f])], None, ta, te, td, img_path, cam_path)
    for p in (img_path, cam_path):
        data = open(p, "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n" and len(data) > 500
    import struct
    w, h = struct.unpack(">II", open(img_path, "rb").read()[16:24])
    assert (w, h) == (V * (size + 2) + 2, size + 4)      # one row of V views, 2-pixel padding (util.py:667-673)
    assert ta.grad is None


def test_stream_pipeline_gives_the_same_images():
    """mvtn_amd.pipeline.StreamPipeline: independent forwards issued round-robin on two streams, then joined."""
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes
    from mvtn_amd.pipeline import StreamPipeline
    V, size = 6, 128
    r = MVRenderer(V, size, pc_rendering=False, light_direction="fixed", rotation_check="off").eval()
    jobs = []
    for k in range(6):
        vl, fl = _mesh_batch([("S", k), ("M", 10 + k)])
        packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
        jobs.append((packed,) + tuple(_t(x[:, :V]) for x in _views(2, 6, 50 + k)))
    with torch.no_grad():
        want = [r(p, None, a, e, d)[0].clone() for p, a, e, d in jobs]
    torch.cuda.synchronize()
    pipe = StreamPipeline(2)
    got = []
    for rep in range(3):
        got += [pipe.submit(r, p, None, a, e, d)[0] for p, a, e, d in jobs]
    pipe.join()
    total = torch.stack([g.sum() for g in got])     # consumed on the caller's stream after the join
    torch.cuda.synchronize()
    for i, g in enumerate(got):
        assert torch.equal(g, want[i % len(jobs)]), i
    assert torch.isfinite(total).all()


# ---- round 2: the fast rasteriser (exact edge signs + provably ordered affine inverse depth) --------------------------

@pytest.mark.parametrize("name,specs,V,size,dist,jitter", [
    ("C2 slice", [("M", 0), ("M", 1), ("M", 2), ("M", 3)], 12, 224, 2.2, False),
    ("tier L, 20 views", [("L", 2), ("L", 3)], 20, 224, 2.2, True),
    ("ragged + CAD soup", [("S", 3), ("L", 4), ("soup", 5), ("M", 6)], 6, 224, 2.2, True),
    ("close-up", [("M", 0), ("M", 1)], 12, 224, 1.2, True),
    ("camera inside the object", [("M", 0), ("M", 1)], 12, 224, 0.8, True),
    ("odd image size", [("S", 10), ("soup", 11)], 2, 37, 2.2, True),
    ("128", [("M", 7)], 4, 128, 2.2, False),
])
def test_fast_rasteriser_equals_exact_kernel(name, specs, V, size, dist, jitter):
    """The default mesh forward (k_mesh_raster_fast: fall-back to the exact body per strip) against MVR_RASTER_EXACT_ONLY
    (k_mesh_raster_shade on every strip): pixel->face maps identical, images within the colour tolerance (the two shade
    with differently ordered but equivalent arithmetic), and the number of strips the fast path gave up on."""
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    vl, fl = _mesh_batch(specs)
    B = len(vl)
    az, el, di = _views(B, V if V in (4, 6, 12, 20) else 4, 7, jitter)
    az, el = az[:, :V], el[:, :V]
    di = np.full((B, V), dist, np.float32) + (di[:, :V] - 2.2 if jitter else 0)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    light = _t(np.array([[0.0, 1.0, 0.0]], np.float32))
    obj = bg = (0.99999,) * 3
    out = {}
    for mode in ("fast", "exact"):
        params = F.mesh_params(V, size, 1, obj, bg, exact_raster=(mode == "exact"))
        img, p2f, cnt = F.render_meshes_fragments(packed, R, T, light, params, return_redo=True)
        torch.cuda.synchronize()
        out[mode] = (img.cpu().numpy(), p2f.cpu().numpy(), int(cnt.item()))
    n_strips = B * V * ((size + 7) // 8)
    print("%s: %d of %d strips fell back to the exact rasteriser" % (name, out["fast"][2], n_strips))
    assert np.array_equal(out["fast"][1], out["exact"][1]), "%d pixels differ" % int((out["fast"][1] != out["exact"][1]).sum())
    assert np.abs(out["fast"][0] - out["exact"][0]).max() <= 5e-5
    if "soup" not in str(specs) and dist >= 1.6:   # (nearer than 1.5 the clip plane z = 0.5 cuts faces: those views take the exact path)
        assert out["fast"][2] <= n_strips // 100, "ordinary meshes should almost never need the exact path"
    if "soup" in str(specs) and size % 2 == 0:   # (odd widths take the exact kernel: nothing to settle)
        # coincident double-sided faces: undecidable for the affine depth keys.  Rounds 2-4 handed such strips to the exact kernel's
        # body; round 5 settles the recorded (pixel, face) candidates inside the fast kernel (mvr_mesh_fast_stats counts them)
        _, _, st = F.render_meshes_fragments(packed, R, T, light, F.mesh_params(V, size, 1, obj, bg), return_stats=True)
        st = [int(x) for x in st.cpu()]
        assert st[1] > 0 and st[0] <= n_strips // 50, st   # (a strip whose candidates overflow their area is still redone; which one depends on the list order of the run)


def test_full_size_c2_all_meshes_against_the_oracle(oracle):
    """BASELINE configs[1] at FULL size: all 16 tier-M meshes x 12 spherical views at 224^2 against the oracle
    (round 1 checked a 2-mesh slice): pixel->face bit-exact, images within 1e-4."""
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    B, V, size = 16, 12, 224
    vl, fl = _mesh_batch([("M", s) for s in range(B)])
    az, el, di = _views(B, V, 0, jitter=False)
    obj = bg = (0.99999,) * 3
    want = oracle.render_meshes(vl, fl, az, el, di, size, obj_color=obj, bg_color=bg, mode=1)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.mesh_params(V, size, 1, obj, bg)
    img, p2f = F.render_meshes_fragments(packed, R, T, _t(np.array([[0.0, 1.0, 0.0]], np.float32)), params)
    torch.cuda.synchronize()
    assert np.array_equal(p2f.cpu().numpy(), want["pix_to_face"])
    assert np.abs(img.cpu().numpy() - want["images"]).max() <= TOL_PIXEL


def test_prepass_lists_of_very_large_face_counts(oracle):
    """The pre-pass keeps a thread's first 8 faces in registers and the records of later ones in LDS as far as it has
    room; a face's first row-bin hands it a 15-bit rank.  Beyond those limits: (a) 40 960 faces over the image and 98 304
    within a few rows, two views each (the last faces have no LDS record: every bin of theirs takes an atomic in the fill
    pass); (b) one view of 61 440 faces, all nearer than the origin and inside one strip: >= 32 767 ranked faces in one
    counter, so the view gets no strip lists and every strip is scanned by the exact path.  Faces are duplicated
    (coincident: the (z, face index) tie rule decides).  Pixel -> face bit-exact against the oracle.
    (Round 5: the lists are built in one pass into per-strip regions of at most 4 096 entries; a strip that needs more — here
    the ONE strip that holds all 61 440 faces of case (b), and the strips of case (a)'s 98 304-face mesh — is scanned by the
    exact path, the other strips of the view keep their lists.)"""
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    v, f = syn.synthetic_mesh("L", 3)
    size = 64
    obj = bg = (0.99999,) * 3
    light = _t(np.array([[0.0, 1.0, 0.0]], np.float32))
    f5 = np.concatenate([f, f[:, [1, 2, 0]], f[::-1], f[:, [2, 0, 1]], f])
    vs, _ = syn.synthetic_mesh("S", 4)
    cases = [([v, (v * 0.08 + np.array([0.0, 0.2, 0.0])).astype(np.float32)], [np.concatenate([f, f[::-1]])[:40960], f5[:98304]],
              _views(2, 2, 31), None),
             # camera on the +z axis; the object sits between it and the origin, ~6 rows tall, inside the strip of rows 16-23
             ([(v[:1500] * 0.08 + np.array([0.0, 0.36, 0.6])).astype(np.float32)], [np.random.default_rng(5).integers(0, 1500, (61440, 3))],
              (np.zeros((1, 1), np.float32), np.zeros((1, 1), np.float32), np.full((1, 1), 2.2, np.float32)), 1)]
    for vl, fl, (az, el, di), want_redo in cases:
        B, V = az.shape
        want = oracle.render_meshes(vl, fl, az, el, di, size, obj_color=obj, bg_color=bg, mode=1)
        packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(x) for x in vl], [torch.from_numpy(np.ascontiguousarray(x)) for x in fl]), _dev())
        R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
        for exact in (False, True):
            params = F.mesh_params(V, size, 1, obj, bg, exact_raster=exact)
            img, p2f, redo = F.render_meshes_fragments(packed, R, T, light, params, return_redo=True)
            torch.cuda.synchronize()
            assert np.array_equal(p2f.cpu().numpy(), want["pix_to_face"]), exact
            assert np.abs(img.cpu().numpy() - want["images"]).max() <= TOL_PIXEL
            if want_redo is not None and not exact:
                assert int(redo) == want_redo, int(redo)   # the one overflowed strip went to the exact path (which scans for it)
        assert all((want["pix_to_face"][b * V:(b + 1) * V] >= 0).sum() > 0 for b in range(B))


def test_full_size_c4_all_clouds_against_the_oracle(oracle):
    """BASELINE configs[3] at FULL size: 20 clouds x 2048 points x 20 views at 224^2 against the oracle."""
    from mvtn_amd import functional as F
    B, P, V, size = 20, 2048, 20, 224
    clouds = []
    for i in range(B):
        v, f = syn.synthetic_mesh("M", i)
        clouds.append(syn.surface_points(v, f, P, seed=i))
    pts = np.stack(clouds)
    az, el, _ = _views(B, V, 0, jitter=False)
    di = np.full((B, V), 1.0, np.float32)
    obj = bg = (0.99999,) * 3
    want = oracle.render_points(pts, az, el, di, size, radius=0.006, obj_color=obj, bg_color=bg)
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.points_params(B, P, V, size, 0.006, obj, bg)
    img, p2p = F.render_points_fragments(_t(pts), None, R, T, _t(di.reshape(-1)), params)
    torch.cuda.synchronize()
    assert np.array_equal(p2p.cpu().numpy().reshape(want["idx"].shape), want["idx"])
    assert np.abs(img.cpu().numpy() - want["images"]).max() <= 1e-6


def test_random_object_colour_in_eval_is_pure_white(oracle):
    """models/renderer.py:156-157: object_color="random" in eval() renders torch_color("white") WITHOUT max_lightness,
    i.e. (1, 1, 1) — not the 0.99999 of the named colours."""
    from mvtn_amd import MVRenderer, Meshes
    v, f = syn.synthetic_mesh("S", 1)
    az, el, di = syn.circular_views(4)
    r = MVRenderer(4, 64, pc_rendering=False, object_color="random", light_direction="fixed").eval()
    img, _ = r([Meshes([v], [f])], None, _t(az[None]), _t(el[None]), _t(di[None]))
    want = oracle.render_meshes([v], [f], az[None], el[None], di[None], 64, obj_color=(1.0, 1.0, 1.0), bg_color=(0.99999,) * 3)
    assert np.abs(img.cpu().numpy() - want["images"]).max() <= 2e-5     # (1e-5 apart from the 0.99999 rendering: must be the 1.0 one)
    wrong = oracle.render_meshes([v], [f], az[None], el[None], di[None], 64, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3)
    cov = want["pix_to_face"] >= 0
    d_right = np.abs(img.cpu().numpy() - want["images"]).reshape(4, 3, 64, 64).max(1)[cov].mean()
    d_wrong = np.abs(img.cpu().numpy() - wrong["images"]).reshape(4, 3, 64, 64).max(1)[cov].mean()
    assert d_right < d_wrong


def test_custom_vertex_colours_with_a_packed_batch(oracle):
    """object_color="custom" + a pre-packed PackedMeshes: the per-vertex colours must reach the kernel (they
    were dropped and the image came out white); same image as handing the list of meshes."""
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes
    vl, fl = _mesh_batch([("S", 0), ("S", 1)])
    B, V, size = 2, 3, 64
    rng = np.random.default_rng(3)
    mx = max(len(v) for v in vl)
    col = rng.uniform(0.1, 0.9, (B, mx, 3)).astype(np.float32)
    az, el, di = _views(B, 4, 2)
    az, el, di = az[:, :V], el[:, :V], di[:, :V]
    r = MVRenderer(V, size, pc_rendering=False, object_color="custom", light_direction="fixed").eval()
    meshes = [Meshes([v], [f]) for v, f in zip(vl, fl)]
    a, e, d, c = _t(az), _t(el), _t(di), _t(col)
    img_list, _ = r(meshes, None, a, e, d, c)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    img_packed, _ = r(packed, None, a, e, d, c)
    assert torch.equal(img_list, img_packed)
    vc = np.concatenate([col[b, :len(v)] for b, v in enumerate(vl)], 0)
    want = oracle.render_meshes(vl, fl, az, el, di, size, vert_colors=vc, bg_color=(0.99999,) * 3)
    assert np.abs(img_packed.cpu().numpy() - want["images"]).max() <= TOL_PIXEL
    assert float((img_packed - 0.99999).abs().max()) > 0.05      # not a white render


@pytest.mark.parametrize("exact", [False, True])
def test_z_clip_cuts_faces_like_the_reference(oracle, exact):
    """PyTorch3D >= 0.5 clips faces at view depth znear / 2 = 0.5 (SURVEY A.9 ii; active as soon as transform_distance
    lets dist drop towards 1.1, models/mvtn.py:33,68,109): faces wholly nearer than the plane disappear, faces that
    straddle it are cut into one or two triangles.  Against the oracle's clip_faces restatement: pixel -> face map
    bit-exact and images within tolerance at dist 2.2 (nothing to cut), 1.1 and 0.9 (camera inside the bounding sphere:
    vertices behind the camera, faces cut on both sides), with and without back-face culling, through the default
    forward (views with cut faces fall back to the exact kernel per strip) and the exact-only kernel; the count of cut
    faces equals the oracle's; z_clip = 0 switches the rule off."""
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    vl, fl = _mesh_batch([("M", 0), ("S", 4), ("soup", 2)])
    B, V, size = 3, 4, 96
    az, el, _ = _views(B, 4, 3)
    obj, bg = (0.9, 0.8, 0.7), (0.1, 0.2, 0.3)
    light = _t(np.array([[0.0, 1.0, 0.0]], np.float32))
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    for dist, cull, expect_cut in ((2.2, False, False), (1.1, False, True), (1.1, True, True), (0.9, False, True)):
        di = np.full((B, V), dist, np.float32)
        R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
        params = F.mesh_params(V, size, 1, obj, bg, cull_backfaces=cull, exact_raster=exact)   # z_clip = None -> 0.5 (perspective_correct)
        assert abs(params.z_clip - 0.5) < 1e-12
        img, p2f, ns = F.render_meshes_fragments(packed, R, T, light, params, return_straddlers=True)
        want = oracle.render_meshes(vl, fl, az, el, di, size, obj_color=obj, bg_color=bg, z_clip=0.5, cull_backfaces=cull, mode=1)
        assert int(ns.item()) == int(want["straddlers"].sum())
        assert (int(ns.item()) > 0) == expect_cut
        p2f, img = p2f.cpu().numpy(), img.cpu().numpy()
        assert np.array_equal(p2f, want["pix_to_face"]), (dist, cull, int((p2f != want["pix_to_face"]).sum()))
        well = _well_conditioned_pixels(vl, fl, want["pix_to_face"], V)
        diff = np.abs(img - want["images"]).reshape(B * V, 3, size, size).max(1)
        assert float(diff[well].max()) <= TOL_PIXEL, (dist, cull, float(diff[well].max()))
        if expect_cut:   # and the rule can be switched off
            off = F.mesh_params(V, size, 1, obj, bg, cull_backfaces=cull, z_clip=0.0, exact_raster=exact)
            img0, p2f0, ns0 = F.render_meshes_fragments(packed, R, T, light, off, return_straddlers=True)
            legacy = oracle.render_meshes(vl, fl, az, el, di, size, obj_color=obj, bg_color=bg, z_clip=0.0, cull_backfaces=cull, mode=1)
            assert int(ns0.item()) == 0 and np.array_equal(p2f0.cpu().numpy(), legacy["pix_to_face"])
            assert not np.array_equal(legacy["pix_to_face"], want["pix_to_face"])


def test_regularize_views_reproduces_the_reference_fixture():
    """tests/golden/regularize_views.npz was written by the REFERENCE's regualarize_rendered_views (ops.py:168-176) on the
    CPU; with dropout off every draw comes from the CPU generator, so the product (fused HIP pass, draws re-indexed from
    the reference's view-major order) must reproduce those outputs bit for bit under the same seed."""
    import os
    from mvtn_amd import regularize as rg
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "regularize_views.npz"))
    n = 0
    for name in fx["names"]:
        B, V, C, H, p, aug, cr, seed = fx["cfg_" + str(name)]
        if p > 0:
            continue      # the keep mask is drawn on the GPU generator in the product (on the CPU in the fixture)
        x = torch.from_numpy(fx["x_" + str(name)]).to(_dev())
        torch.manual_seed(int(seed))
        got = rg.regualarize_rendered_views(x, float(p), bool(aug), float(cr))
        assert np.array_equal(got.cpu().numpy(), fx["y_" + str(name)]), name
        n += 1
    assert n >= 2


def test_hip_path_against_the_torch_restatement_of_c1():
    """The HIP path against the second, torch-made implementation of BASELINE configs[0] directly (no C oracle in between;
    tests/golden/torch_restatement_c1.npz): pixel -> face identical up to a handful of edge pixels, colours within 1e-4."""
    import os
    from mvtn_amd import MVRenderer, Meshes, functional as F
    from mvtn_amd.structures import PackedMeshes
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "torch_restatement_c1.npz"))
    v, f = syn.synthetic_mesh(str(fx["mesh"][0]), int(fx["mesh"][1]))
    S, V = int(fx["image_size"]), 4
    packed = PackedMeshes.from_meshes(Meshes([v], [f]), _dev())
    R, T = F.look_at_view_transform(_t(fx["dist"]), _t(fx["elev"]), _t(fx["azim"]))
    assert np.abs(R.cpu().numpy() - fx["R"]).max() <= 2.5e-7
    params = F.mesh_params(V, S, 1, (0.99999,) * 3, (0.99999,) * 3)
    img, p2f = F.render_meshes_fragments(packed, R, T, _t(np.array([[0.0, 1.0, 0.0]], np.float32)), params)
    differ = p2f.cpu().numpy() != fx["pix_to_face"].astype(np.int32)
    assert differ.sum() <= 8
    assert np.abs(img.cpu().numpy()[0] - fx["images"]).max(1)[~differ].max() <= TOL_PIXEL
    r = MVRenderer(V, S, pc_rendering=False, light_direction="fixed").eval()
    img2, _ = r([Meshes([v], [f])], None, _t(fx["azim"][None]), _t(fx["elev"][None]), _t(fx["dist"][None]))
    assert torch.equal(img2, img)


@pytest.mark.parametrize("p,augment", [(0.3, True), (0.0, False), (0.5, False)])
def test_regularize_normalize_views_fused_equals_chained(p, augment):
    """Row f3, second half: regualarize_rendered_views -> mvctosvc (ops.py:283) -> Normalize
    (viewGCN/tools/Trainer_mvt.py:41-49) as ONE pass: forward bit-identical to the three steps chained with torch ops on
    the fused regulariser's output; gradients bit-identical on interior pixels, 1e-5 on the replicated border sums."""
    from mvtn_amd import regularize as rg
    dev = _dev()
    B, V, C, H = 2, 3, 3, 64
    g = torch.Generator().manual_seed(7)
    x0 = torch.rand(B, V, C, H, H, generator=g).to(dev)
    G = torch.randn(B * V, C, H, H, generator=g).to(dev)
    mean = torch.tensor(rg.IMAGENET_MEAN, device=dev).view(1, 3, 1, 1)
    std = torch.tensor(rg.IMAGENET_STD, device=dev).view(1, 3, 1, 1)
    xa = x0.clone().requires_grad_(True)
    torch.manual_seed(3)
    chained = (rg.mvctosvc(rg.regualarize_rendered_views(xa, p, augment, 0.3)) - mean) / std
    chained.backward(G)
    xb = x0.clone().requires_grad_(True)
    torch.manual_seed(3)
    fused = rg.regularize_normalize_views(xb, p, augment, 0.3)
    assert fused.shape == (B * V, C, H, H) and torch.equal(fused, chained.detach())
    fused.backward(G)
    inner = (slice(None),) * 3 + (slice(1, H - 1), slice(1, H - 1))
    assert torch.equal(xa.grad[inner], xb.grad[inner])
    assert (xa.grad - xb.grad).abs().max().item() <= 1e-5 * max(1.0, xa.grad.abs().max().item())
    # the stand-alone Normalize of the trainer
    y = rg.normalize_views(rg.mvctosvc(x0))
    assert torch.equal(y, (rg.mvctosvc(x0) - mean) / std)


@pytest.mark.parametrize("cfg,V", [("circular", 4), ("spherical", 12), ("spherical", 20)])
def test_view_selectors_on_the_device_feed_the_renderer(golden, cfg, V):
    """Row f1: the non-learned selectors (models/mvtn.py:13-111) as device modules: `.cuda()` buffers, the reference's
    (B, V) stride-(0, 1) expanded outputs with the golden values, consumed by MVRenderer without a copy to the host —
    same images as handing contiguous tensors; with transform_distance the jitter is drawn on the device."""
    from mvtn_amd import MVRenderer, Meshes
    from mvtn_amd.views import view_selector
    B, size = 3, 64
    sel = view_selector(cfg, V, canonical_elevation=30.0, canonical_distance=2.2).to(_dev()).eval()
    az, el, di = sel(None, c_batch_size=B)
    g = golden["selectors"]["%s_%d" % (cfg, V)]
    assert az.is_cuda and tuple(az.shape) == (B, V) and list(az.stride()) == g["stride"]
    assert np.array_equal(az[0].cpu().numpy(), np.array(g["azim"], np.float32))
    assert np.array_equal(el[0].cpu().numpy(), np.array(g["elev"], np.float32))
    assert np.array_equal(di[1].cpu().numpy(), np.array(g["dist"], np.float32))
    vl, fl = _mesh_batch([("S", 0), ("S", 1), ("S", 2)])
    meshes = [Meshes([v], [f]) for v, f in zip(vl, fl)]
    r = MVRenderer(V, size, pc_rendering=False, light_direction="relative").eval()
    with torch.no_grad():
        img, _ = r(meshes, None, az, el, di)
        img2, _ = r(meshes, None, az.contiguous(), el.contiguous(), di.contiguous())
    assert torch.equal(img, img2) and float((img - 0.99999).abs().max()) > 0.1
    jit = view_selector(cfg, V, canonical_distance=2.2, transform_distance=True).to(_dev()).train()
    torch.manual_seed(0)
    _, _, dj = jit(None, c_batch_size=B)
    assert dj.is_cuda and dj.is_contiguous() and float(dj.min()) >= 1.1 - 1e-6 and float(dj.max()) <= 3.3 + 1e-6 and float(dj.std()) > 0.1
    with torch.no_grad():
        imgj, _ = r(meshes, None, az, el, dj)      # distances down to 1.1: the z-clip cull may act, nothing may crash
    assert torch.isfinite(imgj).all()


# ---- round 2: the soft mesh path -----------------------------------------------------------------------

def _soft_blur(sigma):
    return float(np.log(1.0 / 1e-4 - 1.0) * sigma)      # the PyTorch3D tutorials' choice: prob 1e-4 at the rim


@pytest.mark.parametrize("specs,V,size,K,shader,sigma,persp,clipb", [
    ([("S", 0)], 4, 64, 4, "soft_phong", 1e-4, True, None),
    ([("M", 1), ("S", 2)], 3, 96, 8, "soft_phong", 1e-4, True, None),
    ([("S", 3)], 2, 64, 6, "soft_silhouette", 1e-3, True, None),
    ([("soup", 5), ("S", 6)], 2, 48, 5, "soft_phong", 3e-4, False, False),
    ([("S", 7)], 2, 37, 1, "soft_phong", 1e-4, True, None),
])
def test_soft_mesh_forward_parity(oracle, specs, V, size, K, shader, sigma, persp, clipb):
    """blur_radius > 0, faces_per_pixel = K, SoftPhongShader / SoftSilhouetteShader against the oracle: the K-nearest
    fragment lists bit-identical, RGBA within 1e-4."""
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    vl, fl = _mesh_batch(specs)
    B = len(vl)
    az, el, di = _views(B, 4, 21)
    az, el, di = az[:, :V], el[:, :V], di[:, :V]
    blur = _soft_blur(sigma)
    obj, bg = (0.9, 0.8, 0.7), (0.1, 0.2, 0.3)
    want = oracle.render_meshes_soft(vl, fl, az, el, di, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, obj_color=obj, bg_color=bg,
                                     perspective_correct=persp, clip_barycentric_coords=(blur > 0) if clipb is None else clipb)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.mesh_params(V, size, 1, obj, bg, perspective_correct=persp, z_clip=0.0)
    soft = F.soft_params(blur, K, shader, sigma, 1e-4, clipb)
    img, p2f = F.render_meshes_soft(packed, R, T, _t([[0.0, 1.0, 0.0]]), params, soft, return_fragments=True)
    torch.cuda.synchronize()
    p2f, img = p2f.cpu().numpy(), img.cpu().numpy()
    n_diff = int((p2f != want["pix_to_face"]).sum())
    well = _well_conditioned_pixels(vl, fl, want["pix_to_face"][..., 0], V)
    for k in range(1, K):
        well &= _well_conditioned_pixels(vl, fl, want["pix_to_face"][..., k], V)
    diff = np.abs(img - want["images"]).reshape(B * V, 4, size, size).max(1)
    print("fragment mismatches %d of %d, max|d| %.3g, first slot filled %.3f, last slot filled %.3f" % (
        n_diff, p2f.size, float(diff[well].max()), (p2f[..., 0] >= 0).mean(), (p2f[..., -1] >= 0).mean()))
    assert n_diff == 0
    assert float(diff[well].max()) <= TOL_PIXEL


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


@pytest.mark.parametrize("shader,light,sigma,K", [("soft_phong", "relative", 1e-4, 4), ("soft_silhouette", "fixed", 1e-3, 5),
                                                   ("soft_phong", "fixed", 1e-3, 3)])
def test_soft_mesh_backward_against_finite_differences(oracle, shader, light, sigma, K):
    """Gradients of L = sum(G * RGBA) w.r.t. (azim, elev, dist) through MVRenderer(shader=...) — colours, the depth
    softmax AND the coverage probabilities (signed distances: the gradient a hard rasteriser does not have) — against
    fp64 central differences of the oracle with the fragment selection held fixed."""
    from mvtn_amd import MVRenderer, Meshes
    vl, fl = _mesh_batch([("S", 0), ("S", 4)])
    B, V, size = 2, 2, 48
    az, el, di = _views(B, 4, 33)
    az, el, di = az[:, :V], el[:, :V], di[:, :V]
    blur = _soft_blur(sigma)
    r = MVRenderer(V, size, pc_rendering=False, light_direction=light, faces_per_pixel=K, shader=shader, blur_radius=blur,
                   sigma=sigma, gamma=1e-4, return_alpha=True, z_clip=0.0).eval()
    leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
    rgba, _ = r([Meshes([v], [f]) for v, f in zip(vl, fl)], None, *leaves)
    assert rgba.shape == (B, V, 4, size, size)
    relative = light == "relative"
    ldir = oracle.camera_position(az, el, di) if relative else np.array([[0.0, 1.0, 0.0]], np.float32)
    want = oracle.render_meshes_soft(vl, fl, az, el, di, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, light_dir=ldir,
                                     obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3)
    assert np.abs(rgba.detach().cpu().numpy() - want["images"]).max() <= TOL_PIXEL
    rng = np.random.default_rng(5)
    G = rng.normal(size=rgba.shape)
    if shader == "soft_silhouette":
        G[:, :, :3] = 0.0          # the silhouette loss lives in the alpha channel
    rgba.backward(_t(G))
    fd = _fd_grad_soft(oracle, vl, fl, az, el, di, size, G, want["pix_to_face"], relative, blur, K, shader, sigma)
    for name, leaf, ref in zip(("azim", "elev", "dist"), leaves, fd):
        got = leaf.grad.cpu().numpy().astype(np.float64)
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        print(name, "rel err %.3g" % rel, "max |grad| %.3g" % np.abs(ref).max())
        assert rel < 5e-3, (name, rel, got, ref)
    rgb_only = MVRenderer(V, size, pc_rendering=False, light_direction=light, faces_per_pixel=K, shader=shader, blur_radius=blur,
                          sigma=sigma, z_clip=0.0).eval()
    with torch.no_grad():
        img3, _ = rgb_only([Meshes([v], [f]) for v, f in zip(vl, fl)], None, *[x.detach() for x in leaves])
    assert img3.shape == (B, V, 3, size, size) and torch.equal(img3, rgba.detach()[:, :, :3])
