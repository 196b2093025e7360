"""-m gpu, round 3: files -> `io.mesh_from_file` -> `MVRenderer` against the oracle, rendered views ->
`save_grid` -> PNG decode against the tensor (row f2)."""
import os

import numpy as np
import pytest
import torch

from mvtn_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL_PIXEL = 1e-4


def _dev():
    assert torch.cuda.is_available(), "GPU test needs a GPU"
    return torch.device("cuda:0")


def _t(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(_dev())


def _views(B, V, seed):
    az, el, di = syn.spherical_views(V)
    rng = np.random.default_rng(seed)
    az = np.tile(az, (B, 1)) + rng.uniform(-5, 5, (B, V))
    el = np.tile(el, (B, 1)) + rng.uniform(-5, 5, (B, V))
    di = np.tile(di, (B, 1)) + rng.uniform(-0.1, 0.1, (B, V))
    return az.astype(np.float32), el.astype(np.float32), di.astype(np.float32)


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


@pytest.mark.parametrize("fmt", ["off", "off_glued", "obj"])
def test_mesh_files_through_the_renderer_against_the_oracle(oracle, tmp_path, fmt):
    """SURVEY 8 f5 on the GPU: a mesh FILE -> `io.mesh_from_file` (load, the dataset's -90 degree rotation about x, centre +
    normalise: custom_dataset.py:100-132) -> `MVRenderer` equals the oracle rendering the arrays the loader produced, and
    the loader's arrays equal the same operations applied to the arrays that were written."""
    from mvtn_amd import MVRenderer, io
    v, f = syn.synthetic_mesh("S", 3)
    v = (v * np.array([1.3, 0.8, 1.0], np.float32) + np.array([0.4, -0.2, 0.1], np.float32)).astype(np.float32)
    path = str(tmp_path / ("m." + ("obj" if fmt == "obj" else "off")))
    if fmt == "obj":
        _write_obj(path, v, f)
    else:
        _write_off(path, v, f, glued=fmt == "off_glued")
    mesh = io.mesh_from_file(path)
    lv, lf = mesh.verts_list()[0], mesh.faces_list()[0]
    assert lf.dtype == torch.int64 and np.array_equal(lf.numpy(), f)
    # the reference's chain on the written values (%.9g round-trips a float32): rotate (float64), to float32, centre, normalise
    v_text = np.array([[float("%.9g" % x) for x in p_] for p_ in v], np.float64)     # what a text parser reads back
    assert np.array_equal(v_text.astype(np.float32), v)
    vv = io.rotation_matrix((1, 0, 0), -90.0).dot(v_text.T).T
    want_v = io.torch_center_and_normalize(torch.from_numpy(vv).to(torch.float), p=2)
    assert torch.equal(lv, want_v)
    assert abs(float(torch.linalg.norm(lv, dim=1).max()) - 1.0) < 1e-6 and float(lv.mean(0).abs().max()) < 1e-6
    V, size = 4, 96
    az, el, di = _views(1, V, 5)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="fixed").eval()
    with torch.no_grad():
        img, _ = r([mesh], None, _t(az), _t(el), _t(di))
    want = oracle.render_meshes([lv.numpy()], [lf.numpy()], az, el, di, size, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3)
    err = float(np.abs(img.cpu().numpy() - want["images"]).max())
    assert err <= TOL_PIXEL, err
    assert float(img.min()) < 0.9


def test_save_grid_of_rendered_views_decodes_to_the_tensor(tmp_path):
    """SURVEY 8 f2 on the GPU: `render_and_save` (models/renderer.py:200-207) writes the first object's views as one row
    (util.py:667-673: make_grid + image writer); the decoded PNG equals make_grid of the rendered tensor to 1/255."""
    from test_viz import _decode_png
    from mvtn_amd import MVRenderer, Meshes, viz
    dev = _dev()
    v, f = syn.synthetic_mesh("M", 2)
    V, size = 6, 64
    az, el, di = _views(1, V, 9)
    r = MVRenderer(V, size, pc_rendering=False, object_color="red", background_color="white", light_direction="fixed").eval()
    gpath, cpath = str(tmp_path / "views.png"), str(tmp_path / "cams.png")
    r.render_and_save([Meshes([v], [f])], None, _t(az), _t(el), _t(di), gpath, cpath)
    with torch.no_grad():
        img, _ = r([Meshes([v], [f])], None, _t(az), _t(el), _t(di))
    px = _decode_png(gpath).astype(np.float32) / 255.0
    grid = viz.make_grid(img[0].cpu(), nrow=V).clamp(0, 1).permute(1, 2, 0).numpy()
    assert px.shape == grid.shape == (size + 4, V * (size + 2) + 2, 3)
    assert float(np.abs(px - grid).max()) <= 1.0 / 255.0 + 1e-6
    # the frames are where make_grid puts them, and they are not blank
    cell = px[2:2 + size, 2:2 + size]
    assert float(np.abs(cell - img[0, 0].permute(1, 2, 0).cpu().numpy().clip(0, 1)).max()) <= 1.0 / 255.0 + 1e-6
    assert float(cell.min()) < 0.7 and os.path.getsize(cpath) > 1000


# ---- the second, torch-made implementation (tests/golden/make_torch_restatement_r3.py) against the HIP path directly ---------

def _fx(name):
    return np.load(os.path.join(os.path.dirname(__file__), "golden", name))


@pytest.mark.parametrize("K", [1, 4])
def test_hip_point_path_against_the_torch_restatement(K):
    """models/renderer.py:116-151 as torch ops (2 clouds x 20 views of BASELINE configs[3]) vs the HIP point path through
    MVRenderer: every slot of every pixel identical; colours within 1e-6 except where the compositor's 1e-4 clamp is active
    (sum of weights < 1e-4: a handful of rim pixels whose colour w f / 1e-4 multiplies a 1-ulp coordinate difference by 1e4 —
    see tests/test_oracle.py)."""
    from mvtn_amd import MVRenderer, functional as F
    fx = _fx("torch_restatement_points.npz")
    S, r, P = int(fx["image_size"]), float(fx["radius"]), int(fx["n_points"])
    pts = []
    for seed in fx["seeds"]:
        v, f = syn.synthetic_mesh("M", int(seed))
        pts.append(syn.surface_points(v, f, P, seed=int(seed)))
    pts = np.stack(pts)
    az, el, di = (np.tile(fx[k], (2, 1)) for k in ("azim", "elev", "dist"))
    rr = MVRenderer(20, S, pc_rendering=True, object_color="white", background_color="white", points_radius=r, points_per_pixel=K).eval()
    with torch.no_grad():
        img, cams = rr(None, _t(pts), _t(az), _t(el), _t(di))
    assert np.abs(cams.R.cpu().numpy() - np.tile(fx["R"], (2, 1, 1))).max() <= 2.5e-7
    params = F.points_params(2, P, 20, S, r, (0.99999,) * 3, (0.99999,) * 3, points_per_pixel=K)
    img2, idx = F.render_points_fragments(_t(pts), None, cams.R, cams.T, _t(di.reshape(-1)), params)
    assert torch.equal(img2, img)
    idx = idx.cpu().numpy().reshape(40, S, S, K)
    want = fx["idx_k%d" % K].astype(np.int32)
    differ = (idx != want).any(-1)
    assert differ.sum() == 0, "slot lists differ at %d pixels" % differ.sum()   # (observed: 0 of 2 007 040 / 8 028 160 slots)
    diff = np.abs(img.cpu().numpy() - fx["images_k%d" % K]).max(2).reshape(differ.shape)
    big = diff > 1e-6
    print("K=%d: slot lists differ at %d pixels; %d pixels above 1e-6 (clamp region), max %.3g" % (K, differ.sum(), big.sum(), diff.max()))
    assert big.sum() <= 16 and diff[~big].max() <= 1e-6
    assert (idx[big][:, 0] >= 0).all()      # the outliers are covered pixels (rim of a disc), never background


def test_hip_face_clipping_against_the_torch_restatement():
    """clip_faces / convert_clipped_rasterization_to_original_faces as torch ops (dist 1.1: 65 and 63 faces cut) vs the HIP
    path: same counts of cut faces, pixel -> face identical at every pixel, colours within 1e-4."""
    from mvtn_amd import Meshes, functional as F
    from mvtn_amd.structures import PackedMeshes
    fx = _fx("torch_restatement_zclip.npz")
    v, f = syn.synthetic_mesh(str(fx["mesh"][0]), int(fx["mesh"][1]))
    S, V = int(fx["image_size"]), len(fx["azim"])
    packed = PackedMeshes.from_meshes(Meshes([v], [f]), _dev())
    R, T = F.look_at_view_transform(_t(fx["dist"]), _t(fx["elev"]), _t(fx["azim"]))
    for exact in (False, True):
        params = F.mesh_params(V, S, 1, (0.99999,) * 3, (0.99999,) * 3, z_clip=float(fx["z_clip"]), exact_raster=exact)
        img, p2f, n_st = F.render_meshes_fragments(packed, R, T, _t(np.array([[0.0, 1.0, 0.0]], np.float32)), params, return_straddlers=True)
        assert int(n_st) == int(fx["straddlers"].sum())
        differ = p2f.cpu().numpy() != fx["pix_to_face"].astype(np.int32)
        assert differ.sum() == 0, differ.sum()   # (observed: 0 of 32 768 pixels)
        assert np.abs(img.cpu().numpy()[0] - fx["images"]).max(1)[~differ].max() <= TOL_PIXEL


def test_hip_soft_phong_against_the_torch_restatement():
    """SoftPhongShader with faces_per_pixel 4 as torch ops (phong_shading of every fragment + softmax_rgb_blend) vs the HIP soft
    path: fragment lists identical (ties ordered by face index on both sides), RGBA within 1e-4."""
    from mvtn_amd import Meshes, functional as F
    from mvtn_amd.structures import PackedMeshes
    fx = _fx("torch_restatement_soft.npz")
    v, f = syn.synthetic_mesh(str(fx["mesh"][0]), int(fx["mesh"][1]))
    S, V, K = int(fx["image_size"]), len(fx["azim"]), int(fx["faces_per_pixel"])
    packed = PackedMeshes.from_meshes(Meshes([v], [f]), _dev())
    R, T = F.look_at_view_transform(_t(fx["dist"]), _t(fx["elev"]), _t(fx["azim"]))
    params = F.mesh_params(V, S, 1, (0.99999,) * 3, (0.99999,) * 3)
    soft = F.soft_params(float(fx["blur_radius"]), K, "soft_phong", float(fx["sigma"]), float(fx["gamma"]))
    img, p2f = F.render_meshes_soft(packed, R, T, _t([[0.0, 1.0, 0.0]]), params, soft, return_fragments=True)
    differ = (p2f.cpu().numpy().reshape(V, S, S, K) != fx["pix_to_face"].astype(np.int32)).any(-1)
    assert differ.sum() == 0, differ.sum()   # (observed: 0 of 32 768 fragment slots)
    diff = np.abs(img.cpu().numpy().reshape(V, 4, S, S) - fx["images"]).max(1)
    assert diff[~differ].max() <= TOL_PIXEL


def test_library_switches_follow_the_oracle_switches(oracle):
    """MVRenderer(fov_scale=..., normals_three_cross=True) (mvr_mesh_params.fov_scale, MVR_NORMALS_THREE_CROSS) against the
    oracle with the same switches: pixel -> face maps bit-exact, colours within 1e-4; and each switch changes something."""
    from mvtn_amd import MVRenderer, Meshes
    v, f = syn.synthetic_mesh("M", 4)
    V, size = 4, 96
    az, el, di = _views(1, V, 3)
    s1 = float.fromhex("0x1.bb67b0p+0")
    kw = dict(obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3)
    outs = {}
    for name, three, s in (("default", True, 0.0), ("one_cross", False, 0.0), ("fov_scale", True, s1)):
        try:
            oracle.set_options(three, s)
            want = oracle.render_meshes([v], [f], az, el, di, size, **kw)
        finally:
            oracle.set_options()
        r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="fixed",
                       fov_scale=s, normals_three_cross=three).eval()
        a, e, d = (_t(x).requires_grad_(True) for x in (az, el, di))
        img, _ = r([Meshes([v], [f])], None, a, e, d)
        err = float(np.abs(img.detach().cpu().numpy() - want["images"]).max())
        assert err <= TOL_PIXEL, (name, err)
        img.sum().backward()
        assert torch.isfinite(a.grad).all() and float(a.grad.abs().max()) > 0
        outs[name] = (img.detach().cpu().numpy(), want["images"])
    assert 0 < np.abs(outs["one_cross"][0] - outs["default"][0]).max() < 1e-5      # normals: colours move in the 7th digit
    assert np.abs(outs["fov_scale"][0] - outs["default"][0]).max() > 0                # s one ulp larger: the image changes
    # the HIP path follows the oracle THROUGH the switch (not merely both within tolerance of the default)
    d_hip, d_orc = outs["one_cross"][0] - outs["default"][0], outs["one_cross"][1] - outs["default"][1]
    assert np.abs(d_hip - d_orc).max() < 5e-6


@pytest.mark.parametrize("shader,K,dist", [("soft_phong", 4, 1.1), ("soft_phong", 2, 0.9), ("soft_silhouette", 3, 1.1)])
def test_soft_path_cuts_straddling_faces(oracle, shader, K, dist):
    """Round 3: the soft path cuts faces at the z-clip plane like the hard path (PyTorch3D >= 0.5 clip_faces; the pieces are
    separate candidates of the K-nearest lists and map back to the same face).  Camera inside the bounding sphere (dist 1.1,
    0.9): fragment lists (face indices; a face may appear twice in a pixel) identical to the oracle's, RGBA within 1e-4;
    the gradients are finite and repeat bit for bit."""
    from mvtn_amd import MVRenderer, Meshes, functional as F
    from mvtn_amd.structures import PackedMeshes
    v, f = syn.synthetic_mesh("S", 7)
    V, size, sigma = 3, 64, 1e-4
    az, el, _ = _views(1, V, 13)
    di = np.full((1, V), dist, np.float32)
    blur = float(np.log(1.0 / 1e-4 - 1.0) * sigma)
    obj, bg = (0.9, 0.8, 0.7), (0.1, 0.2, 0.3)
    want = oracle.render_meshes_soft([v], [f], az, el, di, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, obj_color=obj, bg_color=bg, z_clip=0.5)
    whole = oracle.render_meshes_soft([v], [f], az, el, di, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, obj_color=obj, bg_color=bg, z_clip=0.0)
    assert (want["pix_to_face"] != whole["pix_to_face"]).mean() > 0.01, "the case must depend on the cut"
    packed = PackedMeshes.from_meshes(Meshes([v], [f]), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    params = F.mesh_params(V, size, 1, obj, bg, z_clip=0.5)
    soft = F.soft_params(blur, K, shader, sigma, 1e-4)
    img, p2f = F.render_meshes_soft(packed, R, T, _t([[0.0, 1.0, 0.0]]), params, soft, return_fragments=True)
    torch.cuda.synchronize()
    p2f, img = p2f.cpu().numpy().reshape(V, size, size, K), img.cpu().numpy().reshape(V, 4, size, size)
    n_diff = int((p2f != want["pix_to_face"]).sum())
    twice = int(((p2f[..., 1:] == p2f[..., :-1]) & (p2f[..., 1:] >= 0)).sum())
    diff = np.abs(img - want["images"].reshape(V, 4, size, size)).max(1)
    print("fragment mismatches %d of %d; a face twice in a pixel's list: %d times; max|d| %.3g" % (n_diff, p2f.size, twice, diff.max()))
    assert n_diff == 0
    assert diff.max() <= TOL_PIXEL
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="black", light_direction="fixed", shader=shader,
                   blur_radius=blur, faces_per_pixel=K, sigma=sigma, return_alpha=True).eval()
    grads = []
    for _ in range(2):
        a, e, d = (_t(x).requires_grad_(True) for x in (az, el, di))
        out, _ = r([Meshes([v], [f])], None, a, e, d)
        out.sum().backward()
        grads.append([x.grad.clone() for x in (a, e, d)])
    assert all(torch.isfinite(g).all() for g in grads[0]) and float(grads[0][0].abs().max()) > 0
    assert all(torch.equal(g0, g1) for g0, g1 in zip(*grads))


def test_full_c3_object_gradients_against_finite_differences(oracle):
    """One whole object of BASELINE configs[2] — tier M, all 12 jittered views at 224 x 224, relative light — against fp64
    central differences of the oracle with the pixel -> face maps held fixed (Weak 9).  The views are
    independent, so ONE pair of oracle renders per parameter gives all 12 derivatives: every view's parameter moves by +-h
    and the loss is taken per view (6 renders of 12 views instead of 72)."""
    from mvtn_amd import MVRenderer, Meshes
    v, f = syn.synthetic_mesh("M", 31)
    V, size = 12, 224
    az, el, di = _views(1, V, 32)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative").eval()
    leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
    images, _ = r([Meshes([v], [f])], None, *leaves)
    kw = dict(obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3)
    want = oracle.render_meshes([v], [f], az, el, di, size, light_dir=oracle.camera_position(az, el, di), **kw)
    assert np.abs(images.detach().cpu().numpy() - want["images"]).max() <= TOL_PIXEL
    G = np.random.default_rng(33).normal(size=images.shape)
    images.backward(_t(G))
    v64 = v.astype(np.float64)
    base = [x.astype(np.float64) for x in (az, el, di)]

    def per_view_loss(a, e, d):
        o = oracle.render_meshes([v64], [f], a, e, d, size, light_dir=oracle.camera_position(a, e, d, np.float64), dtype=np.float64,
                                 fixed_p2f=want["pix_to_face"], **kw)
        return (o["images"] * G).sum(axis=(0, 2, 3, 4))          # (V,)
    for k, (name, h) in enumerate((("azim", 1e-4), ("elev", 1e-4), ("dist", 1e-5))):
        plus, minus = [x.copy() for x in base], [x.copy() for x in base]
        plus[k] += h
        minus[k] -= h
        ref = (per_view_loss(*plus) - per_view_loss(*minus)) / (2 * h)
        got = leaves[k].grad.cpu().numpy().astype(np.float64)[0]
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        print("%s: max relative error over the 12 views %.2e (|grad| up to %.3g)" % (name, rel, np.abs(ref).max()))
        assert rel < 2e-3, (name, rel, got, ref)


@pytest.mark.parametrize("size,strip_rows,tier", [(224, 0, "L"), (224, 16, "L"), (256, 16, "L"), (256, 32, "M"), (100, 5, "L"), (64, 32, "L"), (37, 3, "M")])
def test_fast_rasteriser_lds_layouts_against_the_exact_kernel(size, strip_rows, tier):
    """The fast rasteriser's shading table runs over the key, record and item areas of the workgroup's LDS and the
    covered-pixel list carries 32-bit entries; both are sized from the strip's shape.  Other strip shapes than the
    default 224 x 8 — and dense meshes, whose strips need several table passes — against the exact kernel: identical
    pixel -> face maps, colours within the two kernels' shading tolerance."""
    from mvtn_amd import functional as F
    from mvtn_amd.structures import Meshes, PackedMeshes
    vl, fl = zip(*[syn.synthetic_mesh(tier, 40 + i) for i in range(2)])
    V = 6
    az, el, di = _views(2, V, 11)
    packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    light = _t(np.array([[0.0, 1.0, 0.0]], np.float32))
    out = {}
    for mode in ("fast", "exact"):
        params = F.mesh_params(V, size, 1, (0.9, 0.8, 0.7), (1.0, 1.0, 1.0), exact_raster=(mode == "exact"), strip_rows=strip_rows)
        img, p2f, cnt = F.render_meshes_fragments(packed, R, T, light, params, return_redo=True)
        torch.cuda.synchronize()
        out[mode] = (img.cpu().numpy(), p2f.cpu().numpy(), int(cnt.item()))
    assert (out["exact"][1] >= 0).mean() > 0.1, "the object must cover part of the frame"
    assert np.array_equal(out["fast"][1], out["exact"][1]), "%d pixels differ" % int((out["fast"][1] != out["exact"][1]).sum())
    assert np.abs(out["fast"][0] - out["exact"][0]).max() <= 5e-5
