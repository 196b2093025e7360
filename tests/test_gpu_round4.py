"""-m gpu, round 4: the HIP backward kernels against torch.autograd through the second, torch-made implementation
(tests/golden/make_torch_restatement_grad.py -> torch_restatement_grad.npz; VERDICT r3 missing 1a)."""
import os

import numpy as np
import pytest
import torch

from mvtn_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU test needs a GPU"
    return torch.device("cuda:0")


def _t(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).to(_dev())


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


@pytest.mark.parametrize("gather", [False, True], ids=["table_kernel", "gather_kernel"])
def test_backward_kernels_against_torch_autograd_c1(gather):
    """BASELINE configs[0] (tier M, 4 circular views, 128 x 128, fixed light): both backward kernels against
    torch.autograd through the torch restatement with the pixel -> face map held fixed; rel <= 2e-3 asked, ~1e-5 seen."""
    fx = _fx("torch_restatement_grad.npz")
    v, f = syn.synthetic_mesh("M", 0)
    S = int(fx["c1_image_size"])
    az, el, di = (fx["c1_" + k][None].astype(np.float32) for k in ("azim", "elev", "dist"))
    G = np.random.default_rng(int(fx["c1_g_seed"])).standard_normal((1, 4, 3, S, S)).astype(np.float32)
    grads, p2f = _hip_grads([v], [f], az, el, di, S, G, False, gather)
    assert (p2f != fx["c1_pix_to_face"].astype(np.int32)).sum() == 0      # the same function is being differentiated
    for name, got in zip(("azim", "elev", "dist"), grads):
        ref = fx["c1_grad_" + name].astype(np.float64)[None]
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        print("c1 %s (%s): max relative error %.2e" % (name, "gather" if gather else "table", rel))
        assert rel <= 2e-4, (name, got, ref)


@pytest.mark.parametrize("gather", [False, True], ids=["table_kernel", "gather_kernel"])
def test_backward_kernels_against_torch_autograd_c3(gather):
    """Two objects of BASELINE configs[2] (tier M, 12 jittered views each, 224 x 224, RELATIVE light: the gradient also
    flows through the light direction = camera position, models/renderer.py:167-171) against torch.autograd."""
    fx = _fx("torch_restatement_grad.npz")
    if "c3_grad_azim" not in fx:
        pytest.skip("fixture without the c3 case")
    S = int(fx["c3_image_size"])
    az, el, di = (fx["c3_" + k].astype(np.float32) for k in ("azim", "elev", "dist"))
    vl, fl = zip(*[syn.synthetic_mesh("M", int(s)) for s in fx["c3_mesh_seeds"]])
    G = np.stack([np.random.default_rng(int(s)).standard_normal((12, 3, S, S)).astype(np.float32) for s in fx["c3_g_seeds"]])
    grads, p2f = _hip_grads(list(vl), list(fl), az, el, di, S, G, True, gather)
    differ = p2f.reshape(2, 12, S, S) != fx["c3_pix_to_face"].astype(np.int32)
    # (the torch rasteriser works on bmm-projected vertices: a last-bit difference of a vertex may move an edge pixel; a pixel
    # that changes face changes the function by one pixel's worth.  Observed, and asserted: ONE pixel, in view 11 of object 0)
    assert differ.sum() == 1 and differ[0, 11].sum() == 1, differ.reshape(24, -1).sum(1)
    # a view that holds such a pixel differentiates a (one pixel) different function: its three derivatives are compared with a
    # loose bound, every other view with the tight one
    dirty = differ.reshape(24, -1).any(1).reshape(2, 12)
    for name, got in zip(("azim", "elev", "dist"), grads):
        ref = fx["c3_grad_" + name].astype(np.float64)
        err = np.abs(got - ref) / np.abs(ref).max()
        clean = err[~dirty].max()
        print("c3 %s (%s): max relative error %.2e over the %d views with identical maps (%.2e over the %d with a pixel on another face)" % (
            name, "gather" if gather else "table", clean, int((~dirty).sum()), err[dirty].max() if dirty.any() else 0.0, int(dirty.sum())))
        # fp32 kernels against a float64 derivative: <= 3.3e-3 at one view, <= 1.5e-3 at a second, <= 4e-4 elsewhere.  Round 4
        # blamed relu kinks of the diffuse term; round 5 measured it: NO covered pixel of this case has |n . l| < 1e-5 (the
        # fixture counts them).  What the outliers are is the next test: a few dozen pixels on nearly edge-on faces.
        assert clean <= 5e-3, (name, clean)
        assert (err[~dirty] > 2e-3).sum() <= 2, (name, err)
        assert err.max() <= 3e-2, (name, err.max())


@pytest.mark.parametrize("gather", [False, True], ids=["table_kernel", "gather_kernel"])
def test_backward_outliers_are_the_pixels_of_edge_on_faces(gather):
    """what the 2 of 72 C3 derivatives that miss 2e-3 (previous test) come from.  The fixture holds, per
    threshold t, the bit mask of the pixels whose winning face is badly conditioned — cond = (larger extent of its NDC box)^2 /
    |2 x NDC area| > t: nearly edge-on, the barycentrics' derivative ~ 1 / area — and torch.autograd's gradients of the loss WITHOUT
    those pixels.  With the same mask on the upstream gradient the fp32 kernels agree with float64 autograd to <= 2e-4 on every
    derivative of every view with an identical pixel -> face map once the 40 pixels with cond > 100 (0.008 % of the covered
    ones) are out; with only the 4 pixels of cond > 300 out, 7 derivatives still miss 2e-4; with 1 pixel out, 12.  (And no
    pixel sits on a kink of relu(n . l): `c3_kink_bits` is empty, the masked and unmasked gradients are equal bit for bit.)"""
    fx = _fx("torch_restatement_grad.npz")
    if "c3_cond_bits" not in fx:
        pytest.skip("fixture without the conditioning masks")
    assert np.unpackbits(fx["c3_kink_bits"]).sum() == 0
    for k in ("azim", "elev", "dist"):
        assert np.array_equal(fx["c3_grad_" + k], fx["c3_grad_" + k + "_nokink"])
    S = int(fx["c3_image_size"])
    az, el, di = (fx["c3_" + k].astype(np.float32) for k in ("azim", "elev", "dist"))
    vl, fl = zip(*[syn.synthetic_mesh("M", int(s)) for s in fx["c3_mesh_seeds"]])
    G = np.stack([np.random.default_rng(int(s)).standard_normal((12, 3, S, S)).astype(np.float32) for s in fx["c3_g_seeds"]])
    thr = [float(t) for t in fx["c3_cond_thresholds"]]
    over = {}
    for k, t in enumerate(thr):
        mask = np.unpackbits(fx["c3_cond_bits"][k], axis=-1)[..., :S].astype(bool)      # (2, 12, S, S): cond > t
        grads, p2f = _hip_grads(list(vl), list(fl), az, el, di, S, G * (~mask)[:, :, None], True, gather)
        dirty = (p2f.reshape(2, 12, S, S) != fx["c3_pix_to_face"].astype(np.int32)).reshape(24, -1).any(1).reshape(2, 12)
        err = np.stack([np.abs(got - fx["c3_grad_%s_cond" % name][k].astype(np.float64)) / np.abs(fx["c3_grad_%s_cond" % name][k]).max()
                        for name, got in zip(("azim", "elev", "dist"), grads)])
        over[t] = int((err[:, ~dirty] > 2e-4).sum())
        print("cond > %g masked: %d pixels; derivatives of clean views above 2e-4: %d of %d; max %.2e (dirty view: %.2e)" % (
            t, mask.sum(), over[t], 3 * (~dirty).sum(), err[:, ~dirty].max(), err[:, dirty].max()))
        if t <= 100.0:
            assert over[t] == 0 and err[:, dirty].max() <= 2e-3, (t, err)
    assert over[1000.0] >= over[300.0] > 0     # the fewer of those pixels are masked, the more derivatives miss the bound


@pytest.mark.parametrize("S", [96, 95])
def test_fast_rasteriser_faces_with_tiny_edge_coefficients(S):
    """(S = 95, ADVICE r4: an odd width has a pixel centre at exactly 0, outside fast_coef_tiny's proof — odd widths take the exact
    kernel; the case stays here so that a future fast path for them meets these faces.)  Round 4: the fast rasteriser no longer tests every inside pixel for an edge value in (0, 2^-100); a face with a nonzero
    edge coefficient below 2^-44 — the only kind that can produce one (mvr_mesh_raster_fast.inc: fast_coef_tiny) — is not
    trusted and its pixels are settled by the exact test.  Triangles with an edge that is almost, but not exactly, horizontal or
    vertical (coordinate differences of 1e-20 ... 1e-30 after an axis-aligned camera), in front of and behind ordinary ones:
    pixel -> face maps identical to the oracle's and to the exact kernel's."""
    from mvtn_amd import Meshes, functional as F
    from mvtn_amd.structures import PackedMeshes
    from oracle import oracle
    v = np.array([[-0.5, 0.0, 0.1], [0.5, 1e-20, 0.1], [0.0, 0.6, 0.1],          # nearly horizontal edge (dy = 1e-20)
                  [0.0, -0.5, -0.2], [1e-25, 0.5, -0.2], [0.45, 0.0, -0.2],   # nearly vertical edge (dx = 1e-25)
                  [-0.6, 0.0, 0.3], [0.6, 1e-30, 0.3], [0.1, -0.55, 0.3],   # another one, nearest to the camera
                  [-0.7, -0.7, -0.4], [0.7, -0.7, -0.4], [0.0, 0.7, -0.4]], np.float32)   # an ordinary triangle behind them
    f = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11]], np.int64)
    az, el, di = np.array([[0.0, 180.0]], np.float32), np.array([[0.0, 0.0]], np.float32), np.array([[2.2, 2.2]], np.float32)
    want = oracle.render_meshes([v], [f], az, el, di, S, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3)
    packed = PackedMeshes.from_meshes(Meshes([v], [f]), _dev())
    R, T = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)))
    light = _t([[0.0, 1.0, 0.0]])
    maps = {}
    for mode in ("fast", "exact"):
        params = F.mesh_params(2, S, 1, (0.99999,) * 3, (0.99999,) * 3, exact_raster=(mode == "exact"))
        img, p2f = F.render_meshes_fragments(packed, R, T, light, params)
        maps[mode] = p2f.cpu().numpy()
        assert (maps[mode] != want["pix_to_face"]).sum() == 0, mode
        assert np.abs(img.cpu().numpy() - want["images"]).max() <= 1e-4, mode
    assert (want["pix_to_face"] >= 0).mean() > 0.1      # the triangles are on screen


@pytest.mark.parametrize("zc", [0.5, 0.0])
@pytest.mark.parametrize("shader,K", [("soft_silhouette", 3), ("soft_phong", 4)])
def test_soft_backward_on_cut_faces_against_finite_differences(shader, K, zc):
    """VERDICT r3, missing 3: fragments on faces that the z-clip plane CUTS.  Rounds 2-3 differentiated the whole face (wrong next
    to the clip line, where the piece has an edge of its own that moves with the camera) and checked it against an oracle whose
    fixed-list evaluation ALSO used whole faces.  Round 4: mvr::soft_fragment_bwd_piece differentiates the piece (its vertices on
    the clip plane as functions of the camera), and the oracle evaluates a fixed list on the pieces the free render found
    (`fixed_piece`).  Camera inside the bounding sphere (dist 1.1: dozens of cut faces per view; with z_clip 0 as the control):
    gradients of sum(G * RGBA) against fp64 central differences of the clipping oracle, fragment lists and pieces held fixed."""
    from mvtn_amd import MVRenderer, Meshes
    from oracle import oracle
    v, f = syn.synthetic_mesh("S", 7)
    V, size, sigma = 2, 48, 1e-3 if shader == "soft_silhouette" else 3e-4
    az0, el0, _ = syn.spherical_views(12)
    az, el = az0[None, 3:3 + V].astype(np.float32) + 2.0, el0[None, 3:3 + V].astype(np.float32) + 3.0
    di = np.full((1, V), 1.1, np.float32)
    blur = float(np.log(1.0 / 1e-4 - 1.0) * sigma)
    kw = dict(obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3)
    r = MVRenderer(V, size, pc_rendering=False, light_direction="fixed", faces_per_pixel=K, shader=shader, blur_radius=blur,
                   sigma=sigma, gamma=1e-4, return_alpha=True, z_clip=zc).eval()
    leaves = [_t(x).requires_grad_(True) for x in (az, el, di)]
    rgba, _ = r([Meshes([v], [f])], None, *leaves)
    want = oracle.render_meshes_soft([v], [f], az, el, di, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, z_clip=zc, **kw)
    whole = oracle.render_meshes_soft([v], [f], az, el, di, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, z_clip=0.0, **kw)
    assert zc == 0.0 or (want["pix_to_face"] != whole["pix_to_face"]).mean() > 0.01, "the case must depend on the cut"
    assert np.abs(rgba.detach().cpu().numpy() - want["images"]).max() <= 1e-4
    G = np.random.default_rng(9).normal(size=rgba.shape)
    if shader == "soft_silhouette":
        G[:, :, :3] = 0.0
    rgba.backward(_t(G))
    v64 = v.astype(np.float64)

    def L(a, e, d):
        o = oracle.render_meshes_soft([v64], [f], a, e, d, size, blur, K, shader=shader, sigma=sigma, gamma=1e-4, dtype=np.float64,
                                      fixed_p2f=want["pix_to_face"], fixed_piece=want["pix_to_piece"], z_clip=zc, **kw)
        return float((o["images"] * G).sum())
    base = [x.astype(np.float64) for x in (az, el, di)]
    worst = 0.0
    for k, name in enumerate(("azim", "elev", "dist")):
        ref = np.zeros(base[k].shape)
        for idx in np.ndindex(*base[k].shape):
            p, m = [x.copy() for x in base], [x.copy() for x in base]
            p[k][idx] += 1e-5
            m[k][idx] -= 1e-5
            ref[idx] = (L(*p) - L(*m)) / 2e-5
        got = leaves[k].grad.cpu().numpy().astype(np.float64)
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-12)
        worst = max(worst, rel)
        print("%s z_clip %.1f %s: relative error %.3g (|grad| up to %.3g)" % (shader, zc, name, rel, np.abs(ref).max()))
    assert worst < 1e-3, worst   # (observed: <= 3.3e-5 with and without cut faces)


# ---- the C++ autograd node (csrc/mvr_torch_op.cpp) against the Python one (functional._RenderMeshViews) -------------------------
# Both are plumbing over the same two C ABI calls; the operator exists for its host cost.  Same inputs ->
# the same bytes out of both, for the images, the cameras, the status words and the three gradients, on every branch of the
# backward: images only (the training step), images + cameras (the chained fall-back), cameras only, a window of draws.
@pytest.mark.gpu
@pytest.mark.parametrize("light", ["fixed", "relative"])
@pytest.mark.parametrize("through", ["images", "images+cameras", "cameras", "window"])
def test_torch_operator_matches_the_python_autograd_node(light, through):
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes, _lib
    from mvtn_amd.regularize import draw_regularization
    dev = _dev()
    if os.environ.get("MVR_TORCH_OP", "1") == "0":
        pytest.skip("MVR_TORCH_OP=0: the C++ node is switched off for this run")
    op = _lib.torch_op()
    assert op is not None, "mvtn_amd/_mvr_torch_op.so is not loaded (the build failed)"
    B, V, S = 3, 6, 96
    vl, fl = zip(*[syn.synthetic_mesh("S", i) for i in range(B)])
    packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
    az, el, di = syn.spherical_views(V)
    rng = np.random.default_rng(5)
    az, el, di = (np.tile(x, (B, 1)) + rng.uniform(-3, 3, (B, V)).astype(np.float32) * s for x, s in ((az, 1), (el, 1), (di, 0.02)))
    up = torch.randn(B, V, 3, S, S, device=dev, generator=torch.Generator(dev).manual_seed(3))
    draws = None
    if through == "window":
        torch.manual_seed(11)
        draws = draw_regularization(B, V, S, dev, dropout_p=0.3, augment_training=True, crop_ratio=0.3)

    def run(use_op):
        saved = _lib._TORCH_OP
        _lib._TORCH_OP = saved if use_op else None
        try:
            ta, te, td = (_t(x).to(dev).requires_grad_(True) for x in (az, el, di))
            r = MVRenderer(V, S, pc_rendering=False, light_direction=light).eval()
            img, cams = r(packed, None, ta, te, td, **({"window": draws} if draws is not None else {}))
            loss = 0
            if through != "cameras":
                loss = loss + (img * up).sum()
            if through in ("images+cameras", "cameras"):
                loss = loss + (cams.R * 0.37).sum() + (cams.T * -1.3).sum()
            loss.backward()
            torch.cuda.synchronize()
            return [x.detach().cpu().numpy() for x in (img, cams.R, cams.T, ta.grad, te.grad, td.grad)]
        finally:
            _lib._TORCH_OP = saved
    a, b = run(True), run(False)
    for name, x, y in zip(("images", "R", "T", "grad_azim", "grad_elev", "grad_dist"), a, b):
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), "%s differs between the C++ and the Python node" % name
    assert np.abs(a[3]).max() > 0
    # without autograd: the same frames, no saved state
    with torch.no_grad():
        r = MVRenderer(V, S, pc_rendering=False, light_direction=light).eval()
        img, _ = r(packed, None, _t(az).to(dev), _t(el).to(dev), _t(di).to(dev), **({"window": draws} if draws is not None else {}))
    assert img.cpu().numpy().tobytes() == a[0].tobytes()
