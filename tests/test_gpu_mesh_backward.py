"""-m gpu: the mesh backward (row a13: k_mesh_backward_tab / k_mesh_backward + finalize) against fp64 finite differences of the oracle and torch.autograd fixtures (float64 and float32).

Filed by component in round 6; where each test came from (stable test names, so that rounds stay comparable):
  test_mesh_backward_is_bit_reproducible                                      test_gpu_parity.py
  test_mesh_backward_through_cut_faces_against_finite_differences             test_gpu_parity.py
  test_mesh_backward_at_the_c3_image_size_against_finite_differences          test_gpu_parity.py
  test_mesh_backward_table_kernel_equals_gather_kernel                        test_gpu_parity.py
  test_mesh_backward_random_configurations                                    test_gpu_parity.py
  test_full_c3_object_gradients_against_finite_differences                    test_gpu_round3.py
  test_backward_kernels_against_torch_autograd_c1                             test_gpu_round4.py
  test_backward_kernels_against_torch_autograd_c3                             test_gpu_round4.py
  test_backward_outliers_are_the_pixels_of_edge_on_faces                      test_gpu_round4.py
  test_cad_batch_through_the_module_and_backward                              test_gpu_round5.py
"""
import numpy as np
import pytest
import torch

from mvtn_amd import synthetic as syn

from gpu_helpers import _dev, _fd_grad_full, _fx, _hip_grads, _mesh_batch, _oracle_both_normal_sums, _t, _views, TOL_PIXEL  # noqa: F401

pytestmark = pytest.mark.gpu


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
    # that changes face changes the function by one pixel's worth.  Observed: ONE pixel, in view 11 of object 0 — an fp32
    # coincidence of this compiler / ROCm / architecture, so a BOUND is asserted and the location printed)
    print("c3: pixels on another face than the fixture's map, per (object, view):", {(int(b), int(v)): int(differ[b, v].sum()) for b, v in zip(*np.nonzero(differ.reshape(2, 12, -1).any(2)))})
    assert differ.sum() <= 2, differ.reshape(24, -1).sum(1)
    # a view that holds such a pixel differentiates a (one pixel) different function: its three derivatives are compared with a
    # loose bound, every other view with the tight one
    dirty = differ.reshape(24, -1).any(1).reshape(2, 12)
    for name, got in zip(("azim", "elev", "dist"), grads):
        ref = fx["c3_grad_" + name].astype(np.float64)
        err = np.abs(got - ref) / np.abs(ref).max()
        clean = err[~dirty].max()
        print("c3 %s (%s): max relative error %.2e over the %d views with identical maps (%.2e over the %d with a pixel on another face)" % (
            name, "gather" if gather else "table", clean, int((~dirty).sum()), err[dirty].max() if dirty.any() else 0.0, int(dirty.sum())))
        # fp32 kernels against a float64 derivative.  Rounds 4-5: <= 3.3e-3 at one view, <= 1.5e-3 at a second, <= 4e-4 elsewhere
        # (round 4 blamed relu kinks of the diffuse term; round 5 measured it: NO covered pixel of this case has |n . l| < 1e-5 —
        # the outliers are a few dozen pixels on nearly edge-on faces, next test).  Round 6: the barycentrics of the backward from
        # the RAY-RELATIVE vectors v_i - z_i d (differences first, as in the reference's edge functions) instead of the cross products
        # v_j x v_k of absolute view-space coordinates: <= 1.3e-3 at the two outlier views, <= 1.2e-4 elsewhere.
        assert clean <= 2e-3, (name, clean)
        assert err.max() <= 5e-3, (name, err.max())
    # Round 6: the same losses differentiated by torch.autograd in FLOAT32 — the arithmetic the reference
    # itself runs (tests/golden/make_torch_restatement_grad32.py) — beside the float64 fixture.  float32 autograd has the SAME two
    # outlier views — (object 0, view 9) and (object 1, view 3): 1.1e-3 / 5.9e-4 of the largest azim derivative — and the library
    # (since its barycentrics come from the ray-relative vectors) sits beside it: 1.2e-3 / 8.9e-4, means 9.4e-5 vs 7.7e-5 azim,
    # 4.0e-5 vs 3.2e-5 elev, 6.1e-5 vs 5.2e-5 dist (with the cross-product form of rounds 2-5: 2.6e-3 / 1.0e-3, means 2-5 x torch's,
    # and 3 of 69 derivatives beyond the bound).  Asserted, per derivative of every view with an identical map, the verdict's bound:
    # |library - f64| <= 2 |torch f32 - f64| + 2e-4 of the largest derivative.  (tools/r06_c3_err.py prints all 72.)
    f32 = _fx("torch_restatement_grad32.npz")
    for name, got in zip(("azim", "elev", "dist"), grads):
        ref = fx["c3_grad_" + name].astype(np.float64)
        scale = np.abs(ref).max()
        e_lib = np.where(dirty, 0.0, np.abs(got - ref) / scale)
        e_t32 = np.where(dirty, 0.0, np.abs(f32["c3_grad_%s_f32" % name] - ref) / scale)
        viol = ~(e_lib <= 2.0 * e_t32 + 2e-4)
        print("c3 %s (%s): |library - f64| max %.2e mean %.2e, |torch f32 - f64| max %.2e mean %.2e (relative to the largest derivative); beyond 2 x + 2e-4 at %s" % (
            name, "gather" if gather else "table", e_lib.max(), e_lib.mean(), e_t32.max(), e_t32.mean(), [tuple(int(x) for x in ij) for ij in np.argwhere(viol)]))
        assert not viol.any(), (name, e_lib, e_t32)
        assert e_lib.mean() <= 2.0 * e_t32.mean(), (name, e_lib.mean(), e_t32.mean())


@pytest.mark.parametrize("gather", [False, True], ids=["table_kernel", "gather_kernel"])
def test_backward_outliers_are_the_pixels_of_edge_on_faces(gather):
    """what the 2 of 72 C3 derivatives that miss 2e-3 (previous test) come from.  The fixture holds, per
    threshold t, the bit mask of the pixels whose winning face is badly conditioned — cond = (larger extent of its NDC box)^2 /
    |2 x NDC area| > t: nearly edge-on, the barycentrics' derivative ~ 1 / area — and torch.autograd's gradients of the loss WITHOUT
    those pixels.  With the same mask on the upstream gradient the fp32 kernels agree with float64 autograd to <= 2e-4 on every
    derivative of every view with an identical pixel -> face map once the 40 pixels with cond > 100 (0.008 % of the covered
    ones) are out (round 6, ray-relative barycentrics: <= 1.5e-4, and already with only the 4 pixels of cond > 300 out; with 1 pixel
    out 2 derivatives miss 2e-4.  Rounds 4-5, cross-product form: 7 with 4 pixels out, 12 with 1).  (And no
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
    assert over[1000.0] >= over[300.0] >= over[100.0]     # the fewer of those pixels are masked, the more derivatives miss the bound


def test_cad_batch_through_the_module_and_backward(oracle):
    """MVRenderer on a CAD-like batch with the relative light: forward against the oracle, and both backward kernels agree (the
    table kernel reads the rasteriser's pixel lists — since round 5 every strip of this geometry stays on the fast path)."""
    from mvtn_amd import MVRenderer, Meshes
    oracle.set_threads(8)
    vl, fl = zip(*[syn.cad_mesh(s, n_faces=1280) for s in (3, 4)])
    rng = np.random.default_rng(5)
    az, el, di = (np.tile(x, (2, 1)) for x in syn.spherical_views(6))
    az = az + rng.uniform(-5, 5, az.shape).astype(np.float32)
    el = el + rng.uniform(-5, 5, el.shape).astype(np.float32)
    G = torch.randn(2, 6, 3, 128, 128, generator=torch.Generator().manual_seed(1)).to(_dev())
    grads = []
    for gather in (False, True):
        a, e, d = (_t(x).requires_grad_(True) for x in (az, el, di))
        r = MVRenderer(6, 128, pc_rendering=False, light_direction="relative").eval()
        if gather:
            import mvtn_amd.functional as F
            orig = F.mesh_params
            F.mesh_params = lambda *args, **kw: orig(*args, **dict(kw, gather_backward=True))
        try:
            img, _ = r([Meshes([torch.from_numpy(v)], [torch.from_numpy(f)]) for v, f in zip(vl, fl)], None, a, e, d)
        finally:
            if gather:
                F.mesh_params = orig
        img.backward(G)
        grads.append([x.grad.double().cpu().numpy() for x in (a, e, d)])
        if not gather:
            want, _, _ = _oracle_both_normal_sums(oracle, list(vl), list(fl), az, el, di, 128, light_dir=oracle.camera_position(az, el, di))
            assert np.abs(img.detach().cpu().numpy() - want["images"]).max() <= 1e-4
    for ga, gb in zip(*grads):
        assert np.isfinite(ga).all() and np.abs(ga - gb).max() <= 5e-4 * (np.abs(gb).max() + 1.0)
