"""-m gpu: the soft mesh path (row f4: SoftPhongShader / SoftSilhouetteShader, K nearest fragments, coverage gradient, cut faces) against the oracle.

Filed by component in round 6; where each test came from (stable test names, so that rounds stay comparable):
  test_soft_mesh_forward_parity                                               test_gpu_parity.py
  test_soft_mesh_backward_against_finite_differences                          test_gpu_parity.py
  test_hip_soft_phong_against_the_torch_restatement                           test_gpu_round3.py
  test_soft_path_cuts_straddling_faces                                        test_gpu_round3.py
  test_soft_backward_on_cut_faces_against_finite_differences                  test_gpu_round4.py
"""
import numpy as np
import pytest
import torch

from mvtn_amd import synthetic as syn

from gpu_helpers import _dev, _fd_grad_soft, _fx, _mesh_batch, _soft_blur, _t, _views, _well_conditioned_pixels, TOL_PIXEL  # noqa: F401

pytestmark = pytest.mark.gpu


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
