"""-m gpu: the view regulariser, Normalize and sampling windows (row f3) fused into the rasterisers or as their own pass, against the torch composition and the reference's fixture.

Filed by component in round 6; where each test came from (stable test names, so that rounds stay comparable):
  test_normalize_fused_into_the_rasteriser_stores                             test_gpu_parity.py
  test_regulariser_window_rendered_by_the_rasteriser                          test_gpu_parity.py
  test_regularize_views_matches_the_torch_composition                         test_gpu_parity.py
  test_regularize_views_after_the_renderer_trains_end_to_end                  test_gpu_parity.py
  test_regularize_views_reproduces_the_reference_fixture                      test_gpu_parity.py
  test_regularize_normalize_views_fused_equals_chained                        test_gpu_parity.py
  test_with_window_params_are_collectable_after_a_forward                     test_gpu_round5.py
"""
import gc
import os
import weakref

import numpy as np
import pytest
import torch

from mvtn_amd import synthetic as syn

from gpu_helpers import _dev, _mesh_batch, _t, _views, _regularize_composition  # noqa: F401

pytestmark = pytest.mark.gpu


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


def test_with_window_params_are_collectable_after_a_forward():
    """render_mesh_views used to cache a torch view of the params struct ON the struct — a reference cycle
    through C++ that the garbage collector cannot see; every `with_window` copy (one per training step, holding four device
    tensors) then lived forever.  The view is now made per call."""
    from mvtn_amd import Meshes, functional as F
    from mvtn_amd.structures import PackedMeshes
    v, f = syn.synthetic_mesh("S", 0)
    packed = PackedMeshes.from_meshes(Meshes([v], [f]), _dev())
    base = F.mesh_params(2, 64, 1, (1, 1, 1), (1, 1, 1))
    n = 2
    p = F.with_window(base, torch.ones(n, device=_dev()), torch.zeros(n, dtype=torch.uint8, device=_dev()),
                      torch.full((n,), 2, dtype=torch.int32, device=_dev()), torch.full((n,), 2, dtype=torch.int32, device=_dev()), 2)
    assert p is not base
    ref = weakref.ref(p)
    a, e, d = _t([0.0, 90.0]), _t([30.0, 30.0]), _t([2.2, 2.2])
    out = F.render_mesh_views(packed, a, e, d, _t([[0.0, 1.0, 0.0]]), p)
    torch.cuda.synchronize()
    del out, p
    gc.collect()
    assert ref() is None, "the with_window params object is still alive after its forward"
