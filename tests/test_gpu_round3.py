"""-m gpu, round 3: the forward issued in pieces on two streams (results never depend on the piece count), files ->
`io.mesh_from_file` -> `MVRenderer` against the oracle, rendered views -> `save_grid` -> PNG decode
against the tensor (row f2)."""
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


@pytest.fixture
def pieces_reset():
    from mvtn_amd import functional as F
    yield F
    F.set_forward_pieces(0)


@pytest.mark.parametrize("specs,V,size,light", [
    ([("M", 0), ("S", 1), ("M", 2), ("L", 3), ("S", 4)], 6, 96, "relative"),     # ragged batch, 5 objects: pieces of 1-3 meshes
    ([("M", k) for k in range(8)], 12, 224, "fixed"),                            # half of C2
])
def test_forward_pieces_do_not_change_results(pieces_reset, specs, V, size, light):
    """mvr_set_forward_pieces: images, pixel -> face maps, cameras and gradients are IDENTICAL for 1 / 2 / 3 / 4 / 8 pieces and
    for the automatic choice (each piece's pre-pass evaluates the look-at of all views; the workspace is indexed by
    global mesh / view numbers)."""
    F = pieces_reset
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes
    dev = _dev()
    vl, fl = [], []
    for tier, seed in specs:
        v, f = syn.synthetic_mesh(tier, seed)
        vl.append(v); fl.append(f)
    B = len(specs)
    packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)
    az, el, di = _views(B, V, 7)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction=light).eval()
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(3)).to(dev)
    ref = None
    for pieces in (1, 2, 3, 4, 8, 0):
        F.set_forward_pieces(pieces)
        a, e, d = (_t(x).requires_grad_(True) for x in (az, el, di))
        img, cams = r(packed, None, a, e, d)
        grads = torch.autograd.grad((img * G).sum(), (a, e, d))
        with torch.no_grad():
            img2, _ = r(packed, None, a.detach(), e.detach(), d.detach())     # the inference path (no pixel lists)
        torch.cuda.synchronize()
        got = (img.detach().clone(), img2.clone(), cams.R.clone(), cams.T.clone(), [g.clone() for g in grads])
        if ref is None:
            ref = got
            assert float(img.min()) < 0.9 and torch.isfinite(img).all()
            continue
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), "images differ with %d pieces" % pieces
        assert torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
        for g, w in zip(got[4], ref[4]):
            assert torch.equal(g, w), "gradients differ with %d pieces" % pieces
    assert torch.equal(ref[0], ref[1])


def test_forward_pieces_fragments_and_remedy(pieces_reset):
    """pixel -> face maps with pieces, and the device-side rotation remedy (a degenerate view in the LAST piece makes every
    piece re-draw all views: the verdict is global) — identical to one piece."""
    F = pieces_reset
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes
    dev = _dev()
    vl, fl = zip(*[syn.synthetic_mesh("S", k) for k in range(6)])
    B, V, size = 6, 4, 64
    packed = PackedMeshes.from_meshes(Meshes(list(vl), list(fl)), dev)
    az, el, di = _views(B, V, 11)
    R, T, _ = F.look_at_view_transform(_t(di.reshape(-1)), _t(el.reshape(-1)), _t(az.reshape(-1)), return_status=True)
    params = F.mesh_params(V, size, 1, (1.0, 1.0, 1.0), (1.0, 1.0, 1.0))
    light = _t(np.array([[0.0, 1.0, 0.0]], np.float32))
    want = None
    for pieces in (1, 2, 3):
        F.set_forward_pieces(pieces)
        img, p2f = F.render_meshes_fragments(packed, R, T, light, params)
        torch.cuda.synchronize()
        if want is None:
            want = (img.clone(), p2f.clone())
            assert int((p2f >= 0).sum()) > 1000
        else:
            assert torch.equal(p2f, want[1]) and torch.equal(img, want[0])
    el_bad = el.copy()
    el_bad[5, 3] = 90.0      # invalid rotation matrix -> remedy re-draws ALL views
    outs = []
    for pieces in (1, 3):
        F.set_forward_pieces(pieces)
        torch.manual_seed(5)     # the remedy's counter-based draws are keyed by (torch.initial_seed(), calls of this module)
        r = MVRenderer(V, size, pc_rendering=False, rotation_check="device").eval()
        with torch.no_grad():
            img, cams = r(packed, None, _t(az), _t(el_bad), _t(di))
        r.check_status()
        torch.cuda.synchronize()
        outs.append((img.clone(), cams.R.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    R_plain, _, _ = F.look_at_view_transform(_t(di.reshape(-1)), _t(el_bad.reshape(-1)), _t(az.reshape(-1)), return_status=True)
    assert not torch.equal(outs[0][1][0], R_plain[0]), "the remedy re-draws every view, also those of the first piece"
    assert torch.isfinite(outs[0][0]).all()


def test_forward_pieces_inside_a_hip_graph(pieces_reset):
    """The side stream forks from and joins the capturing stream through events: a training step with a 2-piece forward
    captures into a HIP graph and replays bit-identically to eager execution."""
    F = pieces_reset
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes
    dev = _dev()
    vl, fl = zip(*[syn.synthetic_mesh("M", k) for k in range(4)])
    B, V, size = 4, 6, 96
    packed = PackedMeshes.from_meshes(Meshes(list(vl), list(fl)), dev)
    r = MVRenderer(V, size, pc_rendering=False, light_direction="relative").eval()
    G = torch.randn(B, V, 3, size, size, generator=torch.Generator().manual_seed(4)).to(dev)
    sets = [tuple(_t(x) for x in _views(B, V, seed)) for seed in (1, 2, 3)]
    F.set_forward_pieces(2)

    def step(a, e, d):
        img, _ = r(packed, None, a, e, d)
        return img, torch.autograd.grad((img * G).sum(), (a, e, d))

    eager = []
    for a, e, d in sets:
        out, grads = step(*[x.clone().requires_grad_(True) for x in (a, e, d)])
        eager.append((out.detach().clone(), [g.clone() for g in grads]))
    sa, se, sd = (x.clone().requires_grad_(True) for x in sets[0])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
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
    vv = io.rotation_matrix((1, 0, 0), -90.0).dot(v.astype(np.float64).T).T
    want_v = io.torch_center_and_normalize(torch.from_numpy(vv).to(torch.float), p=2)
    assert torch.equal(lv, want_v)
    assert abs(float(torch.linalg.norm(lv, dim=1).max()) - 1.0) < 1e-6 and float(lv.mean(0).abs().max()) < 1e-6
    V, size = 4, 96
    az, el, di = _views(1, V, 5)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white").eval()
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
    r = MVRenderer(V, size, pc_rendering=False, object_color="red", background_color="white").eval()
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
