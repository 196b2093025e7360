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
