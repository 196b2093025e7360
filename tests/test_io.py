"""Mesh ingestion: OFF / OBJ parsing and the dataset-side normalisation, CPU only."""
import importlib.util

import numpy as np
import pytest
import torch

from mvtn_amd import io as mio

io = mio

CUBE_OFF = """OFF
# a cube with quad faces
8 6 12
0 0 0
1 0 0
1 1 0
0 1 0
0 0 1
1 0 1
1 1 1
0 1 1
4 0 3 2 1
4 4 5 6 7
4 0 1 5 4
4 1 2 6 5
4 2 3 7 6
4 3 0 4 7  255 0 0
"""


def _signed_volume(v, f):
    return float(np.einsum("ij,ij->i", v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6.0)


def test_off_quads_glued_header_and_winding(tmp_path):
    p = tmp_path / "cube.off"
    p.write_text(CUBE_OFF)
    v, f = mio.load_off(str(p))
    assert v.shape == (8, 3) and f.shape == (12, 3) and f.dtype == np.int64
    assert abs(_signed_volume(v, f) - 1.0) < 1e-12          # fan triangulation keeps the outward winding
    q = tmp_path / "glued.off"
    q.write_text(CUBE_OFF.replace("OFF\n# a cube with quad faces\n8 6 12", "OFF8 6 12"))   # ModelNet's header quirk
    v2, f2 = mio.load_off(str(q))
    assert np.array_equal(v, v2) and np.array_equal(f, f2)
    with pytest.raises(ValueError):
        (tmp_path / "bad.off").write_text("OFF\n3 1 0\n0 0 0\n1 0 0\n0 1 0\n3 0 1 5\n")
        mio.load_off(str(tmp_path / "bad.off"))


def test_obj_slashes_negative_indices_polygons(tmp_path):
    p = tmp_path / "quad.obj"
    p.write_text("# quad + triangle\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1 4/1/1\nv 0 0 1\nf -1 -5 -4\n")
    v, f = mio.load_obj(str(p))
    assert v.shape == (5, 3)
    assert f.tolist() == [[0, 1, 2], [0, 2, 3], [4, 0, 1]]
    assert mio.load_mesh(str(p))[1].shape == (3, 3)


def test_rotation_and_normalisation_match_the_reference_formulas():
    R = mio.rotation_matrix([1, 0, 0], -90)
    assert np.allclose(R @ np.array([0, 0, 1.0]), [0, -1, 0]) or np.allclose(R @ np.array([0, 0, 1.0]), [0, 1, 0])
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15) and abs(np.linalg.det(R) - 1) < 1e-15
    pts = torch.tensor([[0.0, 0, 0], [2, 0, 0], [0, 4, 0], [0, 0, 6]])
    out = mio.torch_center_and_normalize(pts, p=2)
    assert torch.allclose(out.mean(0), torch.zeros(3), atol=1e-7)
    assert abs(float(out.norm(dim=1).max()) - 1.0) < 1e-6
    out_inf = mio.torch_center_and_normalize(pts, p="inf")
    assert abs(float(out_inf.abs().max()) - 1.0) < 1e-6
    assert torch.allclose(mio.torch_center_and_normalize(pts, p="no"), pts - pts.mean(0))


def test_rotation_matrix_against_the_reference_generated_golden():
    """tests/golden/rotation_matrix.json: util.rotation_matrix of the reference itself (make_rotation_golden.py) on 25 axes /
    angles; the Rodrigues form here equals it to rounding."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rotation_matrix.json")) as f:
        cases = json.load(f)
    assert len(cases) == 25
    for c in cases:
        assert np.abs(mio.rotation_matrix(c["axis"], c["theta_deg"]) - np.asarray(c["R"])).max() < 1e-15


def test_mesh_from_file_yields_what_the_renderer_takes(tmp_path):
    p = tmp_path / "cube.off"
    p.write_text(CUBE_OFF)
    m = mio.mesh_from_file(str(p))
    v, f = m.verts_list()[0], m.faces_list()[0]
    assert v.dtype == torch.float32 and f.dtype == torch.int64 and v.shape == (8, 3) and f.shape == (12, 3)
    assert abs(float(v.norm(dim=1).max()) - 1.0) < 1e-6 and torch.allclose(v.mean(0), torch.zeros(3), atol=1e-6)
    m2 = mio.mesh_from_file(str(p), flip_faces=True)
    assert _signed_volume(m2.verts_list()[0].numpy().astype(np.float64), m2.faces_list()[0].numpy()) < 0


def test_scanobjectnn_bin_matches_the_reference_loader(tmp_path):
    """custom_dataset.py:553-575 ran on synthetic .bin records when tests/golden/scanobjectnn_bin.npz was made
    (tests/golden/make_scanobjectnn_golden.py): same points, bit for bit, for the object-only, with-background and SUNCG
    variants and for a tie between two foreground instances."""
    import os
    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "scanobjectnn_bin.npz"))
    assert len(fx["names"]) == 4
    for name in fx["names"]:
        name = str(name)
        suncg, with_bg = (int(x) for x in fx["cfg_" + name])
        p = tmp_path / (name + ".bin")
        fx["raw_" + name].tofile(str(p))
        got = io.load_scanobjectnn_bin(str(p), "with_bg" if with_bg else "obj_only", bool(suncg))
        assert got.dtype == np.float32 and np.array_equal(got, fx["pts_" + name]), name
    pts = io.scanobjectnn_points(fx["pts_obj"], 256, 2, np.random.RandomState(0))
    np.random.seed(0)
    ref = fx["pts_obj"][np.random.randint(fx["pts_obj"].shape[0], size=256), :]      # the reference's draw
    want = io.torch_center_and_normalize(torch.from_numpy(ref).to(torch.float), p=2)
    assert pts.shape == (256, 3) and torch.equal(pts, want)
    assert abs(float(pts.norm(dim=1).max()) - 1.0) < 1e-6
    if importlib.util.find_spec("h5py") is None:
        with pytest.raises(ImportError, match="h5py"):
            io.load_scanobjectnn_h5(str(tmp_path / "x.h5"))
