"""CPU (-m "not gpu"): host-side logic of the product against the golden vectors captured from the
reference's own caller-side code (tests/golden/caller_side.json), the C-ABI surface, and the packing."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from mvtn_amd import synthetic as syn
from mvtn_amd.helpers import batch_tensor, check_valid_rotation_matrix, flatten_views, torch_color, unbatch_tensor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unit_spherical_grid_matches_reference(golden):
    for n, g in golden["unit_spherical_grid"].items():
        az, el = syn.unit_spherical_grid(int(n))
        assert np.allclose(az, g["azim"], atol=1e-9) and np.allclose(el, g["elev"], atol=1e-9), n


def test_selector_buffers_match_reference(golden):
    for key, g in golden["selectors"].items():
        cfg, V = key.split("_")
        V = int(V)
        if cfg == "circular" or V == 4:  # models/mvtn.py:303: spherical with 4 views falls back to circular
            az, el, di = syn.circular_views(V, 30.0, 2.2)
        else:
            az, el, di = syn.spherical_views(V, 2.2)
        assert np.allclose(az, g["azim"], atol=1e-4), key
        assert np.allclose(el, g["elev"], atol=1e-4), key
        assert np.array_equal(di, np.asarray(g["dist"], np.float32)), key
        assert g["stride"] == [0, 1]  # the selectors hand the renderer stride-(0,1) expanded views


def test_flatten_order_matches_batch_tensor(golden):
    x = torch.arange(6.0).reshape(2, 3)
    assert batch_tensor(x.T, dim=1, squeeze=True).tolist() == golden["batch_tensor_2x3"]
    assert flatten_views(x).tolist() == golden["batch_tensor_2x3"]
    xe = torch.arange(4.0).expand(3, 4)  # stride (0,1), as MVTN's selectors produce
    assert flatten_views(xe).tolist() == golden["batch_tensor_expanded_3x4"]
    assert batch_tensor(xe.T, dim=1, squeeze=True).tolist() == golden["batch_tensor_expanded_3x4"]


def test_relayout_equivalence(golden):
    """models/renderer.py:109-113 replayed with our unbatch_tensor: images[b,v,c] == raw[b*V+v,:,:,c]."""
    B, V, H, W = 3, 4, 5, 6
    raw = torch.arange(B * V * H * W * 4, dtype=torch.float32).reshape(B * V, H, W, 4)
    img = unbatch_tensor(raw.clone(), batch_size=V, dim=1, unsqueeze=True).transpose(0, 1)
    img = img[..., 0:3].transpose(2, 4).transpose(3, 4)
    assert list(img.shape) == golden["relayout"]["shape"] and golden["relayout"]["matches_n_eq_bV_plus_v"]
    ours = raw[..., :3].permute(0, 3, 1, 2).reshape(B, V, 3, H, W)  # what the HIP kernel writes directly
    assert torch.equal(img, ours)


def test_torch_color_matches_reference(golden):
    for name, g in golden["torch_color"].items():
        assert torch_color(name).tolist() == g["plain"]
        got = torch_color(name, max_lightness=True).numpy()
        assert [int(x) for x in got.view(np.uint32)] == g["max_lightness_bits"], name
    torch.manual_seed(0)
    c = torch_color("random", max_lightness=True)
    assert c.shape == (3,) and float(c.max()) < 1.0 and float(c.max()) > 0.9999
    assert torch_color("custom", custom_color=(0.1, 0.2, 0.3)).tolist() == pytest.approx([0.1, 0.2, 0.3])
    with pytest.raises(ValueError):
        torch_color("chartreuse")


def test_check_valid_rotation_matrix_matches_reference(golden):
    g = golden["check_valid_rotation_matrix"]
    eye = torch.eye(3)[None]
    m = eye.clone(); m[0, 0, 0] = 1.0001
    assert check_valid_rotation_matrix(eye) == g["identity"] and check_valid_rotation_matrix(m) == g["diag_1.0001"]
    m = eye.clone(); m[0, 0, 1] = 2e-6
    assert check_valid_rotation_matrix(m) == g["offdiag_2e-6"]
    m = eye.clone(); m[0, 0, 1] = 5e-7
    assert check_valid_rotation_matrix(m) == g["offdiag_5e-7"]


def test_center_and_normalize_matches_reference(golden):
    g = golden["center_and_normalize"]
    got = syn.center_and_normalize(np.asarray(g["in"], np.float64))
    assert np.allclose(got, g["out"], atol=2e-6)


def test_synthetic_meshes():
    for tier, (nv, nf) in {"S": (642, 1280), "M": (2562, 5120), "L": (10242, 20480)}.items():
        v, f = syn.synthetic_mesh(tier, 0)
        assert v.shape == (nv, 3) and f.shape == (nf, 3) and v.dtype == np.float32 and f.dtype == np.int64
        assert abs(np.linalg.norm(v, axis=1).max() - 1) < 1e-6 and np.abs(v.mean(0)).max() < 1e-6
        # closed, consistently outward-wound: signed volume positive, every edge shared by two faces
        a, b, c = (v[f[:, k]].astype(np.float64) for k in range(3))
        assert np.einsum("ij,ij->i", a, np.cross(b, c)).sum() > 0
    v2, _ = syn.synthetic_mesh("M", 0)
    v3, _ = syn.synthetic_mesh("M", 1)
    assert np.array_equal(v2, syn.synthetic_mesh("M", 0)[0]) and not np.array_equal(v2, v3)


def test_cad_mesh_is_what_its_docstring_says():
    """synthetic.cad_mesh (bench.py --geometry cad): the reference's geometry class — big flat triangles beside slivers, coincident
    reversed duplicates (on shared and on duplicated vertices), exactly n_faces faces, normalised like custom_dataset.py:120-121."""
    for seed in range(3):
        v, f = syn.cad_mesh(seed)
        assert f.shape == (5120, 3) and f.dtype == np.int64 and v.dtype == np.float32 and f.min() >= 0 and f.max() < len(v)
        assert abs(np.linalg.norm(v, axis=1).max() - 1) < 1e-6 and np.abs(v.mean(0)).max() < 1e-5
        a, b, c = (v[f[:, k]].astype(np.float64) for k in range(3))
        area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
        assert area.max() > 100 * np.median(area)                       # two-triangle sides next to tessellated patches
        edges = np.stack([np.linalg.norm(a - b, axis=1), np.linalg.norm(b - c, axis=1), np.linalg.norm(c - a, axis=1)], 1)
        assert (edges.max(1) / np.maximum(edges.min(1), 1e-12) > 8).mean() > 0.15   # slivers
        # coincident pairs: the same three POSITIONS with opposite winding, by index and by duplicated vertices
        key = [tuple(sorted(map(tuple, tri))) for tri in v[f].tolist()]
        from collections import Counter
        n_dup = sum(1 for k, n in Counter(key).items() if n > 1)
        assert 0.05 * len(f) < n_dup < 0.45 * len(f)
        shared = sum(1 for k, n in Counter(tuple(sorted(t)) for t in f.tolist()).items() if n > 1)
        assert 0 < shared < n_dup
        assert np.array_equal(v, syn.cad_mesh(seed)[0])
    assert syn.cad_mesh(0, n_faces=1280)[1].shape == (1280, 3)
    assert syn.ragged_tiers(16).count("S") == 4 and syn.ragged_tiers(16).count("M") == 8 and syn.ragged_tiers(16).count("L") == 4
    assert [len(syn.geometry_mesh("ragged", "M", i)[1]) for i in range(4)] == [1280, 5120, 20480, 5120]
    assert len(syn.geometry_mesh("cad", "L", 1)[1]) == 20480 and len(syn.geometry_mesh("ico", "S", 1)[1]) == 1280


def test_c_abi_library_exports_every_declared_symbol():
    """libmvr_hip.so loads without a GPU and exports exactly what include/mvr.h declares."""
    from mvtn_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mvr.h")).read()
    declared = set(re.findall(r"\b(mvr_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libmvr_hip.so does not export %s" % name
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    assert lib.mvr_version() == _lib.MVR_ABI_VERSION == 103
    # struct layouts agree with the header (sizes computed by hand from include/mvr.h)
    assert ctypes.sizeof(_lib.MeshBatch) == 6 * 4 + 5 * 8
    assert ctypes.sizeof(_lib.MeshParams) == 6 * 4 + 5 * 4 + 6 * 4 + 4 + 6 * 4 + 4 * 8 + 8
    assert ctypes.sizeof(_lib.PointsParams) == 8 * 4 + 8 + 6 * 4 + 6 * 4 + 4 * 8 + 8


def test_c_abi_argument_validation_without_gpu():
    """Argument errors are reported before any HIP call, so they can be exercised on a CPU-only box."""
    from mvtn_amd import _lib
    lib = _lib.load()
    assert lib.mvr_look_at(None, None, None, 4, None, None, None, None, None) == 1
    assert b"bad argument" in lib.mvr_last_error()
    mb = _lib.MeshBatch(0, 0, 0, 0, 0, 0, None, None, None, None, None)
    mp = _lib.MeshParams(1, 8, 8, 0, 1, 0, 60.0, 0.5, 0.3, 0.2, 64.0)
    assert lib.mvr_render_meshes_forward(mb, mp, None, None, None, None, None, None, 0, None) == 1
    pp = _lib.PointsParams(1, 10, 1, 8, 8, 0, 1, 0, -1.0)
    assert lib.mvr_render_points_forward(pp, None, None, None, None, None, None, None, None, 0, None) == 1
    assert b"radius" in lib.mvr_last_error()
    pp = _lib.PointsParams(1, 10, 1, 8, 8, 0, 151, 0, 0.01)
    assert lib.mvr_render_points_forward(pp, None, None, None, None, None, None, None, None, 0, None) == 1
    assert b"points_per_pixel" in lib.mvr_last_error()
    from mvtn_amd import functional as F
    pk = F.points_params(2, 100, 3, 32, 0.01, (1, 1, 1), (0, 0, 0), points_per_pixel=4)
    p1 = F.points_params(2, 100, 3, 32, 0.01, (1, 1, 1), (0, 0, 0))
    assert lib.mvr_points_workspace_bytes(pk) >= lib.mvr_points_workspace_bytes(p1) + 6 * 32 * 32 * 4 * 4
    assert lib.mvr_mesh_workspace_bytes(None, None) == 0


def test_flag_constants_and_workspace_rules_without_gpu():
    """The Python mirror of the mvr_mesh_params flags equals the header's enum; MVR_KEEP_PIXEL_LISTS enlarges the workspace
    (and only appends to it); z_clip > 0 without perspective-correct rasterisation is rejected (PyTorch3D never clips
    then) — all before any HIP call."""
    from mvtn_amd import _lib, functional as F
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mvr.h")).read()
    enum = dict((k, int(v)) for k, v in re.findall(r"\b(MVR_[A-Z_]+)\s*=\s*(\d+)", header))
    for name in ("MVR_CULL_BACKFACES", "MVR_PERSPECTIVE_CORRECT", "MVR_WORKSPACE_FROM_FORWARD", "MVR_RASTER_EXACT_ONLY",
                 "MVR_BACKWARD_GATHER", "MVR_KEEP_PIXEL_LISTS", "MVR_NORMALS_THREE_CROSS", "MVR_PROJECTION_FMA"):
        assert getattr(_lib, name) == enum[name], name
    assert F.mesh_params(12, 224, 1, (1, 1, 1), (1, 1, 1), projection_fma=True).flags & _lib.MVR_PROJECTION_FMA
    assert not F.mesh_params(12, 224, 1, (1, 1, 1), (1, 1, 1)).flags & _lib.MVR_PROJECTION_FMA   # the default is the closed form
    flags = [enum[k] for k in ("MVR_CULL_BACKFACES", "MVR_PERSPECTIVE_CORRECT", "MVR_WORKSPACE_FROM_FORWARD", "MVR_RASTER_EXACT_ONLY",
                               "MVR_BACKWARD_GATHER", "MVR_KEEP_PIXEL_LISTS")]
    assert sorted(flags) == [1, 2, 4, 8, 16, 32]
    mb = _lib.MeshBatch(2, 200, 400, 200, 100, 0, 1, 1, 1, 1, None)   # (pointers only have to be non-null for the size query)
    base = F.mesh_params(12, 224, 1, (1, 1, 1), (1, 1, 1))
    keep = _lib.MeshParams.from_buffer_copy(base)
    keep.flags |= _lib.MVR_KEEP_PIXEL_LISTS
    n0, n1 = lib.mvr_mesh_workspace_bytes(mb, base), lib.mvr_mesh_workspace_bytes(mb, keep)
    n_strips, cap = 224 // 8, 8 * 224
    assert n1 >= n0 + 24 * n_strips * (8 + cap * 6) and n1 <= n0 + 24 * n_strips * (8 + cap * 6) + 3 * 256
    bad = F.mesh_params(12, 224, 1, (1, 1, 1), (1, 1, 1), perspective_correct=False, z_clip=0.5)
    assert lib.mvr_render_meshes_forward(mb, bad, 1, 1, 1, 1, None, 1, 1 << 40, None) == 1
    assert b"z_clip" in lib.mvr_last_error()
    assert F.mesh_params(12, 224, 1, (1, 1, 1), (1, 1, 1), perspective_correct=False).z_clip == 0.0   # the default follows the reference's rule
    assert abs(F.mesh_params(12, 224, 1, (1, 1, 1), (1, 1, 1)).z_clip - 0.5) < 1e-12
    # the sampling-window fields go together (mesh and point parameters), also checked before any HIP call
    for make, call in ((lambda: F.mesh_params(12, 224, 1, (1, 1, 1), (1, 1, 1)),
                        lambda p: lib.mvr_render_meshes_forward(mb, p, 1, 1, 1, 1, None, 1, 1 << 40, None)),
                       (lambda: F.points_params(2, 100, 12, 224, 0.006, (1, 1, 1), (0, 0, 0)),
                        lambda p: lib.mvr_render_points_forward(p, 1, None, 1, 1, 1, 1, None, 1, 1 << 40, None))):
        for fields in ({"win_dx": 1}, {"win_dx": 1, "win_dy": 1}, {"win_scale": 1}, {"win_pad": 3}, {"reserved2_": 1}, {"fov_scale": -1.0}):
            p = make()
            if not all(hasattr(p, k) for k in fields):   # (reserved2_: points parameters; fov_scale: mesh parameters)
                continue
            for k, v in fields.items():
                setattr(p, k, v)
            assert call(p) == 1, fields
            assert any(w in lib.mvr_last_error() for w in (b"win_", b"reserved2_", b"fov_scale")), (fields, lib.mvr_last_error())


def test_mesh_workspace_sizes_without_gpu():
    """Round 6: what a mesh forward asks its caller to allocate.  The settle's candidates live
    in the gap of the strips' list regions (round 5: an area of their own, 16 B per pixel per view): C2 (16 tier-M meshes x 12 views,
    224 x 224) 387 -> 271 MB, all of C5 on one GPU 3.68 -> 2.36 GB.  No list regions where the pre-pass cannot build lists (more
    than 254 strips; the comment in mesh_ws_layout used to call them 'address space': callers commit every byte)."""
    from mvtn_amd import _lib, functional as F
    lib = _lib.load()

    def ws(B, V, nv, nf, size, **kw):
        mb = _lib.MeshBatch(B, B * nv, B * nf, nf, nv, 0, 1, 1, 1, 1, None)
        return lib.mvr_mesh_workspace_bytes(mb, F.mesh_params(V, size, 1, (1, 1, 1), (1, 1, 1), **kw))
    c2, c5 = ws(16, 12, 2562, 5120, 224), ws(64, 20, 10242, 20480, 224)
    print("workspace: C2 %.1f MB, all of C5 %.2f GB" % (c2 / 1e6, c5 / 1e9))
    assert c2 <= 280e6 and c5 <= 2.4e9
    # per (view, strip) a region of half the faces + a quarter entry per pixel, 16 B per entry (tier M at 224 x 8: 3 008 entries)
    assert abs(c2 - 16 * 12 * 28 * 3008 * 16) <= 16e6
    # 2 400 rows in strips of 8 = 300 strips > 254: no strip lists -> no regions: the workspace is the per-vertex / per-face tables
    tall = ws(2, 2, 2562, 5120, 2400, strip_rows=8)
    assert tall <= 4 * (2562 * 16 * 4 + 5120 * 2) + 4 * 300 * 128 + 2 * 2562 * (24 + 32) + 65536, tall
    # a mesh of few faces still gets room for two candidates per pixel of a strip (one doubled quad over the whole image)
    tiny = ws(1, 1, 8, 5, 224)
    assert tiny >= 28 * (5 + 224 * 8) * 16


def test_product_has_no_cpu_fallback_and_never_imports_the_oracle():
    import mvtn_amd
    from mvtn_amd import MVRenderer
    r = MVRenderer(2, 16, pc_rendering=False)
    with pytest.raises(RuntimeError):
        r(None, None, torch.zeros(1, 2), torch.zeros(1, 2), torch.ones(1, 2))
    pkg = os.path.dirname(mvtn_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, fn


def test_packed_meshes_layout():
    from mvtn_amd.structures import Meshes, PackedMeshes
    vl, fl = zip(syn.synthetic_mesh("S", 0), syn.cad_soup(1), syn.synthetic_mesh("M", 2))
    ms = Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl])
    assert len(ms) == 3 and len(ms[1].verts_list()) == 1 and ms.verts_padded().shape == (3, 2562, 3)
    for src in (ms, [ms[i] for i in range(3)]):  # batched object or list of single-mesh objects (both reference forms)
        p = PackedMeshes.from_meshes(src, "cpu")
        assert p.n_meshes == 3 and p.total_verts == sum(len(v) for v in vl) and p.total_faces == sum(len(f) for f in fl)
        assert p.max_verts == 2562 and p.max_faces == 5120 and p.faces.dtype == torch.int32
        assert p.vert_off.tolist() == [0, 642, 642 + len(vl[1]), p.total_verts]
        assert torch.equal(p.faces[int(p.face_off[1]):int(p.face_off[2])], torch.from_numpy(fl[1]).to(torch.int32))
    with pytest.raises(ValueError):
        PackedMeshes.from_meshes(Meshes([torch.zeros(3, 3)], [torch.tensor([[0, 1, 3]])]), "cpu")
    col = torch.rand(3, 2562, 3)
    p = PackedMeshes.from_meshes(ms, "cpu", vert_colors=col)
    assert torch.equal(p.vert_colors[642:642 + len(vl[1])], col[1, :len(vl[1])])


def test_pack_meshes_host_c_abi():
    """mvr_pack_meshes_layout / mvr_pack_meshes_host (host-only C-ABI functions, include/mvr.h): the one staging
    block holds exactly what the numpy packing gives, sections are 256-byte aligned, bad indices are refused."""
    from mvtn_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(3)
    sizes = [(5, 7), (1, 1), (300, 1000), (64, 33)]        # ragged, including a one-vertex degenerate mesh
    vs = [rng.normal(size=(nv, 3)).astype(np.float32) for nv, _ in sizes]
    fs = [rng.integers(0, nv, size=(nf, 3)).astype(np.int64) for nv, nf in sizes]
    fs[2][-1, -1] = sizes[2][0] - 1                          # the largest legal index
    n = len(sizes)
    nv = (ctypes.c_int32 * n)(*[s[0] for s in sizes])
    nf = (ctypes.c_int32 * n)(*[s[1] for s in sizes])
    lay = _lib.PackLayout()
    assert lib.mvr_pack_meshes_layout(n, nv, nf, ctypes.byref(lay)) == 0
    assert (lay.total_verts, lay.total_faces, lay.max_verts, lay.max_faces) == (370, 1041, 300, 1000)
    ats = [lay.verts_at, lay.faces_at, lay.vert_off_at, lay.face_off_at, lay.bytes]
    assert all(a % 256 == 0 for a in ats) and ats == sorted(ats) and lay.faces_at >= 370 * 12
    block = np.full(lay.bytes, 0xAB, np.uint8)
    vp = (ctypes.c_void_p * n)(*[v.ctypes.data for v in vs])
    fp = (ctypes.c_void_p * n)(*[f.ctypes.data for f in fs])
    assert lib.mvr_pack_meshes_host(ctypes.byref(lay), vp, nv, fp, nf, block.ctypes.data) == 0
    assert np.array_equal(block[lay.verts_at:lay.verts_at + 370 * 12].view(np.float32).reshape(-1, 3), np.concatenate(vs))
    assert np.array_equal(block[lay.faces_at:lay.faces_at + 1041 * 12].view(np.int32).reshape(-1, 3), np.concatenate(fs).astype(np.int32))
    assert block[lay.vert_off_at:lay.vert_off_at + 20].view(np.int32).tolist() == [0, 5, 6, 306, 370]
    assert block[lay.face_off_at:lay.face_off_at + 20].view(np.int32).tolist() == [0, 7, 8, 1008, 1041]
    for bad in (-1, 300, 2 ** 31, 2 ** 40 + 3, -2 ** 33):   # out of range, including values whose low word looks legal
        g = fs[2].copy()
        g[17, 1] = bad
        fp2 = (ctypes.c_void_p * n)(fs[0].ctypes.data, fs[1].ctypes.data, g.ctypes.data, fs[3].ctypes.data)
        assert lib.mvr_pack_meshes_host(ctypes.byref(lay), vp, nv, fp2, nf, block.ctypes.data) == 1, bad
        assert b"mesh 2" in lib.mvr_last_error()
    nv0 = (ctypes.c_int32 * n)(5, 0, 300, 64)
    assert lib.mvr_pack_meshes_layout(n, nv0, nf, ctypes.byref(lay)) == 1 and b"mesh 1" in lib.mvr_last_error()
    assert lib.mvr_pack_meshes_layout(0, nv, nf, ctypes.byref(lay)) == 1


def test_packed_meshes_accepts_what_callers_hand_over():
    from mvtn_amd.structures import Meshes, PackedMeshes
    v, f = syn.synthetic_mesh("S", 0)
    want = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v)], [torch.from_numpy(f)]), "cpu")
    # non-contiguous vertices, int32 faces, float64 vertices, numpy inputs: converted on the way in
    vt = torch.from_numpy(np.ascontiguousarray(v.T)).T
    for vv, ff in ((vt, torch.from_numpy(f)), (torch.from_numpy(v).double(), torch.from_numpy(f).int()), (v, f)):
        class One:          # a single-mesh object with the two accessors the reference uses (models/renderer.py:67-68)
            def verts_list(self, vv=vv): return [vv]
            def faces_list(self, ff=ff): return [ff]
        got = PackedMeshes.from_meshes([One()], "cpu")
        assert torch.equal(got.verts, want.verts) and torch.equal(got.faces, want.faces)
        assert got.vert_off.tolist() == [0, 642] and got.face_off.tolist() == [0, 1280]
    with pytest.raises(ValueError):
        PackedMeshes.from_meshes([], "cpu")
    with pytest.raises(ValueError):
        PackedMeshes.from_meshes(Meshes([torch.zeros(0, 3)], [torch.zeros(1, 3, dtype=torch.int64)]), "cpu")


def test_cameras_object():
    from mvtn_amd.structures import Cameras
    from oracle import oracle as orc
    R, T = orc.look_at([10.0, 200.0], [20.0, -40.0], [2.2, 1.5])
    cam = Cameras(torch.from_numpy(R), torch.from_numpy(T))
    C = orc.camera_position([10.0, 200.0], [20.0, -40.0], [2.2, 1.5])
    assert np.allclose(cam.get_camera_center().numpy(), C, atol=1e-5)
    # util.py:604-618 plot_cameras: world-to-view inverse maps the view-space origin to the camera centre
    pts = cam.get_world_to_view_transform().inverse().transform_points(torch.zeros(1, 3))
    assert np.allclose(pts[:, 0].numpy(), C, atol=1e-5)


def test_mvrenderer_mode_switches():
    """models/renderer.py:153-171: random colour / random light only in training mode."""
    from mvtn_amd import MVRenderer
    r = MVRenderer(4, 32, pc_rendering=False, object_color="random", light_direction="random")
    r.train()
    np.random.seed(0)
    torch.manual_seed(0)
    c1, l1 = r.rendering_color(), r.light_direction(None, None, None)
    c2, l2 = r.rendering_color(), r.light_direction(None, None, None)
    assert not torch.equal(c1, c2) and l1 != l2 and len(l1) == 1 and len(l1[0]) == 3
    r.eval()
    assert torch.equal(r.rendering_color(), torch_color("white"))
    assert MVRenderer(4, 32, light_direction="fixed").light_direction(None, None, None) == ((0, 1.0, 0),)
    assert MVRenderer(4, 32, object_color="custom").rendering_color((0.1, 0.2, 0.3)) == (0.1, 0.2, 0.3)
    assert MVRenderer(4, 32, points_per_pixel=5).points_per_pixel == 5
    with pytest.raises(ValueError):
        MVRenderer(4, 32, points_per_pixel=0)
    with pytest.raises(ValueError):
        MVRenderer(4, 32, rotation_check="maybe")


def test_view_selectors_match_reference(golden):
    """models/mvtn.py:13-111 restated in mvtn_amd/views.py: buffers, (B,V) stride-(0,1) outputs, dispatch."""
    from mvtn_amd.views import RandomViewSelector, view_selector
    for key, g in golden["selectors"].items():
        cfg, V = key.split("_")
        sel = view_selector(cfg, int(V), canonical_elevation=30.0, canonical_distance=2.2).eval()
        az, el, di = sel(None, c_batch_size=3)
        assert list(az.shape) == g["shape"] and list(az.stride()) == g["stride"]
        assert np.allclose(az[0].numpy(), g["azim"], atol=1e-4) and np.allclose(el[0].numpy(), g["elev"], atol=1e-4)
        assert az[2].tolist() == az[0].tolist() and di[1].tolist() == g["dist"]
    torch.manual_seed(0)
    az, el, di = RandomViewSelector(8, transform_distance=True)(None, c_batch_size=5)
    assert az.shape == (5, 8) and float(az.min()) >= -180 and float(az.max()) < 180
    assert float(el.min()) >= -90 and float(el.max()) < 90 and float(di.min()) > 2.2 * 0.5 and float(di.max()) < 2.2 * 1.51
    noisy = view_selector("spherical", 12, input_view_noise=0.1).train()
    a1, _, _ = noisy(None, 2)
    a2, _, _ = noisy(None, 2)
    assert not torch.equal(a1, a2) and a1.shape == (2, 12)
    with pytest.raises(NotImplementedError):
        view_selector("learned_spherical", 12)


def test_non_square_images_are_rejected_without_gpu():
    from mvtn_amd import _lib
    lib = _lib.load()
    dummy = ctypes.c_void_p(16)  # never dereferenced: the size check comes first
    mb = _lib.MeshBatch(1, 3, 1, 1, 3, 0, 16, 16, 16, 16, None)
    mp = _lib.MeshParams(1, 8, 9, 0, 1, 0, 60.0, 0.5, 0.3, 0.2, 64.0)
    assert lib.mvr_render_meshes_forward(mb, mp, dummy, dummy, dummy, dummy, None, dummy, 1 << 20, None) == 4
    assert b"non-square" in lib.mvr_last_error()


def test_regularization_draws_follow_the_reference_order():
    """mvtn_amd.regularize.draw_regularization: same generator calls, in the same order, as ops.py:168-176 +
    util.py:193,224-231; per-image draws re-ordered from the reference's view-major flattening (m = v*B + b,
    super_batched_op dim=1) to memory order n = b*V + v."""
    from mvtn_amd import regularize as rg
    assert rg.crop_padding(224, 0.3) == 67 and rg.crop_padding(224, 0.0) == 0 and rg.crop_padding(128, 0.3) == 38
    B, V, H = 3, 4, 224
    torch.manual_seed(11)
    scale, flip, ci, cj, pad = rg.draw_regularization(B, V, H, torch.device("cpu"), dropout_p=0.4, augment_training=True, crop_ratio=0.3)
    torch.manual_seed(11)
    keep = torch.nn.functional.dropout3d(torch.ones(B, V, 1, 1, 1), p=0.4, training=True).reshape(-1)   # same mask shape, no 5-D warning
    flipped = torch.rand(B * V) < 0.5
    i = torch.randint(0, 2 * 67 + 1, (B * V,))
    j = torch.randint(0, 2 * 67 + 1, (B * V,))
    assert pad == 67 and torch.equal(scale, keep) and set(scale.tolist()) <= {0.0, float(torch.tensor(1.0) / torch.tensor(0.6))}
    x = torch.arange(B * V).reshape(B, V, 1, 1, 1).float()
    m_of_n = batch_tensor(x, dim=1, squeeze=True).reshape(-1).long()      # image m of the flattened batch holds memory image m_of_n[m]
    assert torch.equal(flip[m_of_n].bool(), flipped) and torch.equal(ci[m_of_n].long(), i) and torch.equal(cj[m_of_n].long(), j)
    assert flip.dtype == torch.uint8 and ci.dtype == torch.int32
    assert rg.draw_regularization(B, V, H, torch.device("cpu")) == (None, None, None, None, 0)
    img = torch.zeros(B, V, 3, 8, 8)
    assert rg.regualarize_rendered_views(img) is img            # defaults (config.yaml:42-43): identity, as in the reference
    with pytest.raises(ValueError):
        rg.regualarize_rendered_views(torch.zeros(2, 3, 8, 8))
    with pytest.raises(RuntimeError):                          # no CPU fallback
        rg.regualarize_rendered_views(img, dropout_p=0.5)


def test_regularize_c_abi_argument_validation_without_gpu():
    from mvtn_amd import _lib
    lib = _lib.load()
    buf = (ctypes.c_float * 16)()
    out = (ctypes.c_float * 16)()
    a, o = ctypes.addressof(buf), ctypes.addressof(out)
    assert lib.mvr_regularize_views_forward(None, 1, 1, 4, 4, None, None, None, None, 0, o, None) == 1
    assert lib.mvr_regularize_views_forward(a, 1, 1, 4, 4, None, None, None, None, 0, a, None) == 1          # in place
    assert lib.mvr_regularize_views_forward(a, 0, 1, 4, 4, None, None, None, None, 0, o, None) == 1
    assert lib.mvr_regularize_views_forward(a, 1, 1, 4, 4, None, None, a, None, 1, o, None) == 1            # crop_i without crop_j
    assert lib.mvr_regularize_views_backward(a, 1, 1, 4, 4, None, None, None, None, -1, o, None) == 1
    assert b"mvr_regularize_views" in lib.mvr_last_error()


def test_regularize_composition_is_pinned_to_the_reference():
    """The torch composition the GPU test compares the fused regulariser against (tests/reg_composition.py) reproduces,
    bit for bit and under the same seeds, what the REFERENCE's regualarize_rendered_views returned on the CPU
    (tests/golden/make_regularize_golden.py ran ops.py:168-176 itself): dropout mask, flips, replicate padding, windows
    and the view-major flattening of super_batched_op."""
    from reg_composition import regularize_composition
    fx = np.load(os.path.join(ROOT, "tests", "golden", "regularize_views.npz"))
    assert len(fx["names"]) >= 5
    for name in fx["names"]:
        B, V, C, H, p, aug, cr, seed = fx["cfg_" + str(name)]
        x = torch.from_numpy(fx["x_" + str(name)])
        torch.manual_seed(int(seed))
        got = regularize_composition(x, float(p), bool(aug), float(cr))
        assert np.array_equal(got.numpy(), fx["y_" + str(name)]), name


def test_flag_variants_of_the_params_struct_follow_their_source():
    """The autograd nodes ask for the same two flag variants of a params struct on every step; they are remembered on the
    struct (functional._copy_params) — and must be rebuilt when the caller changes a field of the source afterwards."""
    from mvtn_amd import _lib, functional as F
    p = F.mesh_params(12, 224, 1, (1.0, 1.0, 1.0), (1.0, 1.0, 1.0))
    a = F._copy_params(p, _lib.MVR_KEEP_PIXEL_LISTS)
    assert a is F._copy_params(p, _lib.MVR_KEEP_PIXEL_LISTS)                 # remembered
    assert a.flags == p.flags | _lib.MVR_KEEP_PIXEL_LISTS and a.strip_rows == p.strip_rows
    b = F._copy_params(p, _lib.MVR_WORKSPACE_FROM_FORWARD)
    assert b is not a and b.flags == p.flags | _lib.MVR_WORKSPACE_FROM_FORWARD
    p.strip_rows = 5                                                          # the caller edits the source
    a2 = F._copy_params(p, _lib.MVR_KEEP_PIXEL_LISTS)
    assert a2 is not a and a2.strip_rows == 5
    p.flags |= _lib.MVR_BACKWARD_GATHER
    assert F._copy_params(p, _lib.MVR_KEEP_PIXEL_LISTS).flags == p.flags | _lib.MVR_KEEP_PIXEL_LISTS


def test_renderer_bookkeeping_is_not_module_state():
    """The per-call status words of the device-side rotation check live in a plain object (assigning nn.Module attributes
    costs microseconds per forward) and do not show up as parameters, buffers or sub-modules."""
    from mvtn_amd import MVRenderer
    r = MVRenderer(4, 32, pc_rendering=False)
    assert len(list(r.parameters())) == 0 and len(list(r.buffers())) == 0 and len(list(r.children())) == 0
    assert r._st.calls == 0 and r._st.pending == [] and r._st.pool == []
    r.check_status()   # nothing pending: returns
    assert r.state_dict() == {}


def test_stale_library_is_refused_unless_opted_in(monkeypatch):
    """A library that cannot be rebuilt here and was built from other sources is refused (struct layouts move
    between builds); MVR_ALLOW_STALE_LIB=1 loads it with a warning; another ABI generation is refused either way."""
    import warnings
    from mvtn_amd import _lib, build as b
    real = _lib.load()
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.delenv("MVR_LIB", raising=False)
    monkeypatch.setattr(b, "is_current", lambda lib=None: False)
    monkeypatch.setattr(b, "build_hip", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no hipcc here")))
    monkeypatch.setattr(b, "source_hash", lambda: "0" * 64)
    monkeypatch.delenv("MVR_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(ImportError, match="MVR_ALLOW_STALE_LIB"):
        _lib.load()
    monkeypatch.setenv("MVR_ALLOW_STALE_LIB", "1")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert _lib.load() is not None
    assert any("other sources" in str(x.message) for x in w)
    # an ABI generation this package does not bind: refused even with the opt-in
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "MVR_ABI_VERSION", 9999)
    with pytest.raises(ImportError, match="ABI version"):
        _lib.load()
    monkeypatch.setattr(_lib, "_LIB", real)


def test_torch_operator_library_loads_and_refuses_cpu_tensors():
    """mvtn_amd/_mvr_torch_op.so (csrc/mvr_torch_op.cpp: the mesh step as one torch operator with a C++ autograd node) builds,
    registers `mvr::render_mesh_views` and receives the C ABI entry points; like every product path it has no CPU fallback:
    host tensors are refused with a message, not rendered."""
    import torch
    from mvtn_amd import _lib, functional as F
    if os.environ.get("MVR_TORCH_OP", "1") == "0":
        pytest.skip("MVR_TORCH_OP=0: the C++ node is switched off for this run")
    op = _lib.torch_op()
    assert op is not None, "the torch operator library did not build / load"
    assert "render_mesh_views" in str(torch.ops.mvr.render_mesh_views)
    p = F.mesh_params(2, 32, 1, (1.0, 1.0, 1.0), (1.0, 1.0, 1.0))
    pt = torch.frombuffer(p, dtype=torch.uint8)
    z = torch.zeros(2)
    verts, faces = torch.zeros(3, 3), torch.zeros(1, 3, dtype=torch.int32)
    off = torch.tensor([0, 3], dtype=torch.int32)
    foff = torch.tensor([0, 1], dtype=torch.int32)
    with pytest.raises(RuntimeError, match="on the GPU"):
        op.render_mesh_views(z, z, z, None, None, verts, faces, off, foff, None, 1, 3, pt, [], -1, 0, 0, False)
    with pytest.raises(RuntimeError, match="mvr_mesh_params has"):
        op.render_mesh_views(z, z, z, None, None, verts, faces, off, foff, None, 1, 3, pt[:-4].clone(), [], -1, 0, 0, False)
