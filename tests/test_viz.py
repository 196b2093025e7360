"""render_and_save helpers: grid layout and PNG encoding, CPU only."""
import struct
import zlib

import numpy as np
import pytest
import torch

from mvtn_amd import viz
from mvtn_amd.structures import Cameras


def _decode_png(path):
    """Minimal decoder for what write_png emits (8-bit RGB, filter 0 on every row)."""
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, {}
    while pos < len(data):
        n, = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks.setdefault(tag, b"")
        chunks[tag] += body
        pos += 12 + n
    w, h, depth, ctype = struct.unpack(">IIBB", chunks[b"IHDR"][:10])
    assert (depth, ctype) == (8, 2)
    raw = zlib.decompress(chunks[b"IDAT"])
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


def test_make_grid_matches_torchvision_layout():
    imgs = torch.arange(5 * 3 * 4 * 6, dtype=torch.float32).reshape(5, 3, 4, 6)
    g = viz.make_grid(imgs, nrow=3, padding=2, pad_value=-1.0)
    assert g.shape == (3, 2 * (4 + 2) + 2, 3 * (6 + 2) + 2)
    assert torch.equal(g[:, 2:6, 2:8], imgs[0]) and torch.equal(g[:, 2:6, 10:16], imgs[1])
    assert torch.equal(g[:, 8:12, 10:16], imgs[4])                      # second row, second column
    assert (g[:, :2] == -1).all() and (g[:, 8:12, 18:24] == -1).all()    # padding and the unused sixth cell


def test_save_grid_png_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    imgs = torch.from_numpy(rng.uniform(-0.2, 1.2, (4, 3, 8, 8)).astype(np.float32))   # values outside [0, 1] are clamped
    path = str(tmp_path / "grid.png")
    viz.save_grid(imgs, path, nrow=4)
    px = _decode_png(path)
    want = (viz.make_grid(imgs, nrow=4).clamp(0, 1) * 255).to(torch.uint8).permute(1, 2, 0).numpy()
    assert px.shape == want.shape == (8 + 4, 4 * 10 + 2, 3)
    assert np.array_equal(px, want)


def test_cameras_world_transform_inverts_the_view_transform():
    # what save_cameras relies on (util.py:604-618): points given in camera space come back in world space
    R = torch.eye(3)[None].repeat(2, 1, 1)
    R[1] = torch.tensor([[0.0, 0, 1], [0, 1, 0], [-1, 0, 0]])
    T = torch.tensor([[0.0, 0, 2.2], [0.1, -0.2, 1.5]])
    cams = Cameras(R, T, "perspective")
    p_cam = torch.tensor([[0.0, 0, 0], [0.3, -0.1, 1.0]])
    world = cams.get_world_to_view_transform().inverse().transform_points(p_cam)
    back = world @ R + T[:, None, :]
    assert torch.allclose(back, p_cam[None].expand(2, -1, -1), atol=1e-6)


def test_save_cameras_writes_a_plot(tmp_path):
    pytest = __import__("pytest")
    pytest.importorskip("matplotlib")
    R = torch.eye(3)[None].repeat(3, 1, 1)
    T = torch.tensor([[0.0, 0, 2.2], [0.1, 0.0, 2.0], [0.0, -0.1, 1.8]])
    path = str(tmp_path / "cams.png")
    viz.save_cameras(Cameras(R, T, "perspective"), path, scale=0.22, dpi=50)
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n" and len(data) > 1000


def test_save_grid_honours_the_extension(tmp_path):
    """util.py:667-673 lets imageio pick the format from the file name: a .jpg must be a JPEG (r1 wrote PNG bytes)."""
    pytest.importorskip("PIL")
    from PIL import Image
    imgs = torch.rand(3, 3, 8, 8)
    path = str(tmp_path / "grid.jpg")
    viz.save_grid(imgs, path, nrow=3)
    with open(path, "rb") as f:
        assert f.read(3) == b"\xff\xd8\xff"
    assert Image.open(path).size == (3 * 10 + 2, 12)


def test_camera_plot_draws_every_camera_where_it_is():
    """util.py:604-664 (missing 4: the plot was only checked by its file size).  The figure's line handles are read
    back: one polyline per camera, 12 points each (util.py:588-601: the last one, (0, 0, 3) scale, ends the segment along the viewing direction); its apex (point 5 of the wireframe, the camera-space origin) sits at the
    camera centre -T R^T; the plot's (x, y, z) are the world's (x, z, y); the frustum's four corners keep their camera-space
    distances from the apex (a rigid transform); limits, origin marker and colours are the reference's."""
    pytest = __import__("pytest")
    pytest.importorskip("matplotlib")
    import matplotlib.pyplot as plt
    import numpy as np
    from mvtn_amd import functional as F  # noqa: F401  (import check only: the plot needs no GPU)
    az, el, di = torch.tensor([0.0, 90.0, 200.0, -45.0]), torch.tensor([30.0, -20.0, 60.0, 0.0]), torch.tensor([2.2, 1.5, 3.0, 2.0])
    # look-at in torch: C = d (cos e sin a, sin e, cos e cos a); z = -C / |C|, x = up x z, y = z x x; R columns = axes
    a, e = torch.deg2rad(az), torch.deg2rad(el)
    C = torch.stack([di * torch.cos(e) * torch.sin(a), di * torch.sin(e), di * torch.cos(e) * torch.cos(a)], 1)
    z = torch.nn.functional.normalize(-C, dim=1)
    x = torch.nn.functional.normalize(torch.cross(torch.tensor([[0.0, 1.0, 0.0]]).expand_as(z), z, dim=1), dim=1)
    y = torch.nn.functional.normalize(torch.cross(z, x, dim=1), dim=1)
    R = torch.stack([x, y, z], dim=2)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    fig, ax, handles = viz.cameras_figure(Cameras(R, T, "perspective"), scale=0.22)
    try:
        assert len(handles) == 4
        wire = viz.camera_wireframe(0.22).numpy()
        for n, h in enumerate(handles):
            xs, ys, zs = (np.asarray(v, dtype=np.float64) for v in h.get_data_3d())
            assert xs.shape == (12,)
            world = np.stack([xs, zs, ys], 1)                      # plot (x, y, z) = world (x, z, y)
            assert np.allclose(world[5], C[n].numpy(), atol=1e-5)   # the apex is the camera centre
            assert np.allclose(world[8], world[5]) and np.allclose(world[10], world[5])
            for k in (0, 1, 2, 3):                                 # the image-plane corners: rigid transform of the wireframe
                assert abs(np.linalg.norm(world[k] - world[5]) - np.linalg.norm(wire[k])) < 1e-5
            centre_of_plane = world[[0, 1, 2, 3]].mean(0)          # lies on the optical axis, towards the origin the cameras look at
            assert np.linalg.norm(centre_of_plane) < np.linalg.norm(world[5])
            # the twelfth point: 3 * scale along the viewing direction (+z of the camera, towards the origin it looks at)
            ahead = world[11] - world[5]
            assert abs(np.linalg.norm(ahead) - 3 * 0.22) < 1e-5 and np.allclose(ahead / np.linalg.norm(ahead), -C[n].numpy() / np.linalg.norm(C[n].numpy()), atol=1e-5)
            assert h.get_color() == "blue" and abs(h.get_linewidth() - 0.3) < 1e-9
        assert tuple(ax.get_xlim()) == (-3.0, 3.0) and tuple(ax.get_ylim()) == (-3.0, 3.0) and tuple(ax.get_zlim()) == (-1.8, 1.8)
        assert len(ax.collections) == 1                            # the red origin marker
    finally:
        plt.close(fig)
