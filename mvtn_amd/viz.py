"""render_and_save helpers (models/renderer.py:200-207): image grid + camera plot.  The reference uses
torchvision.make_grid + imageio (util.py:667-673) and matplotlib (util.py:604-664); here the grid is
assembled with torch and written as PNG with the standard library, the camera plot needs matplotlib."""
import struct
import zlib

import torch


def make_grid(images, nrow=8, padding=2, pad_value=0.0):
    """(K, C, H, W) -> (C, rows*(H+pad)+pad, cols*(W+pad)+pad), torchvision.utils.make_grid layout."""
    K, C, H, W = images.shape
    cols = min(nrow, K)
    rows = (K + cols - 1) // cols
    grid = images.new_full((C, rows * (H + padding) + padding, cols * (W + padding) + padding), pad_value)
    for k in range(K):
        r, c = divmod(k, cols)
        y, x = padding + r * (H + padding), padding + c * (W + padding)
        grid[:, y:y + H, x:x + W] = images[k]
    return grid


def write_png(path, rgb_u8):
    """rgb_u8: (H, W, 3) uint8 torch tensor or numpy array."""
    if torch.is_tensor(rgb_u8):
        rgb_u8 = rgb_u8.cpu().numpy()
    h, w, _ = rgb_u8.shape
    raw = b"".join(b"\x00" + rgb_u8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def save_grid(image_batch, save_path, nrow=8):
    """util.py:667-673: one row of views as an image file.  The reference hands the array to imageio, which picks the
    format from the extension; here ".png" is written with the standard library and any other extension (".jpg", ...)
    goes through Pillow — a file is never written in a format its name does not say."""
    import os
    grid = make_grid(image_batch.detach().float().cpu(), nrow=nrow)
    rgb = (grid.clamp(0, 1) * 255).to(torch.uint8).permute(1, 2, 0).contiguous()
    ext = os.path.splitext(str(save_path))[1].lower()
    if ext == ".png":
        write_png(save_path, rgb)
        return
    try:
        from PIL import Image
    except ImportError:
        raise ValueError("save_grid: writing %r needs Pillow (only .png is written without it)" % ext)
    Image.fromarray(rgb.numpy()).save(save_path)


def camera_wireframe(scale=0.3):
    """util.py:588-601 get_camera_wireframe: the 12-point polyline of a camera symbol in camera space — the frustum (apex =
    point 5) and, last, the segment from the apex along the viewing direction to (0, 0, 3)."""
    a = 0.5 * torch.tensor([-2.0, 1.5, 4.0])
    b = 0.5 * torch.tensor([2.0, 1.5, 4.0])
    c = 0.5 * torch.tensor([-2.0, -1.5, 4.0])
    d = 0.5 * torch.tensor([2.0, -1.5, 4.0])
    C = torch.zeros(3)
    ahead = torch.tensor([0.0, 0.0, 3.0])
    return torch.stack([a, b, d, c, a, C, b, d, C, c, C, ahead]) * scale


def plot_cameras(ax, cameras, color="blue", scale=0.3):
    """util.py:604-618: every camera's wireframe in world space on a 3-D axis; the plot's (x, y, z) are the world's
    (x, z, y) — "the Z and Y axes are flipped intentionally here" — line width 0.3.  Returns the line handles."""
    wire = camera_wireframe(scale).to(cameras.R.device)
    world = cameras.get_world_to_view_transform().inverse().transform_points(wire).detach().cpu()
    handles = []
    for w in world:
        x_, z_, y_ = w.numpy().T.astype(float)
        (h,) = ax.plot(x_, y_, z_, color=color, linewidth=0.3)
        handles.append(h)
    return handles


def cameras_figure(cameras, scale=0.22):
    """The figure util.py:655-664 save_cameras draws: axis limits (-3, 3) x (-3, 3) x (-1.8, 1.8), the world origin as a red
    marker, the cameras in blue.  Returns (figure, axis, line handles)."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    fig = plt.figure()
    ax = fig.add_subplot(projection="3d")
    ax.set_xlim(-3.0, 3.0)
    ax.set_ylim(-3.0, 3.0)
    ax.set_zlim(-1.8, 1.8)
    ax.scatter(xs=[0], ys=[0], zs=[0], linewidth=3, c="r")
    handles = plot_cameras(ax, cameras, color="blue", scale=scale)
    return fig, ax, handles


def save_cameras(cameras, save_path, scale=0.22, dpi=200):
    """util.py:655-664."""
    import matplotlib.pyplot as plt
    fig, _, _ = cameras_figure(cameras, scale)
    fig.savefig(save_path, dpi=dpi)
    plt.close(fig)
