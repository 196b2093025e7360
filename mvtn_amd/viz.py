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


def save_cameras(cameras, save_path, scale=0.22, dpi=200):
    """util.py:604-664: wireframe of every camera in world space."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    a = 0.5 * torch.tensor([-2.0, 1.5, 4.0])
    b = 0.5 * torch.tensor([2.0, 1.5, 4.0])
    c = 0.5 * torch.tensor([-2.0, -1.5, 4.0])
    d = 0.5 * torch.tensor([2.0, -1.5, 4.0])
    C = torch.zeros(3)
    wire = torch.stack([a, b, d, c, a, C, b, d, C, c, C]) * scale
    world = cameras.get_world_to_view_transform().inverse().transform_points(wire.to(cameras.R.device)).cpu()
    fig = plt.figure()
    ax = fig.add_subplot(projection="3d")
    for w in world:
        x, z, y = w.numpy().T  # the reference plots (x, z, y)
        ax.plot(x, y, z, color="#FF7D1E", linewidth=0.6)
    ax.set_xlabel("x"); ax.set_ylabel("z"); ax.set_zlabel("y")
    fig.savefig(save_path, dpi=dpi)
    plt.close(fig)
