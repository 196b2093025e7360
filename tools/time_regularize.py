"""Times the fused view regulariser (row f3: ops.py:168-176) at the C2 shape against the op-by-op torch composition
the reference runs, and reports the fraction of the HBM roofline (algorithmic bytes: read + write each frame once)."""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtn_amd import regularize as rg                      # noqa: E402
from mvtn_amd.helpers import batch_tensor, unbatch_tensor  # noqa: E402

dev = torch.device("cuda:0")
B, V, C, H = 16, 12, 3, 224
x = torch.rand(B, V, C, H, H, device=dev)
G = torch.randn(B, V, C, H, H, device=dev)
nbytes = 2 * x.numel() * 4


def composition(x, p, crop_ratio):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = torch.nn.functional.dropout2d(x, p=p, training=True)
    flat = batch_tensor(x, dim=1, squeeze=True).clone()
    pad = int((1 + crop_ratio) * H) - H
    flipped = torch.rand(flat.shape[0]) < 0.5
    flat[flipped] = torch.flip(flat[flipped], [3])
    padded = torch.nn.ReplicationPad2d(pad)(flat)
    n = flat.shape[0]
    i = torch.randint(0, 2 * pad + 1, (n,))
    j = torch.randint(0, 2 * pad + 1, (n,))
    rows = torch.arange(H)[None] + i[:, None]
    cols = torch.arange(H)[None] + j[:, None]
    out = padded.permute(1, 0, 2, 3)[:, torch.arange(n)[:, None, None], rows[:, :, None], cols[:, None, :]].permute(1, 0, 2, 3)
    return unbatch_tensor(out, dim=1, unsqueeze=True, batch_size=B)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


draws = rg.draw_regularization(B, V, H, dev, 0.3, True, 0.3)
t_f = timed(lambda: rg.apply_regularization(x, *draws))
xg = x.clone().requires_grad_(True)
y = rg.apply_regularization(xg, *draws)
t_b = timed(lambda: torch.autograd.grad(y, xg, G, retain_graph=True))
t_call = timed(lambda: rg.regualarize_rendered_views(x, 0.3, True, 0.3))
t_ref = timed(lambda: composition(x, 0.3, 0.3), n=10)
yr = composition(xg, 0.3, 0.3)
t_refb = timed(lambda: torch.autograd.grad(yr, xg, G, retain_graph=True), n=10)
print("fused forward kernel        %7.1f us  = %.2f TB/s (%.0f %% of 8 TB/s) on %d MB" % (t_f, nbytes / t_f / 1e6, nbytes / t_f / 1e6 / 8 * 100, nbytes >> 20))
print("fused backward kernels      %7.1f us  = %.2f TB/s (%.0f %%)" % (t_b, nbytes / t_b / 1e6, nbytes / t_b / 1e6 / 8 * 100))
print("regualarize_rendered_views  %7.1f us  (draws + H2D of the draws + kernel)" % t_call)
print("torch composition forward   %7.1f us, backward %7.1f us" % (t_ref, t_refb))
