"""The literal torch composition of the reference's `regualarize_rendered_views` (ops.py:168-176 + applied_transforms
ops.py:138-146 + util.RandomHorizontalFlip / RandomCrop, util.py:168-242), spelled with plain torch ops and drawing from
the generators in the reference's order.  Test infrastructure: pinned to the reference's own outputs on the CPU
(tests/test_host_logic.py, fixture tests/golden/regularize_views.npz) and used as the expectation of the fused HIP pass on
the GPU (tests/test_gpu_mesh_forward.py)."""
import warnings

import torch

from mvtn_amd.helpers import batch_tensor, unbatch_tensor


def regularize_composition(x, p, augment, crop_ratio):
    """What ops.py:168-176 does, spelled with plain torch ops (dropout over whole views; per image of the view-major
    flattened batch: masked column flip, replicate padding, window gather), drawing from the generators in that order."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        x = torch.nn.functional.dropout2d(x, p=p, training=True)
    if not augment:
        return x
    B, V, C, H, W = x.shape
    flat = batch_tensor(x, dim=1, squeeze=True)
    pad = int((1 + crop_ratio) * H) - H
    flipped = (torch.rand(flat.shape[0]) < 0.5).to(x.device)
    flat = torch.where(flipped[:, None, None, None], torch.flip(flat, [3]), flat)
    padded = torch.nn.ReplicationPad2d(pad)(flat)
    i = torch.randint(0, 2 * pad + 1, (flat.shape[0],))
    j = torch.randint(0, 2 * pad + 1, (flat.shape[0],))
    out = torch.stack([padded[m, :, int(i[m]):int(i[m]) + H, int(j[m]):int(j[m]) + H] for m in range(flat.shape[0])])
    return unbatch_tensor(out, dim=1, unsqueeze=True, batch_size=B)
