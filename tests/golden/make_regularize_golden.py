"""Generates tests/golden/regularize_views.npz by running the REFERENCE's own `regualarize_rendered_views`
(ops.py:168-176 -> applied_transforms ops.py:138-146 -> util.RandomHorizontalFlip / util.RandomCrop, util.py:168-242) on
small seeded CPU tensors.  Run only where /root/reference exists; only these input/output vectors travel.

ops.py imports `torch._six` (gone from current torch) and `pytorch3d` (not installed) at module top; neither is touched
by the three functions used here, so empty stand-in MODULES satisfy the import (a call into them would raise).

    python tests/golden/make_regularize_golden.py
"""
import collections.abc
import os
import sys
import types

import numpy as np
import torch

for m in ("imageio", "trimesh"):
    sys.modules.setdefault(m, types.ModuleType(m))
six = types.ModuleType("torch._six")
six.inf = float("inf")
six.string_classes = (str,)
six.container_abcs = collections.abc
sys.modules["torch._six"] = six
p3d, p3dr = types.ModuleType("pytorch3d"), types.ModuleType("pytorch3d.renderer")


def _absent(*a, **k):
    raise RuntimeError("pytorch3d is not available: this generator must not reach the renderer")


p3dr.look_at_view_transform = _absent
p3d.renderer = p3dr
sys.modules["pytorch3d"], sys.modules["pytorch3d.renderer"] = p3d, p3dr
sys.path.insert(0, "/root/reference")
import ops  # noqa: E402

out = {}
cases = [("a", 2, 3, 3, 8, 0.0, True, 0.3, 11), ("b", 3, 2, 3, 9, 0.4, True, 0.3, 12), ("c", 2, 4, 1, 6, 0.5, False, 0.3, 13),
         ("d", 1, 3, 3, 5, 0.0, True, 1.5, 14), ("e", 2, 2, 3, 8, 0.25, True, 0.6, 15)]
names = []
for name, B, V, C, H, p, aug, cr, seed in cases:
    g = torch.Generator().manual_seed(100 + seed)
    x = torch.rand(B, V, C, H, H, generator=g)
    torch.manual_seed(seed)
    y = ops.regualarize_rendered_views(x.clone(), dropout_p=p, augment_training=aug, crop_ratio=cr)
    out["x_" + name], out["y_" + name] = x.numpy(), y.contiguous().numpy()
    out["cfg_" + name] = np.array([B, V, C, H, p, float(aug), cr, seed], np.float64)
    names.append(name)
out["names"] = np.array(names)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "regularize_views.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
