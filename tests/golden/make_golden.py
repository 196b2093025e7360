"""Generates tests/golden/caller_side.json by importing the REFERENCE's own caller-side
code (util.py, models/mvtn.py) in the build container.  Run only where /root/reference
exists; the reference itself never travels — only these small input/output vectors do.

The renderer arithmetic (PyTorch3D) cannot be imported (not installed, not vendored), so
no rendered-pixel golden exists: the pixel oracle stays "parity unpinned".

    python tests/golden/make_golden.py
"""
import json
import os
import sys
import types

import numpy as np
import torch

for m in ("imageio", "trimesh"):  # util.py imports them at module top; unused on this path
    sys.modules.setdefault(m, types.ModuleType(m))
sys.path.insert(0, "/root/reference")
import util  # noqa: E402
from models.mvtn import MVTN  # noqa: E402

out = {}

# util.py:359-391
out["unit_spherical_grid"] = {}
for n in (1, 2, 4, 6, 8, 12, 20, 24):
    az, el = util.unit_spherical_grid(n)
    out["unit_spherical_grid"][str(n)] = {"azim": [float(x) for x in az], "elev": [float(x) for x in el]}

# models/mvtn.py view selectors (buffers expanded over the batch)
out["selectors"] = {}
for cfg, V in (("circular", 4), ("circular", 12), ("spherical", 4), ("spherical", 12), ("spherical", 20)):
    torch.manual_seed(0)
    m = MVTN(V, views_config=cfg, canonical_elevation=30.0, canonical_distance=2.2,
             transform_distance=False, input_view_noise=0.0, shape_extractor="PointNet",
             screatch_feature_extractor=True).eval()
    az, el, di = m(None, c_batch_size=3)
    out["selectors"]["%s_%d" % (cfg, V)] = {
        "azim": az[0].tolist(), "elev": el[0].tolist(), "dist": di[0].tolist(),
        "shape": list(az.shape), "stride": list(az.stride())}

# util.py:509-521 batch_tensor as used at models/renderer.py:79 : n = b*V + v
x = torch.arange(6.0).reshape(2, 3)
out["batch_tensor_2x3"] = util.batch_tensor(x.T, dim=1, squeeze=True).tolist()
xe = torch.arange(4.0).expand(3, 4)
out["batch_tensor_expanded_3x4"] = util.batch_tensor(xe.T, dim=1, squeeze=True).tolist()

# output re-layout of models/renderer.py:109-113 replayed with util.unbatch_tensor
B, V, H, W = 3, 4, 5, 6
raw = torch.arange(B * V * H * W * 4, dtype=torch.float32).reshape(B * V, H, W, 4)
img = util.unbatch_tensor(raw.clone(), batch_size=V, dim=1, unsqueeze=True).transpose(0, 1)
img = img[..., 0:3].transpose(2, 4).transpose(3, 4)
ok = all(torch.equal(img[b, v, c], raw[b * V + v, :, :, c]) for b in range(B) for v in range(V) for c in range(3))
out["relayout"] = {"shape": list(img.shape), "matches_n_eq_bV_plus_v": bool(ok)}

# util.py:314-334 torch_color
out["torch_color"] = {}
for name in ("white", "red", "green", "blue", "black"):
    out["torch_color"][name] = {
        "plain": [float(c) for c in util.torch_color(name)],
        "max_lightness": [float(np.float32(c)) for c in util.torch_color(name, max_lightness=True)],
        "max_lightness_bits": [int(np.float32(c).view(np.uint32)) for c in util.torch_color(name, max_lightness=True).numpy()],
    }

# util.py:403-420 check_valid_rotation_matrix
eye = torch.eye(3)[None]
bad = eye.clone(); bad[0, 0, 0] = 1.0001
bad2 = eye.clone(); bad2[0, 0, 1] = 2e-6
ok2 = eye.clone(); ok2[0, 0, 1] = 5e-7
ok3 = eye.clone(); ok3[0, 0, 0] = 1.000005
out["check_valid_rotation_matrix"] = {
    "identity": bool(util.check_valid_rotation_matrix(eye)),
    "diag_1.0001": bool(util.check_valid_rotation_matrix(bad)),
    "offdiag_2e-6": bool(util.check_valid_rotation_matrix(bad2)),
    "offdiag_5e-7": bool(util.check_valid_rotation_matrix(ok2)),
    "diag_1.000005": bool(util.check_valid_rotation_matrix(ok3)),
}

# util.py:437-451 torch_center_and_normalize(p="2") on a fixed point set
rng = np.random.default_rng(0)
pts = rng.normal(size=(16, 3)).astype(np.float32)
out["center_and_normalize"] = {"in": pts.tolist(),
                               "out": util.torch_center_and_normalize(torch.from_numpy(pts), p="2").tolist()}

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "caller_side.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
