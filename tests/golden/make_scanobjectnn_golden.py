"""Generates tests/golden/scanobjectnn_bin.npz: synthetic ScanObjectNN-format .bin records (float32: count, then rows of
11 values x y z nx ny nz r g b instance semantic — or 3 values for the SUNCG variant) pushed through the REFERENCE's own
`ScanObjectNN.load_pc_file` (custom_dataset.py:553-575).  Run only where /root/reference exists; the vectors travel.

custom_dataset.py imports h5py / pandas / trimesh / pytorch3d at module top; none is touched by load_pc_file, so empty
stand-in MODULES satisfy the import.      python tests/golden/make_scanobjectnn_golden.py
"""
import os
import sys
import tempfile
import types

import numpy as np


class _Any(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (), {})


for m in ("imageio", "trimesh", "h5py", "pandas", "pytorch3d", "pytorch3d.io", "pytorch3d.structures", "pytorch3d.renderer",
          "pytorch3d.renderer.mesh"):
    sys.modules.setdefault(m, _Any(m))
sys.path.insert(0, "/root/reference")
import custom_dataset as cd  # noqa: E402

rng = np.random.default_rng(0)
out, names = {}, []
with tempfile.TemporaryDirectory() as tmp:
    for name, n, suncg, variant in (("obj", 400, False, "obj_only"), ("bg", 300, False, "with_bg"), ("suncg", 250, True, "with_bg"),
                                    ("tie", 120, False, "obj_only")):
        cols = 3 if suncg else 11
        rows = rng.normal(size=(n, cols)).astype(np.float32)
        if not suncg:
            inst = rng.choice([0, 1, 2, 5, 7, 9], size=n, p=[0.2, 0.1, 0.1, 0.35, 0.15, 0.1]).astype(np.float32)
            if name == "tie":
                inst = np.repeat([3.0, 8.0, 1.0], n // 3).astype(np.float32)     # two foreground instances of equal size
            rows[:, -1] = inst
        raw = np.concatenate([[np.float32(n)], rows.reshape(-1)]).astype(np.float32)
        path = os.path.join(tmp, name + ".bin")
        raw.tofile(path)
        pts = cd.ScanObjectNN.load_pc_file(types.SimpleNamespace(suncg=suncg, variant=variant), path)
        out["raw_" + name], out["pts_" + name] = raw, np.asarray(pts, np.float32)
        out["cfg_" + name] = np.array([int(suncg), int(variant == "with_bg")])
        names.append(name)
out["names"] = np.array(names)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scanobjectnn_bin.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes", {k: out["pts_" + k].shape for k in names})
