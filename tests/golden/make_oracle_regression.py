"""Freezes the CPU oracle's own output on BASELINE config C1 (1 tier-M mesh, 4 circular views, 128^2) and on a small
point cloud, as hashes + a few statistics -> tests/golden/oracle_regression.json.  These are NOT reference outputs (the
reference's renderer cannot run here: "parity unpinned"); they pin the oracle against accidental drift between rounds.

    python tests/golden/make_oracle_regression.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mvtn_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def h(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def compute():
    v, f = syn.synthetic_mesh("M", 0)
    az, el, di = syn.circular_views(4, 30.0, 2.2)
    white = np.float32(1.0) / np.float32(1.00001)
    m = orc.render_meshes([v], [f], az[None], el[None], di[None], 128, obj_color=(white,) * 3, bg_color=(white,) * 3)
    pts = syn.surface_points(v, f, 2048, 0)[None]
    az2, el2, _ = syn.spherical_views(20)
    p = orc.render_points(pts, az2[None, :4], el2[None, :4], np.ones((1, 4), np.float32), 128, radius=0.02,
                          obj_color=(white,) * 3, bg_color=(0.0, 0.0, 0.0))
    return {
        "mesh_c1": {"pix_to_face_sha256": h(m["pix_to_face"]), "zbuf_sha256": h(m["zbuf"]), "R_sha256": h(m["R"]),
                    "covered": int((m["pix_to_face"] >= 0).sum()), "image_sum": float(m["images"].astype(np.float64).sum()),
                    "image_min": float(m["images"].min()), "image_max": float(m["images"].max())},
        "points": {"idx_sha256": h(p["idx"]), "dist2_sha256": h(p["dist2"]), "covered": int((p["idx"] >= 0).sum()),
                   "image_sum": float(p["images"].astype(np.float64).sum())},
    }


if __name__ == "__main__":
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.json")
    json.dump(compute(), open(dst, "w"), indent=1)
    print("wrote", dst)
