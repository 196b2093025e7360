"""Developer tool: where does k_mesh_raster_shade spend its cycles?  Builds a DIAGNOSTIC copy of the
library (-DMVR_DIAG: per-workgroup s_memtime stamps into a buffer of their own) and prints the share of
scan / small-face / large-face / shade phases for the C2 workload.  Not part of the product or the tests.

    python tools/diag_phases.py [--strip-rows N]
"""
import argparse
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--strip-rows", type=int, default=0)
ap.add_argument("--tier", default="M")
args = ap.parse_args()

diag = os.path.join(ROOT, "mvtn_amd", "libmvr_hip_diag.so")
from mvtn_amd import build as b  # noqa: E402
b.build_hip(out=diag, extra_flags=["-DMVR_DIAG"] + [x for x in os.environ.get("MVR_DIAG_EXTRA", "").split() if x])
os.environ["MVR_LIB"] = diag

import numpy as np  # noqa: E402
import torch  # noqa: E402
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, _lib, synthetic as syn  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh(args.tier, i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="fixed", rotation_check="off", strip_rows=args.strip_rows).eval()
for _ in range(3):
    r(packed, None, ta, te, td)
nblocks = 8 * ((B * V + 7) // 8) * 64
buf = torch.zeros((1 << 20) + 8 * 4096, dtype=torch.int64, device=dev)   # raster stamps [0, 8 nblocks), pre-pass stamps from 1 << 20
lib.mvr_diag_set_buffer.argtypes = [ctypes.c_void_p]
lib.mvr_diag_set_buffer(buf.data_ptr())
r(packed, None, ta, te, td)
torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(-1, 8)[:nblocks]   # the pre-pass stamps live further up in the same buffer
d = d[d[:, 2] > 0]
tot = (d[:, 2] - d[:, 0]).astype(np.float64)
print("workgroups: %d   cycles/workgroup (s_memtime ticks @100MHz?): mean %.0f  max %.0f" % (len(d), tot.mean(), tot.max()))
print("  raster loop total  : %5.1f %%" % (100 * (d[:, 1] - d[:, 0]).sum() / tot.sum()))
print("    scan             : %5.1f %%" % (100 * d[:, 3].sum() / tot.sum()))
print("    setup + prefix   : %5.1f %%" % (100 * d[:, 4].sum() / tot.sum()))
print("    candidate loop   : %5.1f %%" % (100 * d[:, 5].sum() / tot.sum()))
print("  shade + store      : %5.1f %%" % (100 * (d[:, 2] - d[:, 1]).sum() / tot.sum()))
print("  per workgroup: listed faces mean %.1f max %d | candidates mean %.1f max %d" % (d[:, 6].mean(), d[:, 6].max(), d[:, 7].mean(), d[:, 7].max()))
nl, nc = d[:, 6], d[:, 7]
print("  listed-faces histogram (per workgroup):")
for lo, hi in [(0, 1), (1, 65), (65, 129), (129, 257), (257, 321), (321, 385), (385, 513), (513, 769), (769, 100000)]:
    m = (nl >= lo) & (nl < hi)
    print("    [%4d, %5d): %5.1f %% of workgroups, %5.1f %% of cycles, candidates/face %.1f" %
          (lo, hi, 100 * m.mean(), 100 * tot[m].sum() / tot.sum(), nc[m].sum() / max(1, nl[m].sum())))
print("  cycles per workgroup by quartile of listed faces:", [int(np.mean(tot[(nl >= a) & (nl < b)])) if ((nl >= a) & (nl < b)).any() else 0
                                                               for a, b in [(0, 1), (1, 129), (129, 257), (257, 513), (513, 10**6)]])
