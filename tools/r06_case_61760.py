"""Round 6: the stress case (tests/stress_parity.py, seed 41, case 61760) that exposed a latent flaw of the in-kernel exact settle once a
strip may record more than 2 candidates per pixel — a 96-face soup of double-sided quads seen from 0.9 units away; inputs in
tests/golden/stress_case_61760.npz (made by the stress script's own draws).     python3 tools/r06_case_61760.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import oracle
from mvtn_amd import functional as F
from mvtn_amd.structures import Meshes, PackedMeshes
d = np.load(os.path.join(ROOT, "tests", "golden", "stress_case_61760.npz"))
vl, fl = [d["v0"], d["v1"]], [d["f0"], d["f1"]]
az, el, di, size = d["az"], d["el"], d["di"], int(d["size"])
dev = torch.device("cuda:0")
obj, bg = (0.9, 0.8, 0.7), (0.1, 0.2, 0.3)
exact_sum = any(bool(oracle.cancelled_normal_vertices(v_, f_).any()) for v_, f_ in zip(vl, fl))
oracle.set_normals_exact_sum(exact_sum)
want = oracle.render_meshes(vl, fl, az, el, di, size, obj_color=obj, bg_color=bg, mode=1, z_clip=0.0)
oracle.set_normals_exact_sum(False)
t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)
R, T = F.look_at_view_transform(t(di.reshape(-1)), t(el.reshape(-1)), t(az.reshape(-1)))
params = F.mesh_params(2, size, 1, obj, bg, z_clip=0.0)
img, p2f, st = F.render_meshes_fragments(packed, R, T, t([[0, 1, 0]]), params, return_stats=True)
torch.cuda.synchronize()
p2f = p2f.cpu().numpy()
diff = np.argwhere(p2f != want["pix_to_face"])
print("stats", [int(x) for x in st.cpu()], "map diffs", len(diff))
for n_, y_, x_ in diff[:12]:
    print("  view %d pixel (%d, %d): library %d oracle %d" % (n_, y_, x_, p2f[n_, y_, x_], want["pix_to_face"][n_, y_, x_]))
sys.exit(1 if len(diff) else 0)
