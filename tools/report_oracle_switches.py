"""How many pixels the oracle's switches move (rounds 3-6: normals form, K[0][0] +-1 ulp, edge_fma, projection_bmm), on C1 (BASELINE configs[0]: one tier-M mesh,
4 circular views, 128 x 128) and on the whole of C2 (16 tier-M meshes x 12 spherical views, 224 x 224).  on the whole of C2 and (round 6) on three CAD-like objects (cad_mesh seeds 0-2, 5 120 faces, 12 spherical views, 224 x 224), where
the pixels that change face are classified: swaps between a face and its coincident reversed twin, or anything else.  CPU only
(~4 min on 8 cores).       python3 tools/report_oracle_switches.py > profiles/r06_oracle_switches.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvtn_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
orc.set_threads(os.cpu_count() or 1)
S32 = float(np.float32(np.float32(2.0) / (np.float32(2.0) * np.tan(np.float32(np.float32(np.pi / 180 * 60.0) / np.float32(2.0))))))
S_TORCH_CPU = float.fromhex("0x1.bb67b0p+0")   # K[0][0] from torch.tan on this container's CPU (tests/golden/make_torch_restatement.py)


def render(vl, fl, az, el, di, S):
    return orc.render_meshes(vl, fl, az, el, di, S, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3, mode=0)


def report(name, vl, fl, az, el, di, S, twins=False):
    orc.set_options()
    base = render(vl, fl, az, el, di, S)
    cov = base["pix_to_face"] >= 0
    print("%s: %d pixels, %d covered" % (name, cov.size, int(cov.sum())))
    for label, opts, fma, bmm in (("normals as ONE cross product per face", (False, 0.0), False, False),
                                  ("s = K[0][0] of torch CPU (0x1.bb67b0p+0, +1 ulp)", (True, S_TORCH_CPU), False, False),
                                  ("s = numpy tanf route (%s, -1 ulp)" % float(np.float32(S32)).hex(), (True, S32), False, False),
                                  ("edge_fma: face test contracted as nvcc would (switch #3)", (True, 0.0), True, False),
                                  ("projection_bmm: world -> view as a K = 4 GPU matrix product sums it (switch #5)", (True, 0.0), False, True)):
        orc.set_options(*opts)
        orc.set_edge_fma(fma)
        orc.set_projection_bmm(bmm)
        o = render(vl, fl, az, el, di, S)
        orc.set_edge_fma(False)
        orc.set_projection_bmm(False)
        moved = o["pix_to_face"] != base["pix_to_face"]
        d = np.abs(o["images"] - base["images"]).max(2).reshape(moved.shape)
        extra = ""
        if twins and moved.any():   # (one object per call) how many of the changes are swaps between a face and its coincident twin
            v, f = vl[0], fl[0]
            key = lambda i: tuple(sorted(map(tuple, v[f[i]].tolist())))   # noqa: E731
            n_twin = sum(1 for x, y in zip(base["pix_to_face"][moved], o["pix_to_face"][moved]) if x >= 0 and y >= 0 and key(x) == key(y))
            extra = "; of the changes %d are face <-> coincident twin, %d anything else" % (n_twin, int(moved.sum()) - n_twin)
        print("   %-82s pixel -> face changes at %5d pixels; max colour change %.2e overall, %.2e on unchanged pixels; pixels > 1e-4: %d%s" % (
            label, int(moved.sum()), d.max(), d[~moved].max(), int((d > 1e-4).sum()), extra))
    orc.set_options()


v, f = syn.synthetic_mesh("M", 0)
az, el, di = syn.circular_views(4, 30.0, 2.2)
report("C1", [v], [f], az[None], el[None], di[None], 128)
B, V = 16, 12
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
az, el, di = syn.spherical_views(V)
report("C2", list(vl), list(fl), np.tile(az, (B, 1)), np.tile(el, (B, 1)), np.tile(di, (B, 1)), 224)

for seed in range(3):   # round 6: CAD-like objects (coincident reversed shells on shared and on duplicated vertices)
    v, f = syn.cad_mesh(seed)
    report("CAD-like object, seed %d (%d faces)" % (seed, len(f)), [v], [f], az[None], el[None], di[None], 224, twins=True)
