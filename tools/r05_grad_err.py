"""Developer tool (round 5): per-view relative errors of the HIP mesh backward against the torch.autograd fixture (C3 case)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mvtn_amd import synthetic as syn
import gpu_helpers as t4   # (round 6: the helpers of the refiled GPU tests)
fx = t4._fx("torch_restatement_grad.npz")
S = int(fx["c3_image_size"])
az, el, di = (fx["c3_" + k].astype(np.float32) for k in ("azim", "elev", "dist"))
vl, fl = zip(*[syn.synthetic_mesh("M", int(s)) for s in fx["c3_mesh_seeds"]])
G = np.stack([np.random.default_rng(int(s)).standard_normal((12, 3, S, S)).astype(np.float32) for s in fx["c3_g_seeds"]])
np.set_printoptions(linewidth=200, precision=2)
for gather in (False, True):
    grads, p2f = t4._hip_grads(list(vl), list(fl), az, el, di, S, G, True, gather)
    differ = (p2f.reshape(2, 12, S, S) != fx["c3_pix_to_face"].astype(np.int32)).reshape(2, 12, -1).sum(-1)
    print("gather" if gather else "table", "pixels on another face per view:\n", differ)
    for name, got in zip(("azim", "elev", "dist"), grads):
        ref = fx["c3_grad_" + name].astype(np.float64)
        print(name, "ref max %.1f; |err| / max|ref| x 1e4:\n" % np.abs(ref).max(), 1e4 * np.abs(got - ref) / np.abs(ref).max())
print("elev:\n", el, "\nazim:\n", az)

# ---- round 5: the same comparison with the pixels of badly conditioned (edge-on) faces masked out of the loss on both sides
if "c3_cond_bits" in fx:
    thr = fx["c3_cond_thresholds"]
    for k, t in enumerate(thr):
        mask = np.unpackbits(fx["c3_cond_bits"][k], axis=-1)[..., :S].astype(bool)      # (2, 12, S, S): cond > t
        Gm = G * (~mask)[:, :, None]
        grads, p2f = t4._hip_grads(list(vl), list(fl), az, el, di, S, Gm, True, False)
        errs = []
        for i, (name, got) in enumerate(zip(("azim", "elev", "dist"), grads)):
            ref = fx["c3_grad_%s_cond" % name][k].astype(np.float64)
            errs.append(np.abs(got - ref) / np.abs(ref).max())
        e = np.stack(errs)
        print("cond > %g masked (%d pixels, %.3f %% of the covered): max rel err %.2e; derivatives > 2e-4: %d of 72; worst views %s" % (
            t, mask.sum(), 100.0 * mask.sum() / (fx["c3_pix_to_face"] >= 0).sum(), e.max(), (e > 2e-4).sum(), np.argwhere(e > 2e-4).tolist()[:6]))
