import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from gpu_helpers import _fx, _hip_grads
from mvtn_amd import synthetic as syn
fx = _fx("torch_restatement_grad.npz"); f32 = _fx("torch_restatement_grad32.npz")
S = int(fx["c3_image_size"])
az, el, di = (fx["c3_" + k].astype(np.float32) for k in ("azim", "elev", "dist"))
vl, fl = zip(*[syn.synthetic_mesh("M", int(s)) for s in fx["c3_mesh_seeds"]])
G = np.stack([np.random.default_rng(int(s)).standard_normal((12, 3, S, S)).astype(np.float32) for s in fx["c3_g_seeds"]])
grads, p2f = _hip_grads(list(vl), list(fl), az, el, di, S, G, True, False)
np.set_printoptions(precision=2, linewidth=200)
for name, got in zip(("azim", "elev", "dist"), grads):
    ref = fx["c3_grad_" + name].astype(np.float64); sc = np.abs(ref).max()
    e_lib = np.abs(got - ref) / sc; e_t = np.abs(f32["c3_grad_%s_f32" % name] - ref) / sc
    print(name, "scale", sc); print(" lib", e_lib.reshape(-1)); print(" t32", e_t.reshape(-1))
    print(" mean lib %.2e t32 %.2e; rms lib %.2e t32 %.2e" % (e_lib.mean(), e_t.mean(), np.sqrt((e_lib**2).mean()), np.sqrt((e_t**2).mean())))
