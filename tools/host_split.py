"""Developer tool: host time of one C3 training step split into forward call and backward call (perf_counter around each,
queue kept short), for the C++ operator and the Python autograd node.  Usage: python tools/host_split.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, _lib, synthetic as syn  # noqa: E402

dev = torch.device("cuda:0")
B, V, H = 16, 12, 224
vl, fl = zip(*[syn.synthetic_mesh("M", i) for i in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
az, el, di = syn.spherical_views(V)
ta, te, td = (torch.from_numpy(np.tile(x, (B, 1))).to(dev).requires_grad_(True) for x in (az, el, di))
r = MVRenderer(V, H, pc_rendering=False, light_direction="relative").eval()
up = torch.randn(B, V, 3, H, H, device=dev)
op = _lib.torch_op()


def measure(label, use_op, mt=True):
    _lib._TORCH_OP = op if use_op else None
    torch.autograd.set_multithreading_enabled(mt)
    for rep in range(3):
        tf = tb = 0.0
        for _ in range(40):
            for _ in range(10):
                ta.grad = te.grad = td.grad = None
                t0 = time.perf_counter()
                img, _ = r(packed, None, ta, te, td)
                t1 = time.perf_counter()
                img.backward(up)
                t2 = time.perf_counter()
                tf += t1 - t0
                tb += t2 - t1
            torch.cuda.synchronize()
        print("%-34s rep %d: forward %.1f us  backward %.1f us  total %.1f us" % (label, rep, tf / 400 * 1e6, tb / 400 * 1e6, (tf + tb) / 400 * 1e6), flush=True)


measure("python node", False)
measure("C++ operator", True)
measure("C++ operator, autograd 1 thread", True, False)
measure("python node, autograd 1 thread", False, False)
