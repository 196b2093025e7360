"""Per-step GPU time of the first steps after a device synchronisation (C2 forward): where the ~0.6 ms of fixed cost in a
short timed region comes from.  Run on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn   # noqa: E402

dev = torch.device("cuda:0")
B, V = 16, 12
vl, fl = zip(*[syn.synthetic_mesh("M", s) for s in range(B)])
packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
r = MVRenderer(V, 224, pc_rendering=False, light_direction="fixed").eval()
az, el, di = syn.spherical_views(V)
azim, elev, dist = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
with torch.no_grad():
    for _ in range(300):
        r(packed, None, azim, elev, dist)
    for idle_ms in (0.0, 1.0, 20.0):
        torch.cuda.synchronize()
        time.sleep(idle_ms / 1e3)
        n = 40
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t0 = time.perf_counter()
        ev[0].record()
        host = []
        for k in range(n):
            r(packed, None, azim, elev, dist)
            ev[k + 1].record()
            host.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        d = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(n)]
        print("idle %4.1f ms before: first steps (us) %s ... mean of last 20: %.1f; wall %.0f us for %d steps (%.1f/step); host enqueue of step 1 done at %.0f us, all at %.0f us"
              % (idle_ms, " ".join("%.0f" % x for x in d[:8]), sum(d[20:]) / 20, wall * 1e6, n, wall * 1e6 / n, host[0] * 1e6, host[-1] * 1e6))
