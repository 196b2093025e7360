"""Throughput of C2 forwards issued alternately on two HIP streams (independent batches in flight: the pre-pass of one
overlaps the rasteriser tail of the other) against the single-stream rate.  Run on the GPU box."""
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
r = MVRenderer(V, 224, pc_rendering=False, light_direction="fixed", rotation_check="off").eval()
az, el, di = syn.spherical_views(V)
azim, elev, dist = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
streams = [torch.cuda.Stream() for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]


def run(n, multi):
    outs = [None] * len(streams)
    with torch.no_grad():
        for k in range(n):
            if multi:
                s = streams[k % len(streams)]
                with torch.cuda.stream(s):
                    outs[k % len(streams)] = r(packed, None, azim, elev, dist)[0]
            else:
                outs[0] = r(packed, None, azim, elev, dist)[0]
    return outs


from mvtn_amd.pipeline import StreamPipeline   # noqa: E402
pipe = StreamPipeline(len(streams))
for mode in ("off", "device"):
    rr = MVRenderer(V, 224, pc_rendering=False, light_direction="fixed", rotation_check=mode).eval()
    for _ in range(200):
        pipe.submit(rr, packed, None, azim, elev, dist, inputs_ready=(mode == "off"))
    pipe.join()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        pipe.submit(rr, packed, None, azim, elev, dist, inputs_ready=(mode == "off"))
    pipe.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    print("StreamPipeline, rotation_check=%s, inputs_ready=%s: %.1f us per forward = %.2f M views/s" % (mode, mode == "off", dt * 1e6, B * V / dt / 1e6))

for multi in (False, True, False, True):
    run(200, multi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(2000, multi)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    print("%s: %.1f us per forward = %.2f M views/s" % ("%d streams" % len(streams) if multi else "1 stream ", dt * 1e6, B * V / dt / 1e6))
a = run(1, False)[0].clone()
b = run(2, True)
torch.cuda.synchronize()
print("same images:", all(torch.equal(a, x) for x in b if x is not None))
