"""Eager vs HIP-graph replay of a training step (render forward + backward w.r.t. azim / elev / dist) for a
launch-bound batch (C1-sized: 1 mesh x 4 views, 128^2) and for C3 (16 x 12, 224^2).  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn   # noqa: E402

dev = torch.device("cuda:0")


def run(B, V, size):
    vl, fl = zip(*[syn.synthetic_mesh("M", s) for s in range(B)])
    packed = PackedMeshes.from_meshes(Meshes([torch.from_numpy(v) for v in vl], [torch.from_numpy(f) for f in fl]), dev)
    r = MVRenderer(V, size, pc_rendering=False, object_color="white", background_color="white", light_direction="relative").eval()
    az, el = syn.unit_spherical_grid(V) if V in (6, 8, 12, 20) else (syn.circular_views(V)[0], syn.circular_views(V)[1])
    azim = torch.tensor(az, dtype=torch.float32, device=dev).reshape(1, V).repeat(B, 1).requires_grad_(True)
    elev = torch.tensor(el, dtype=torch.float32, device=dev).reshape(1, V).repeat(B, 1).requires_grad_(True)
    dist = torch.full((B, V), 2.2, device=dev).requires_grad_(True)
    G = torch.randn(B, V, 3, size, size, device=dev)

    def step():
        img, _ = r(packed, None, azim, elev, dist)
        return torch.autograd.grad((img * G).sum(), (azim, elev, dist))

    def timed(fn, n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    t_eager = timed(step)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    t_graph = timed(graph.replay)
    print("B=%2d V=%2d %3d^2  fwd+bwd: eager %7.1f us/step   HIP graph replay %7.1f us/step   (%.0f -> %.0f views/s)"
          % (B, V, size, t_eager, t_graph, B * V / t_eager * 1e6, B * V / t_graph * 1e6))


run(1, 4, 128)
run(2, 12, 224)
run(16, 12, 224)
