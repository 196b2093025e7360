"""-m gpu: the N>1 code path with the REAL renderer over RCCL at world size 1 (the GPU box has one GPU; world size 2 runs
over gloo on the CPU in test_distributed.py): render_sharded(gather=True) (+ its reduce-scatter backward) and
render_sharded_overlapped must reproduce the plain MVRenderer call bit for bit."""
import socket

import numpy as np
import pytest
import torch

from mvtn_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rccl_world1():
    import torch.distributed as dist
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("pc", [False, True])
def test_render_sharded_real_renderer_world1(rccl_world1, pc):
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes
    from mvtn_amd.distributed import render_sharded, render_sharded_overlapped
    dev = torch.device("cuda", 0)
    B, V, S = 3, 4, 64
    vl, fl = zip(*[syn.synthetic_mesh("S", i) for i in range(B)])
    meshes = [Meshes([torch.from_numpy(v)], [torch.from_numpy(f)]) for v, f in zip(vl, fl)]
    points = torch.from_numpy(np.stack([syn.surface_points(v, f, 512, i) for i, (v, f) in enumerate(zip(vl, fl))])).to(dev)
    az, el, di = syn.spherical_views(V)
    rng = np.random.default_rng(0)
    azim = torch.from_numpy(np.tile(az, (B, 1)) + rng.uniform(-5, 5, (B, V)).astype(np.float32)).to(dev).requires_grad_(True)
    elev = torch.from_numpy(np.tile(el, (B, 1)) + rng.uniform(-5, 5, (B, V)).astype(np.float32)).to(dev).requires_grad_(True)
    dist_ = torch.from_numpy(np.tile(di, (B, 1))).to(dev)
    r = MVRenderer(V, S, pc_rendering=pc, points_radius=0.02, light_direction="relative").eval()
    m, p = (None, points) if pc else (meshes, None)
    want, _ = r(m, p, azim, elev, dist_)
    got, _ = render_sharded(r, m, p, azim, elev, dist_, gather=True)
    assert got.shape == (B, V, 3, S, S) and torch.equal(got, want)
    if not pc:   # the point path's camera gradients are identically zero at K = 1: nothing to compare
        G = torch.randn_like(want)
        ga, = torch.autograd.grad((got * G).sum(), azim, retain_graph=False)
        gw, = torch.autograd.grad((want * G).sum(), azim)
        assert torch.equal(ga, gw)
        packed = PackedMeshes.from_meshes(Meshes(list(map(torch.from_numpy, vl)), list(map(torch.from_numpy, fl))), dev)
        gp, _ = render_sharded(r, packed, None, azim.detach(), elev.detach(), dist_, gather=False)   # a packed LOCAL shard is accepted
        assert torch.equal(gp, want.detach())
    for chunks in (1, 2, 4):   # 4 pieces for 3 objects: one piece is empty
        ov, cams = render_sharded_overlapped(r, m, p, azim.detach(), elev.detach(), dist_, chunks=chunks)
        assert torch.equal(ov, want.detach()) and not ov.requires_grad
