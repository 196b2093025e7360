"""CPU (-m "not gpu"): the N>1 path — object sharding + all-gather of rendered views (+ its reduce-scatter
backward) — with world_size 2 over gloo.  The renderer is a deterministic stand-in on the CPU ranks (the HIP
library needs a GPU); what is under test is mvtn_amd.distributed."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _FakeRenderer(torch.nn.Module):
    """images[b, v, c, i, j] = azim[b, v] * (c + 1) + elev[b, v] * i - dist[b, v] * j + mean(points[b])"""

    def forward(self, meshes, points, azim, elev, dist_, color=None):
        B, V = azim.shape
        c = torch.arange(1, 4.0).view(1, 1, 3, 1, 1)
        i = torch.arange(4.0).view(1, 1, 1, 4, 1)
        j = torch.arange(5.0).view(1, 1, 1, 1, 5)
        img = azim.view(B, V, 1, 1, 1) * c + elev.view(B, V, 1, 1, 1) * i - dist_.view(B, V, 1, 1, 1) * j
        if points is not None:
            img = img + points.mean((1, 2)).view(B, 1, 1, 1, 1)
        return img, None


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvtn_amd.distributed import all_gather_views, render_sharded, shard_counts, shard_range
        torch.manual_seed(0)  # same full batch on every rank
        V = 3
        azim = torch.randn(B, V, requires_grad=True)
        elev = torch.randn(B, V, requires_grad=True)
        dist_ = torch.rand(B, V) + 1
        points = torch.randn(B, 7, 3)
        r = _FakeRenderer()
        full, _ = r(None, points, azim, elev, dist_)
        got, _ = render_sharded(r, None, points, azim, elev, dist_, gather=True)
        assert got.shape == full.shape and torch.allclose(got, full.detach()), "gathered images differ from the unsharded render"
        # backward: every rank backpropagates the same upstream gradient -> local grads = world * slice (sum over ranks)
        G = torch.randn_like(full)
        (got * G).sum().backward()
        ga = azim.grad.clone()
        azim.grad = None
        (full * G).sum().backward()
        lo, hi = shard_range(B, world, rank)
        want = torch.zeros_like(azim.grad)
        want[lo:hi] = azim.grad[lo:hi] * world
        assert torch.allclose(ga, want, atol=1e-5), "reduce-scatter backward mismatch"
        # no-gather mode returns the local shard only
        loc, _ = render_sharded(r, None, points, azim.detach(), elev.detach(), dist_, gather=False)
        assert loc.shape[0] == hi - lo and torch.allclose(loc, full.detach()[lo:hi])
        assert sum(shard_counts(B, world)) == B
        # equal-sized fast path (all_gather_into_tensor)
        x = torch.full((2, 3), float(rank))
        y = all_gather_views(x)
        assert y.shape == (2 * world, 3) and all(float(y[2 * k, 0]) == k for k in range(world))
        # overlapped, direct-placement exchange: same result as the unsharded render for any piece count,
        # including pieces (and whole ranks' pieces) that are empty
        from mvtn_amd.distributed import render_sharded_overlapped
        for chunks in (1, 2, 3):
            ov, cams = render_sharded_overlapped(r, None, points, azim.detach(), elev.detach(), dist_, chunks=chunks)
            assert ov.shape == full.shape and torch.equal(ov, full.detach()), "overlapped exchange differs (chunks=%d)" % chunks
            assert not ov.requires_grad
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5, 1])
def test_sharded_render_and_allgather_world2(B):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(m == "ok" for _, m in res), res


def test_shard_ranges():
    sys.path.insert(0, ROOT)
    from mvtn_amd.distributed import shard_counts, shard_range
    assert [shard_range(64, 8, r) for r in (0, 7)] == [(0, 8), (56, 64)]
    assert shard_counts(10, 4) == [3, 3, 2, 2]
    assert [shard_range(3, 4, r) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]


def test_packed_shard_must_match():
    """A PackedMeshes handed to render_sharded has to be exactly the local shard: anything else raises instead
    of rendering B meshes against a slice of the angles."""
    sys.path.insert(0, ROOT)
    from mvtn_amd.distributed import _local_meshes

    class _Packed:
        n_meshes = 3
    p = _Packed()
    assert _local_meshes(p, 2, 5, 8, 2) is p
    with pytest.raises(ValueError, match="exactly this rank's shard"):
        _local_meshes(p, 0, 4, 8, 2)
    assert _local_meshes(["a", "b", "c", "d"], 1, 3, 4, 2) == ["b", "c"]
    assert _local_meshes(None, 0, 1, 2, 2) is None
