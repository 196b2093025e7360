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


class _ShapedRenderer(torch.nn.Module):
    """Stand-in with MVRenderer's output layout (b, V, C, S, S) and its attributes (`image_size`, `shader`, `return_alpha`): a
    small S, but the real V = 20 and channel count, so that the gather's padding / slicing and the direct exchange's
    block offsets are exercised with the strides of the real (b, 20, 3, 224, 224) result.  Pixel values identify
    (object, view, channel, row, column)."""

    def __init__(self, size, channels=3):
        super().__init__()
        self.image_size = size
        self.shader, self.return_alpha = ("soft_phong", True) if channels == 4 else ("hard_phong", False)
        self.channels = channels

    def forward(self, meshes, points, azim, elev, dist_, color=None):
        B, V = azim.shape
        S, C = self.image_size, self.channels
        c = torch.arange(1.0, C + 1).view(1, 1, C, 1, 1)
        i = torch.arange(float(S)).view(1, 1, 1, S, 1)
        j = torch.arange(float(S)).view(1, 1, 1, 1, S)
        return azim.view(B, V, 1, 1, 1) * c + elev.view(B, V, 1, 1, 1) * i - dist_.view(B, V, 1, 1, 1) * j, None


def _worker_wide(rank, world, port, B, channels, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mvtn_amd.distributed import render_sharded, render_sharded_overlapped, shard_counts, shard_range
        torch.manual_seed(1)
        V, S = 20, 6
        azim = torch.randn(B, V, requires_grad=True)
        elev = torch.randn(B, V)
        dist_ = torch.rand(B, V) + 1
        r = _ShapedRenderer(S, channels)
        full = r(None, None, azim, elev, dist_)[0].detach()
        assert full.shape == (B, V, channels, S, S)
        got, _ = render_sharded(r, None, None, azim, elev, dist_, gather=True)   # ranks with an empty shard (B < world) included
        assert got.shape == full.shape and torch.equal(got.detach(), full), "gathered images differ from the unsharded render"
        G = torch.randn_like(full)
        (got * G).sum().backward()          # the reduce-scatter backward is a collective every rank enters
        lo, hi = shard_range(B, world, rank)
        want = torch.zeros(B, V)
        c = torch.arange(1.0, channels + 1).view(1, 1, channels, 1, 1)
        want[lo:hi] = (G * c).sum((2, 3, 4))[lo:hi] * world
        assert torch.allclose(azim.grad, want, atol=1e-4), "reduce-scatter backward mismatch"
        assert shard_counts(B, world)[rank] == hi - lo
        for chunks in (1, 2):
            ov, _ = render_sharded_overlapped(r, None, None, azim.detach(), elev, dist_, chunks=chunks)
            assert ov.shape == full.shape and torch.equal(ov, full), "overlapped exchange differs (chunks=%d)" % chunks
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B,channels", [(4, 64, 3), (8, 64, 3), (4, 5, 3), (8, 5, 3), (8, 3, 3), (4, 3, 4)])
def test_sharded_render_world4_and_8(world, B, channels):
    """BASELINE configs[4]'s partitioning (64 objects x 20 views over 8 ranks) and its ragged cousins (5 and 3 objects: uneven
    and EMPTY shards; RGBA frames from a soft shader with return_alpha) over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_wide, args=(r, world, port, B, channels, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(m == "ok" for _, m in res), res


def test_bench_self_launch_fails_loudly(tmp_path):
    """`bench.py --gpus N` outside a launcher starts its own ranks and must exit non-zero when they fail (here: no GPU in
    this container, so either fewer than N devices are visible -> exit 2 before anything starts, or a child dies)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    assert "bench.py" in r.stderr


def test_bench_self_launch_relays_child_failure(monkeypatch, capsys):
    """The relay itself: with enough devices "visible" and children that exit 3, self_launch exits 3 and prints no line."""
    sys.path.insert(0, ROOT)
    import argparse
    import subprocess
    import bench
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)

    class _R:
        returncode, stdout = 3, "rank 1 died\n"
    seen = {}

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return _R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(argparse.Namespace(gpus=4))
    assert e.value.code == 3
    assert "--nproc-per-node" in seen["cmd"] and "4" in seen["cmd"] and "127.0.0.1" in seen["cmd"]
    out = capsys.readouterr()
    assert '"metric"' not in out.out and "failed" in out.err

    class _Ok:
        returncode, stdout = 0, 'banner\n{"metric": "m", "value": 1}\n'
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: _Ok())
    with pytest.raises(SystemExit) as e:
        bench.self_launch(argparse.Namespace(gpus=4))
    assert e.value.code == 0
    assert capsys.readouterr().out.strip() == '{"metric": "m", "value": 1}'


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


def test_empty_shard_tail_channels_follow_the_renderer():
    """RGBA only from the soft MESH shaders with return_alpha; a point-cloud renderer constructed with
    shader="soft_phong", return_alpha=True still returns 3 channels, and so must the tail of an empty shard."""
    from mvtn_amd import MVRenderer
    from mvtn_amd.distributed import _channels
    assert _channels(MVRenderer(4, 32, pc_rendering=False, shader="soft_phong", return_alpha=True)) == 4
    assert _channels(MVRenderer(4, 32, pc_rendering=True, shader="soft_phong", return_alpha=True)) == 3
    assert _channels(MVRenderer(4, 32, pc_rendering=False, shader="hard_phong", return_alpha=True)) == 3
    assert _channels(MVRenderer(4, 32, pc_rendering=False, shader="soft_silhouette")) == 3

    class Stand:   # a stand-in without out_channels: the attribute rule
        shader, return_alpha, pc_rendering = "soft_phong", True, True
    assert _channels(Stand()) == 3
    Stand.pc_rendering = False
    assert _channels(Stand()) == 4
