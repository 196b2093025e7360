"""bench.py's contract pieces that need no GPU: the algorithmic bytes of SURVEY 8(d), the workloads of the `secondary` block against
BASELINE.json's configs, and when the block runs (the driver's plain form of the command only)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_algorithmic_bytes_are_inputs_once_plus_outputs_once():
    b = _bench()
    B, V, H = 16, 12, 224
    nv, nf = [2562] * B, [5120] * B
    out = B * V * 3 * H * H * 4
    mesh_in = sum(v * 12 + f * 24 for v, f in zip(nv, nf))     # float32 vertices, int64 faces at the boundary
    assert b.algorithmic_bytes("mesh_fwd", B, V, H, H, nv, nf, 0) == mesh_in + B * V * 12 + out == 118065792   # the headline's roofline numerator
    assert b.algorithmic_bytes("mesh_fwdbwd", B, V, H, H, nv, nf, 0) == 2 * (mesh_in + B * V * 12 + out) + B * V * 12
    assert b.algorithmic_bytes("points_fwd", 20, 20, H, H, [], [], 2048) == 20 * 2048 * 12 + 400 * 12 + 400 * 3 * H * H * 4 == 241341120


def test_secondary_block_covers_the_baseline_configs_the_headline_does_not():
    b = _bench()
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert len(cfgs) == 5
    s = b.SECONDARY
    assert set(s) == {"cad", "ragged", "c3_fwdbwd", "c4_points", "c5_per_gpu"}
    assert s["cad"][:2] == ("mesh_fwd", "cad") and s["ragged"][:2] == ("mesh_fwd", "ragged")          # configs[1]: batch=16, nb_views=12
    assert s["cad"][3:] == s["ragged"][3:] == s["c3_fwdbwd"][3:] == (16, 12) and "batch=16, nb_views=12" in cfgs[1] and "batch=16, nb_views=12" in cfgs[2]
    assert s["c3_fwdbwd"][0] == "mesh_fwdbwd" and "fwd+bwd" in cfgs[2]
    assert s["c4_points"][0] == "points_fwd" and s["c4_points"][4] == 20 and "nb_views=20" in cfgs[3] and "2048 pts" in cfgs[3]
    assert s["c5_per_gpu"][2:] == ("L", 8, 20) and "batch=64, nb_views=20" in cfgs[4]                  # 64 objects over 8 GPUs


def test_secondary_block_runs_only_in_the_drivers_plain_command(monkeypatch):
    b = _bench()

    def plain(argv):
        monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
        a = b.parse()
        return a.workload is None and a.geometry is None and not a.no_cpu_baseline and not a.no_pipelined and not a.no_secondary
    assert plain(["--gpus", "1", "--steps", "20", "--warmup", "5"]) and plain([])
    for argv in (["--workload", "mesh_fwd"], ["--geometry", "ico"], ["--no-cpu-baseline"], ["--no-pipelined"], ["--no-secondary"]):
        assert not plain(argv), argv
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "plain_command = args.workload is None and args.geometry is None and not args.no_cpu_baseline and not args.no_pipelined" in src
