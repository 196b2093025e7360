#!/usr/bin/env python3
"""bench.py — rendered views/s of the MVRenderer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload mesh_fwd|mesh_fwdbwd|points_fwd|c5|c5_per_gpu|soft_fwdbwd]

`--gpus N` with N > 1 and no RANK in the environment starts its own N ranks (torch.distributed.run on 127.0.0.1, one
per GPU) BEFORE anything touches a GPU, relays rank 0's JSON line and exits with the children's status; started by a
launcher (RANK set) it is one of the ranks.  On a box with fewer than N visible GPUs it fails loudly.

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
MVRenderer.forward (look-at + validity/remedy + vertex normals + projection + rasterise + shade).
Default workload at N = 1: B=16 "ModelNet-simplified-sized" meshes x 12 spherical views at 224x224 (BASELINE.json
configs[1], the configuration the metric is quoted on).  Default at N > 1: BASELINE.json configs[4] — 64 ShapeNet-sized
(tier L) meshes x 20 views IN TOTAL, sharded by object over the N ranks (64/N objects per GPU: fixed total work,
"scaling": "strong"), every step = render the local shard + the RCCL all-gather of the rendered views the north_star
names, so `value` INCLUDES the gather; the gather-free rate and the exchange hidden behind the rendering are reported
beside it (`allgather`).  Any other workload at N > 1 gives every rank its own B objects (weak scaling, no data-path
collective, the gather timed beside it).

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline      algorithmic HBM bytes per launch of the dominant kernel / its HIP-event duration
  cpu_baseline  the CPU oracle's literal pixel-major loop ("restated PyTorch3D-semantics CPU
                reference", 1 thread) on a bounded sample of the same workload
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); ~6300 achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough that the fixed cost around the timed region (~0.6 ms: first launches after the barrier,
    # final synchronisation) is amortised — 100 steps read 5 % low (120.4 vs 114.3 us per C2 step); still < 1 s of GPU time
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--prewarm", type=int, default=-1,
                    help="untimed steps run BEFORE the --warmup steps so that clocks, caches and the allocator are warm even when the "
                         "driver passes a short --warmup; default: what is missing to 300 steps in all (30 for the soft workload); "
                         "0 = none (then exactly --warmup untimed steps run).  Reported as prewarm_steps in the line.")
    ap.add_argument("--workload", default=None, choices=["mesh_fwd", "mesh_fwdbwd", "points_fwd", "c5", "c5_per_gpu", "soft_fwdbwd"],
                    help="default: mesh_fwd (C2) on one GPU, c5 on several.  c5 = BASELINE configs[4]: 64 ShapeNet-sized (tier L) "
                         "meshes x 20 views in total, sharded over the GPUs, the all-gather inside the step; c5_per_gpu = 8 such "
                         "meshes per GPU (one GPU's share at N = 8; weak scaling); soft_fwdbwd = C3's shape through the "
                         "SoftPhongShader path (K = 4, sigma = 1e-4)")
    ap.add_argument("--batch", type=int, default=0, help="objects per GPU (default: the BASELINE config's)")
    ap.add_argument("--views", type=int, default=0)
    ap.add_argument("--image-size", type=int, default=224)
    ap.add_argument("--tier", default="M")
    ap.add_argument("--geometry", default=None, choices=["ico", "cad", "ragged"],
                    help="mesh workloads: what the B objects are.  ico (default, the headline of every round): lumpy icospheres of --tier; "
                         "cad: synthetic.cad_mesh of the tier's face count — boxes / plates as two big triangles per side, cylinder slivers, "
                         "tessellated patches, ~30 %% of the parts doubled with reversed winding (coincident shells), parts inside parts: "
                         "the geometry class of the reference's ModelNet40; ragged: icospheres of tiers S, M, L, M, S, ... (1 : 2 : 1)")
    ap.add_argument("--strip-rows", type=int, default=0)
    ap.add_argument("--rotation-check", default="device", choices=["device", "sync", "off"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra two-stream throughput leg")
    ap.add_argument("--exchange-pieces", type=int, default=-1,
                    help="N>1: also time mvtn_amd.distributed.render_sharded_overlapped with this many pieces (0 = off; default: 2 for c5, else off)")
    ap.add_argument("--host-meshes", action="store_true", help="hand CPU meshes to every step (the reference's interface): packing + H2D inside the timed region; documentation only, never the headline")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) even at world size 1 (self-test of the N>1 code path)")
    ap.add_argument("--cpu-views", type=int, default=16, help="views in the cpu_baseline sample (~0.7 s each for meshes)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block (the other workloads measured after the headline in the driver's command)")
    ap.add_argument("--secondary-steps", type=int, default=500, help="timed steps of each `secondary` workload (behind 100 untimed ones; the fixed ~0.6 ms around a timed region is 6 us per step at 100 steps)")
    ap.add_argument("--ref-n1", action="store_true",
                    help="N > 1, workload c5: rank 0 also renders the WHOLE workload (all 64 objects) alone, after the timed regions, and "
                         "the line carries `scaling_vs_one_gpu` — speed-up and efficiency of the N-GPU run against that one-GPU reference, "
                         "render only and with the all-gather — so that one multi-GPU run yields the whole scaling statement")
    return ap.parse_args()


def algorithmic_bytes(workload, B, V, H, W, nv_list, nf_list, P):
    """Compulsory HBM traffic per batch: every input read once, every output written once."""
    N = B * V
    out = N * 3 * H * W * 4
    if workload.startswith("mesh"):
        mesh_in = sum(nv * 12 + nf * 24 for nv, nf in zip(nv_list, nf_list))  # faces int64 at the boundary
        fwd = mesh_in + N * 12 + out
        if workload == "mesh_fwd":
            return fwd
        return fwd + (out + mesh_in + N * 12 + N * 12)
    return B * P * 12 + N * 12 + out


def cpu_baseline(args, V):
    """Oracle (port), literal RasterizeMeshesNaiveCpu loop order, single thread, bounded sample."""
    from mvtn_amd import synthetic as syn
    from oracle import oracle as orc
    nviews = max(1, args.cpu_views)
    if args.workload == "points_fwd":
        v, f = syn.synthetic_mesh(args.tier, 0)
        pts = syn.surface_points(v, f, 2048, 0)[None]
        az, el, di = syn.spherical_views(20)
        az, el, di = az[None, :nviews], el[None, :nviews], np.full((1, nviews), 1.0, np.float32)
        t0 = time.perf_counter()
        orc.render_points(pts, az, el, di, args.image_size, radius=0.006, mode=0, return_fragments=False)
        dt = time.perf_counter() - t0
        sample = "%d views of cloud seed 0 (2048 pts, %d^2), literal pixel-major loop of RasterizePointsNaiveCpu" % (nviews, args.image_size)
    else:
        v, f = syn.geometry_mesh(args.geometry, args.tier, 1 if args.geometry == "ragged" else 0)   # (ragged: its first tier-M object)
        az, el, di = syn.spherical_views(V)
        az, el, di = az[None, :nviews], el[None, :nviews], di[None, :nviews]
        t0 = time.perf_counter()
        orc.render_meshes([v], [f], az, el, di, args.image_size, mode=0, return_fragments=False)
        dt = time.perf_counter() - t0
        sample = "%d views of mesh seed 0 (%s, tier %s, %d^2), literal pixel-major loop of RasterizeMeshesNaiveCpu + Phong (forward only)" % (
            nviews, args.geometry, args.tier, args.image_size)
    base = {"value": nviews / dt, "unit": "views/s", "cores": 1, "kind": "port", "sample": sample,
            "host_cores_available": os.cpu_count()}
    # the same loop with OpenMP over image rows on every host core; 4x the sample
    ncores = os.cpu_count() or 1
    if ncores > 1:
        orc.set_threads(ncores)
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            if args.workload == "points_fwd":
                orc.render_points(pts, az, el, di, args.image_size, radius=0.006, mode=0, return_fragments=False)
            else:
                orc.render_meshes([v], [f], az, el, di, args.image_size, mode=0, return_fragments=False)
        dt_all = time.perf_counter() - t0
        orc.set_threads(1)
        # NOT an "all cores" baseline: the loop is parallel over the 224 image rows of ONE view only (no parallelism over views,
        # a serial set-up per view), so it stops scaling at a few dozen threads; kept as what it is, under a name that says so
        base["openmp_over_rows"] = {"value": reps * nviews / dt_all, "unit": "views/s", "threads": ncores,
                                    "note": "the same loop with OpenMP over the image rows of one view at a time (SURVEY 8d ii); upstream's CPU "
                                            "rasteriser is single-threaded — `value` above is the baseline"}
    return base


SECONDARY = {   # name -> (workload, geometry, tier, B, V): every BASELINE config the headline (C2 on icospheres) does not time
    "cad": ("mesh_fwd", "cad", "M", 16, 12),            # configs[1] on the geometry class the reference renders (ModelNet40: CAD models)
    "ragged": ("mesh_fwd", "ragged", "M", 16, 12),      # configs[1], ragged tier-S / M / L / M batch
    "c3_fwdbwd": ("mesh_fwdbwd", "ico", "M", 16, 12),   # configs[2]
    "c4_points": ("points_fwd", "ico", "M", 20, 20),    # configs[3]
    "c5_per_gpu": ("mesh_fwd", "ico", "L", 8, 20),      # configs[4]: one GPU's share at N = 8
}


def measure_secondary(name, dev, lib, H, steps, untimed, with_oracle, rotation_check="device"):
    """One workload of the `secondary` block: the same step as the headline's (MVRenderer.forward [+ backward]) on other inputs,
    `untimed` steps, then `steps` timed ones between two synchronisations, then the dominant kernel's HIP-event time over `steps`
    more, the fast rasteriser's statistics and a bounded parity check against the oracle (the checker, never the thing timed)."""
    from mvtn_amd import MVRenderer, Meshes, PackedMeshes, synthetic as syn
    workload, geometry, tier, B, V = SECONDARY[name]
    W = H
    P = 2048
    az, el, di = syn.spherical_views(V)
    if workload == "points_fwd":
        di = np.full(V, 1.0, np.float32)
    azim, elev, dist_t = (torch.from_numpy(np.tile(x, (B, 1))).to(dev) for x in (az, el, di))
    nv_list, nf_list, vl, fl = [], [], [], []
    points = packed = None
    if workload == "points_fwd":
        clouds = []
        for i in range(B):
            v, f = syn.synthetic_mesh(tier, i)
            clouds.append(syn.surface_points(v, f, P, seed=i))
        points = torch.from_numpy(np.stack(clouds)).to(dev)
        renderer = MVRenderer(V, H, pc_rendering=True, object_color="white", background_color="white", points_radius=0.006,
                              rotation_check=rotation_check).eval()
    else:
        for i in range(B):
            v, f = syn.geometry_mesh(geometry, tier, i)
            vl.append(torch.from_numpy(v)); fl.append(torch.from_numpy(f))
            nv_list.append(len(v)); nf_list.append(len(f))
        packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)
        renderer = MVRenderer(V, H, pc_rendering=False, object_color="white", background_color="white",
                              light_direction="relative" if workload == "mesh_fwdbwd" else "fixed", rotation_check=rotation_check).eval()
    if workload == "mesh_fwdbwd":
        rng = np.random.default_rng(1)
        azim = (azim + torch.from_numpy(rng.uniform(-5, 5, (B, V)).astype(np.float32)).to(dev)).requires_grad_(True)
        elev = (elev + torch.from_numpy(rng.uniform(-5, 5, (B, V)).astype(np.float32)).to(dev)).requires_grad_(True)
        dist_t = (dist_t + torch.from_numpy(rng.uniform(-0.1, 0.1, (B, V)).astype(np.float32)).to(dev)).requires_grad_(True)
        upstream = torch.randn(B, V, 3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(2))

    def render():
        if workload == "mesh_fwdbwd":
            azim.grad = elev.grad = dist_t.grad = None
            img, _ = renderer(packed, None, azim, elev, dist_t)
            img.backward(upstream)
            return img
        with torch.no_grad():
            return renderer(packed, points, azim, elev, dist_t)[0]

    # untimed: at least `untimed` steps AND at least 250 ms of them — the GPU idles while the host builds a workload's inputs and its
    # clocks come back up only after some milliseconds of load: 100 steps of the 55-us point step (5 ms) read 8 us per step slow
    import gc
    gc.collect()      # (the previous workloads' garbage — meshes, fixtures, oracle buffers — is collected here, not inside the timed region)
    torch.cuda.empty_cache()   # ... and their device blocks go back to the driver: this workload's buffers are allocated as in a process of its own
    n_untimed, t_w = 0, time.perf_counter()
    # (the results of the steps are dropped at once, as in the headline's timed() region: every step then writes the SAME output block.
    # It matters for C4 only: its 240 MB of views fit the 256 MB Infinity Cache once, not twice — with two blocks alternating
    # (`out = render()` in the loop) the point step reads 62 us instead of 55; C2's 116 MB fit twice)
    while n_untimed < untimed or time.perf_counter() - t_w < 0.25:
        for _ in range(50):
            render()
        torch.cuda.synchronize()
        n_untimed += 50
    gc_was_on = gc.isenabled()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(steps):
        render()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    out = render()
    # the host's share of a step (Python + dispatch + launches), measured with a short queue: 4 x 25 steps, each group behind a synchronize
    host_s = 0.0
    for _ in range(4):
        torch.cuda.synchronize()
        render()
        t1 = time.perf_counter()
        for _ in range(25):
            render()
        host_s += time.perf_counter() - t1
    torch.cuda.synchronize()
    # kernel durations: HIP events on the launch stream (library-side), the same number of steps; the mesh pre-pass in a pass of its own
    kind = 1 if workload == "points_fwd" else 0
    ms, cnt = (ctypes.c_double * 8)(), (ctypes.c_int32 * 8)()
    lib.mvr_profile_begin()
    for _ in range(steps):
        render()
    lib.mvr_profile_end(ms, cnt)
    torch.cuda.synchronize()
    kern_ms = ms[kind] / max(1, cnt[kind])
    alg = algorithmic_bytes("mesh_fwd" if workload == "mesh_fwdbwd" else workload, B, V, H, W, nv_list, nf_list, P)
    res = {"workload": "%s: B=%d x %d views, %dx%d%s" % (workload, B, V, H, W, "" if workload == "points_fwd" else ", geometry %s, tier %s" % (geometry, tier)),
           "value": B * V * steps / elapsed, "unit": "views/s", "ms_per_step": elapsed / steps * 1e3, "steps": steps, "untimed_steps": n_untimed, "host_enqueue_ms_per_step": host_s / 100 * 1e3,
           "kernel": "k_points_raster_shade" if workload == "points_fwd" else "k_mesh_raster_fast", "kernel_ms": kern_ms,
           "algorithmic_bytes_per_launch": alg, "frac": (alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if kern_ms > 0 else None}
    if workload == "mesh_fwdbwd":
        if cnt[2] > 0:
            res["backward_kernel_ms"] = ms[2] / cnt[2]
        alg_step = algorithmic_bytes("mesh_fwdbwd", B, V, H, W, nv_list, nf_list, P)
        res["step_algorithmic_bytes"] = alg_step
        res["step_frac"] = alg_step / (elapsed / steps) / 1e9 / HBM_PEAK_GBS
    if workload.startswith("mesh"):
        from mvtn_amd import functional as F
        with torch.no_grad():
            Rm, Tm = F.look_at_view_transform(dist_t.detach().reshape(-1), elev.detach().reshape(-1), azim.detach().reshape(-1))
            sp = F.mesh_params(V, H, 1, (0.99999,) * 3, (0.99999,) * 3)
            st4 = [int(x) for x in F.render_meshes_fragments(packed, Rm, Tm, torch.tensor([[0.0, 1.0, 0.0]], device=dev), sp, return_stats=True)[2].cpu()]
        n_strips_all = B * V * ((H + 7) // 8)
        res["fast_raster_stats"] = {"strips": n_strips_all, "strips_redone_by_exact_body": st4[0], "settle_candidates": st4[1],
                                    "entries_rasterised_by_a_whole_wave": st4[2], "strip_list_entries": st4[3], "prepass_kernel_ms": None}
        os.environ["MVR_PROF_MASK"] = "0x20"
        lib.mvr_profile_begin()
        for _ in range(steps):
            render()
        ms5, cnt5 = (ctypes.c_double * 8)(), (ctypes.c_int32 * 8)()
        lib.mvr_profile_end(ms5, cnt5)
        del os.environ["MVR_PROF_MASK"]
        torch.cuda.synchronize()
        if cnt5[5]:
            res["fast_raster_stats"]["prepass_kernel_ms"] = ms5[5] / cnt5[5]
    res["max_pixel_err_vs_oracle"] = None
    if with_oracle:
        # bounded: object 0 (ragged: its first tier-M object, index 1), the first 4 views
        from oracle import oracle as orc
        ob = 1 if geometry == "ragged" else 0
        nvw = min(4, V)
        img = out.detach()[ob:ob + 1, :nvw].cpu().numpy()
        a_, e_, d_ = (x.detach()[ob:ob + 1, :nvw].cpu().numpy() for x in (azim, elev, dist_t))
        if workload == "points_fwd":
            want = orc.render_points(points[ob:ob + 1].cpu().numpy(), a_, e_, d_, H, radius=0.006, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3,
                                     return_fragments=False)
        else:
            vn, fn = vl[ob].numpy(), fl[ob].numpy()
            ld = orc.camera_position(a_, e_, d_) if workload == "mesh_fwdbwd" else ((0.0, 1.0, 0.0),)
            # vertices whose face normals cancel (a shell and its reversed copy on shared vertices) are shaded from the noise of a float
            # sum's order in the reference; only for such meshes the oracle sums as the library does (order-independent): switch #4
            exact_sum = bool(orc.cancelled_normal_vertices(vn, fn).any())
            orc.set_normals_exact_sum(exact_sum)
            want = orc.render_meshes([vn], [fn], a_, e_, d_, H, light_dir=ld, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3, return_fragments=False)
            orc.set_normals_exact_sum(False)
            res["oracle_normals_exact_sum"] = exact_sum
        res["max_pixel_err_vs_oracle"] = float(np.abs(img - want["images"]).max())
        res["parity_sample"] = "object %d, first %d views" % (ob, nvw)
    del renderer, packed, points, out
    torch.cuda.empty_cache()
    return res


def self_launch(args):
    """`--gpus N` (N > 1) outside a launcher: start N ranks, relay rank 0's line.  Runs before any GPU call (a process that
    has initialised the GPU must not be replaced or forked into ranks)."""
    import socket
    import subprocess
    n_vis = torch.cuda.device_count()   # does not initialise the GPU on this image
    if n_vis < args.gpus:
        sys.stderr.write("bench.py: --gpus %d requested but only %d GPU(s) are visible on this node\n" % (args.gpus, n_vis))
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if r.returncode != 0 or not lines:
        sys.stderr.write("bench.py: the %d-rank run failed (exit code %d)\n%s\n" % (args.gpus, r.returncode, r.stdout[-2000:]))
        sys.exit(r.returncode or 1)
    print(lines[-1], flush=True)
    sys.exit(0)


def timed(step, steps, barrier, dist_on, dev):
    """K steps bracketed by barrier + synchronize on both sides; MAX over ranks.  Returns seconds."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist_on:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def main():
    args = parse()
    # the `secondary` block belongs to the DRIVER's form of the command (no workload / geometry chosen, nothing switched off): the
    # developer tools, which pass --workload / --geometry / --no-cpu-baseline / --no-pipelined, get the one line they ask for
    plain_command = args.workload is None and args.geometry is None and not args.no_cpu_baseline and not args.no_pipelined
    args.geometry = args.geometry or "ico"
    if args.workload is None:
        args.workload = "c5" if args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1 else "mesh_fwd"
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    # Only the JSON line may reach stdout: libraries (RCCL prints a version banner on fd 1) are diverted to stderr
    # until the result is ready.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from mvtn_amd import MVRenderer, Meshes, PackedMeshes, _lib, synthetic as syn
    from mvtn_amd.distributed import shard_range
    lib = _lib.load()

    wl_name = args.workload
    c5 = wl_name in ("c5", "c5_per_gpu")
    soft = wl_name == "soft_fwdbwd"
    fixed_total = wl_name == "c5" and not args.batch      # BASELINE configs[4]: 64 objects IN TOTAL, sharded over the ranks
    seed0 = 0                                              # first object's seed on this rank
    if c5:
        args.workload, args.tier = "mesh_fwd", "L"
        args.views = args.views or 20
        if fixed_total:
            lo, hi = shard_range(64, world, rank)
            args.batch, seed0 = hi - lo, lo
        else:
            args.batch = args.batch or 8
    if soft:
        args.workload = "mesh_fwdbwd"
    H = W = args.image_size
    if args.workload == "points_fwd":
        B, V, P = args.batch or 20, args.views or 20, 2048
    else:
        B, V, P = args.batch or 16, args.views or 12, 0
    if not fixed_total:
        seed0 = rank * B
    n_views_total = 64 * V if fixed_total else B * V * world
    az, el, di = syn.spherical_views(V)
    if args.workload == "points_fwd":
        di = np.full(V, 1.0, np.float32)
    azim = torch.from_numpy(np.tile(az, (B, 1))).to(dev)
    elev = torch.from_numpy(np.tile(el, (B, 1))).to(dev)
    dist_t = torch.from_numpy(np.tile(di, (B, 1))).to(dev)
    nv_list, nf_list = [], []
    packed = points = None
    if args.workload == "points_fwd":
        clouds = []
        for i in range(B):
            v, f = syn.synthetic_mesh(args.tier, seed0 + i)
            clouds.append(syn.surface_points(v, f, P, seed=seed0 + i))
        points = torch.from_numpy(np.stack(clouds)).to(dev)
        renderer = MVRenderer(V, H, pc_rendering=True, object_color="white", background_color="white",
                              points_radius=0.006, rotation_check=args.rotation_check, strip_rows=args.strip_rows).eval()
    else:
        vl, fl = [], []
        for i in range(B):
            v, f = syn.geometry_mesh(args.geometry, args.tier, seed0 + i)
            vl.append(torch.from_numpy(v))
            fl.append(torch.from_numpy(f))
            nv_list.append(len(v))
            nf_list.append(len(f))
        packed = PackedMeshes.from_meshes(Meshes(vl, fl), dev)  # inputs resident in HBM before the timed region
        if args.host_meshes:
            packed = [Meshes([v], [f]) for v, f in zip(vl, fl)]  # list of single-mesh objects on the CPU, as collate_fn yields
        light = "relative" if args.workload == "mesh_fwdbwd" else "fixed"
        extra = {}
        if soft:   # SoftPhongShader as PyTorch3D's tutorials configure it: sigma 1e-4, blur_radius = log(1/1e-4 - 1) sigma, K = 4
            extra = dict(shader="soft_phong", faces_per_pixel=4, sigma=1e-4, gamma=1e-4, blur_radius=float(np.log(1.0 / 1e-4 - 1.0) * 1e-4))
        renderer = MVRenderer(V, H, pc_rendering=False, object_color="white", background_color="white",
                              light_direction=light, rotation_check=args.rotation_check, strip_rows=args.strip_rows, **extra).eval()
    if args.workload == "mesh_fwdbwd":
        rng = np.random.default_rng(1)
        azim = (azim + torch.from_numpy(rng.uniform(-5, 5, (B, V)).astype(np.float32)).to(dev)).requires_grad_(True)
        elev = (elev + torch.from_numpy(rng.uniform(-5, 5, (B, V)).astype(np.float32)).to(dev)).requires_grad_(True)
        dist_t = (dist_t + torch.from_numpy(rng.uniform(-0.1, 0.1, (B, V)).astype(np.float32)).to(dev)).requires_grad_(True)
        gen = torch.Generator(device=dev).manual_seed(2)
        upstream = torch.randn(B, V, 3, H, W, device=dev, generator=gen)

    def render():
        if args.workload == "mesh_fwdbwd":
            azim.grad = elev.grad = dist_t.grad = None
            img, _ = renderer(packed, None, azim, elev, dist_t)
            img.backward(upstream)
            return img
        if args.workload == "points_fwd":
            with torch.no_grad():
                return renderer(None, points, azim, elev, dist_t)[0]
        with torch.no_grad():
            return renderer(packed, None, azim, elev, dist_t)[0]

    # c5 on several GPUs: the step is the local render + the all-gather of the rendered views (north_star)
    gather_in_step = fixed_total and dist_on and not args.no_gather      # (dist_on at world 1 = --force-dist: the same code path, self-test)
    if gather_in_step:
        from mvtn_amd.distributed import all_gather_views, shard_counts
        counts = shard_counts(64, world)

        def step():
            return all_gather_views(render(), counts)
    else:
        step = render

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # untimed: max(W, 300) steps in all — a short --warmup (the driver passes 5) leaves clocks, caches and the allocator cold
    # (tools/fixed_cost.py: a 20-step region reads 104.5 us / step after 50 steps, 101.6 after 300, 97.6 in steady state).
    # The W warm-up steps proper follow a synchronize of their own: the first launch after a long unsynchronised run pays
    # the HIP runtime's clean-up of the completed commands (200-300 us on this box), which belongs to no step.
    prewarm = max(0, (30 if soft else 300) - args.warmup) if args.prewarm < 0 else args.prewarm
    for _ in range(prewarm):
        out = step()
    barrier()
    for _ in range(args.warmup):
        out = step()
    elapsed = timed(step, args.steps, barrier, dist_on, dev)
    # the host's share of a step: Python + dispatch + launches of 25 steps at a time with an empty queue in front and a
    # synchronize behind (inside the K-step region above the host runs into the queue's depth and waits for the GPU there);
    # the first step of a group is left out (the first launch after a synchronize pays the runtime's clean-up, see above)
    host_s = 0.0
    for _ in range(4):
        torch.cuda.synchronize()
        step()
        t0 = time.perf_counter()
        for _ in range(25):
            step()
        host_s += time.perf_counter() - t0
    torch.cuda.synchronize()
    host_enqueue_ms = host_s / 100 * 1e3
    out = render()
    value = n_views_total * args.steps / elapsed

    # ---- dominant-kernel duration, HIP events on the launch stream, same K steps
    kind = 3 if soft else {"mesh_fwd": 0, "points_fwd": 1, "mesh_fwdbwd": 0}[args.workload]
    lib.mvr_profile_begin()
    for _ in range(args.steps):
        render()
    ms = (ctypes.c_double * 8)()
    cnt = (ctypes.c_int32 * 8)()
    lib.mvr_profile_end(ms, cnt)
    torch.cuda.synchronize()
    kern_ms = ms[kind] / max(1, cnt[kind])
    alg = algorithmic_bytes("mesh_fwd" if args.workload == "mesh_fwdbwd" else args.workload, B, V, H, W, nv_list, nf_list, P)
    if soft:
        alg += B * V * H * W * 4     # the alpha plane of the RGBA frames
    achieved = alg / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    # HBM traffic of the dominant kernel: PMC passes of the same command, committed under profiles/ (FETCH_SIZE / WRITE_SIZE
    # cannot be read inside this process); `from_profiles` names the file so that a stale number is visible as such
    traffic, traffic_src = None, None
    for tpath in ([] if (c5 or soft or args.geometry != "ico") else sorted(__import__("glob").glob(os.path.join(ROOT, "profiles", "r*_traffic_%s.json" % args.workload)))[-1:]):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            traffic_src = os.path.relpath(tpath, ROOT)
        except Exception:
            traffic = None
    kname = "k_mesh_soft_forward" if soft else ["k_mesh_raster_fast", "k_points_raster_shade", "k_mesh_raster_fast"][kind]
    roofline = {"bound": "hbm", "kernel": kname,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_from_profiles": traffic_src, "algorithmic_bytes_per_launch": alg, "kernel_ms": kern_ms,
                "launches_timed": int(cnt[kind])}
    if args.workload == "mesh_fwdbwd" and cnt[4 if soft else 2] > 0:
        roofline["backward_kernel_ms"] = ms[4 if soft else 2] / cnt[4 if soft else 2]
    # VALU issue floor of the mesh kernels.  Measured in round 4 (tools/valu_issue.hip -> profiles/r04_a_valu_issue.csv): with
    # >= 2 waves per SIMD a plain f32 / integer wave64 instruction (v_add / v_mul / v_fma / v_mov / v_and with VGPR operands)
    # occupies the SIMD for 2 cycles, as /opt/skills/guides/MI355X_MICROARCH.md says — not 4, what rounds 1-3 assumed —, the
    # packed f32 forms, min / max / med3, DPP moves, integer mad / shift-add, converts and anything with an SGPR operand 3.2,
    # compares 4, rcp / rsq 6.3.  The floor below prices EVERY instruction at 2 cycles: a lower bound of the issue time; the
    # kernel's real mix is dearer (cycles_per_instruction_floor names the price used).  Instruction count: the committed SQ
    # counter pass (a property of the code and the workload, not of the run).
    try:
        import glob
        sqp = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq_counters.json")))[-1]
        k = json.load(open(sqp)).get(kname)
        if k and k.get("SQ_INSTS_VALU") and kern_ms > 0 and args.workload != "points_fwd" and not c5 and not soft and args.geometry == "ico":
            floor_ms = k["SQ_INSTS_VALU"] * 2.0 / 1024.0 / 2.07e9 * 1e3
            roofline["valu"] = {"wave_instructions": k["SQ_INSTS_VALU"], "cycles_per_instruction_floor": 2.0, "issue_floor_ms": floor_ms, "frac": floor_ms / kern_ms,
                                "from_profiles": os.path.relpath(sqp, ROOT)}
    except Exception:
        pass

    # ---- beside `value`: the render alone when the step carries the gather
    beside = {}
    if gather_in_step:
        t_r = timed(render, args.steps, barrier, dist_on, dev)
        beside["render_only"] = {"views_per_s": n_views_total * args.steps / t_r, "ms_per_step": t_r / args.steps * 1e3,
                                 "note": "the local shards rendered, no all-gather (what data-parallel training consumes)"}

    # ---- beside `value`: the same K forwards issued round-robin on two streams (mvtn_amd.pipeline.StreamPipeline):
    # independent batches in flight, the pre-pass of one fills the rasteriser tail of the other.  An inference-time
    # usage pattern (training steps depend on each other), reported separately and never part of `value`.
    pipelined = None
    if args.workload == "mesh_fwd" and not args.host_meshes and not args.no_pipelined:   # (the point kernels have no tail to fill)
        from mvtn_amd.pipeline import StreamPipeline
        pipe = StreamPipeline(2, dev)
        for _ in range(min(args.warmup, 50)):
            pipe.submit(render, inputs_ready=True)
        pipe.join()

        def piped_all():
            for _ in range(args.steps):
                pipe.submit(render, inputs_ready=True)
            pipe.join()
        tp = timed(piped_all, 1, barrier, dist_on, dev)
        pipelined = {"streams": 2, "views_per_s": B * V * world * args.steps / tp, "ms_per_step": tp / args.steps * 1e3}

    # ---- C3, beside `value` (which stays the step issued eagerly from Python, as in every round): the same training step
    # captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed K times.  The eager step costs the host ~0.2 ms of
    # Python / autograd dispatch, about what the GPU needs for it, so on slow host cores `value` measures the host; a
    # replay costs the host one call.
    graphed = None
    if args.workload == "mesh_fwdbwd" and not args.host_meshes and not args.no_pipelined:
        try:
            ga, ge, gd = (x.detach().clone().requires_grad_(True) for x in (azim, elev, dist_t))   # fresh leaves: the eager
            # ones carry gradient-accumulation nodes bound to the eager stream

            def gstep():
                img, _ = renderer(packed, None, ga, ge, gd)
                return img, torch.autograd.grad(img, (ga, ge, gd), upstream)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    gstep()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                g_keep = gstep()
            for _ in range(min(args.warmup, 50)):
                graph.replay()
            tg_ = timed(graph.replay, args.steps, barrier, dist_on, dev)
            graphed = {"views_per_s": n_views_total * args.steps / tg_, "ms_per_step": tg_ / args.steps * 1e3,
                       "note": "the captured step replayed (torch.cuda.CUDAGraph): same kernels, one host call per step"}
            del graph, g_keep
        except Exception as exc:   # (a capture failure must not lose the measured line)
            graphed = {"error": str(exc)[:200]}

    # ---- C3, beside `value`: the same eager step with the autograd engine's worker threads off
    # (torch.autograd.set_multithreading_enabled(False): the backward then runs on the calling thread).  With one device the
    # worker thread only adds two thread hand-overs per step; how long they take depends on what the host's cores were doing
    # (tools/host_autograd_mt.py: 178 us of host time per step cold, 103 warm, 85 with the switch), so the default line swings
    # between host-bound and GPU-bound from run to run and this one does not.
    single_thread = None
    if args.workload == "mesh_fwdbwd" and not args.host_meshes and not args.no_pipelined and hasattr(torch.autograd, "set_multithreading_enabled"):
        with torch.autograd.set_multithreading_enabled(False):
            for _ in range(min(args.warmup, 50)):
                render()
            ts_ = timed(render, args.steps, barrier, dist_on, dev)
        single_thread = {"views_per_s": n_views_total * args.steps / ts_, "ms_per_step": ts_ / args.steps * 1e3,
                         "note": "eager, torch.autograd.set_multithreading_enabled(False)"}

    # ---- the all-gather of rendered views the north_star names: inside `value` for c5 on several GPUs (above); for the
    # weak-scaling workloads timed here, beside `value`
    gather = None
    if dist_on and not args.no_gather:
        from mvtn_amd.distributed import all_gather_views
        gather = {"bytes_gathered_per_rank": int(out.numel() * 4 * world) if not fixed_total else int(64 * V * 3 * H * W * 4)}
        reps = max(3, min(20, args.steps))
        if gather_in_step:
            gather["inside_value"] = True
        else:
            img = out.detach()
            for _ in range(3):
                all_gather_views(img)

            def with_gather():
                all_gather_views(render().detach())
            tg = timed(with_gather, reps, barrier, dist_on, dev)
            gather["views_per_s_with_allgather"] = n_views_total * reps / tg
        chunks = args.exchange_pieces if args.exchange_pieces >= 0 else (2 if fixed_total else 0)
        if chunks > 0 and args.workload != "mesh_fwdbwd" and not args.host_meshes:
            # the same exchange hidden behind the rendering: the shard is rendered in pieces, piece c travels (direct
            # sends to every peer, landing in place) while piece c+1 renders (mvtn_amd.distributed.render_sharded_overlapped)
            from mvtn_amd.distributed import _chunk_bounds, render_sharded_overlapped
            Btot = 64 if fixed_total else B * world
            lo, hi = shard_range(Btot, world, rank)
            if fixed_total:
                azg, elg, dg = (torch.from_numpy(np.tile(x, (Btot, 1))).to(dev) for x in (az, el, di))
            else:
                azg, elg, dg = (x.detach().repeat(world, 1) for x in (azim, elev, dist_t))
            pieces = ptsg = None
            if args.workload == "points_fwd":
                ptsg = torch.zeros(Btot, P, 3, device=dev)
                ptsg[lo:hi] = points
            else:
                pieces = [PackedMeshes.from_meshes(Meshes(vl[a - lo:b - lo], fl[a - lo:b - lo]), dev) if b > a else None
                          for a, b in _chunk_bounds(lo, hi, chunks)]

            def overlapped():
                return render_sharded_overlapped(renderer, None, ptsg, azg, elg, dg, chunks=chunks, packed_pieces=pieces)[0]
            for _ in range(3):
                overlapped()
            to = timed(overlapped, reps, barrier, dist_on, dev)
            gather["views_per_s_with_overlapped_exchange"] = n_views_total * reps / to
            gather["exchange_pieces"] = chunks

    # ---- what the fast rasteriser met on this geometry (mesh workloads, hard shader): one extra forward through the same library
    # call with its diagnostics read back (mvr_mesh_fast_stats) + the pre-pass kernel's duration (profile kind 5)
    fast_stats = None
    if args.workload.startswith("mesh") and not soft and rank == 0 and not args.host_meshes:
        from mvtn_amd import functional as F
        with torch.no_grad():
            Rm, Tm = F.look_at_view_transform(dist_t.detach().reshape(-1), elev.detach().reshape(-1), azim.detach().reshape(-1))
            sp = F.mesh_params(V, H, 1, (0.99999,) * 3, (0.99999,) * 3, strip_rows=args.strip_rows)
            _, _, st4 = F.render_meshes_fragments(packed, Rm, Tm, torch.tensor([[0.0, 1.0, 0.0]], device=dev), sp, return_stats=True)
        st4 = [int(x) for x in st4.cpu()]
        TH = args.strip_rows or 8
        n_strips_all = B * V * ((H + TH - 1) // TH)
        fast_stats = {"strips": n_strips_all, "strips_redone_by_exact_body": st4[0], "redo_fraction": st4[0] / n_strips_all,
                      "settle_candidates": st4[1],                    # (pixel, face) candidates the exact settle was handed (up to 4 per pixel pair)
                      "strip_list_entries": st4[3],
                      "entries_rasterised_by_a_whole_wave": st4[2],   # the "fat" wide faces only; thin wide entries run one lane per row and are not counted
                      "whole_wave_share": (st4[2] / st4[3]) if st4[3] else None,
                      "prepass_kernel_ms": None}
        os.environ["MVR_PROF_MASK"] = "0x20"   # the pre-pass alone, in a pass of its own
        lib.mvr_profile_begin()
        for _ in range(min(args.steps, 200)):
            render()
        ms5, cnt5 = (ctypes.c_double * 8)(), (ctypes.c_int32 * 8)()
        lib.mvr_profile_end(ms5, cnt5)
        del os.environ["MVR_PROF_MASK"]
        torch.cuda.synchronize()
        if cnt5[5]:
            fast_stats["prepass_kernel_ms"] = ms5[5] / cnt5[5]

    # ---- --ref-n1: the one-GPU reference of the strong-scaling statement, measured in THIS run by rank 0 (the other ranks wait)
    scaling_ref = None
    if args.ref_n1 and dist_on and fixed_total:
        if rank == 0:
            vla, fla = [], []
            for i in range(64):
                v, f = syn.geometry_mesh(args.geometry, args.tier, i)
                vla.append(torch.from_numpy(v)); fla.append(torch.from_numpy(f))
            packed_all = PackedMeshes.from_meshes(Meshes(vla, fla), dev)
            az_a, el_a, di_a = (torch.from_numpy(np.tile(x, (64, 1))).to(dev) for x in (az, el, di))

            def render_all():
                with torch.no_grad():
                    return renderer(packed_all, None, az_a, el_a, di_a)[0]
            for _ in range(20):
                render_all()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps1 = max(20, min(200, args.steps))
            for _ in range(reps1):
                render_all()
            torch.cuda.synchronize()
            t1 = (time.perf_counter() - t0) / reps1
            v1 = 64 * V / t1
            r_only = beside.get("render_only", {}).get("views_per_s", value)
            scaling_ref = {"one_gpu_views_per_s": v1, "one_gpu_ms_per_step": t1 * 1e3, "n_gpus": world,
                           "speedup_render_only": r_only / v1, "efficiency_render_only": r_only / v1 / world,
                           "speedup_with_allgather": (value / v1) if gather_in_step else None,
                           "efficiency_with_allgather": (value / v1 / world) if gather_in_step else None,
                           "note": "strong scaling of the fixed 64-object workload: the same run's N-GPU rates over rank 0 rendering all 64 objects alone"}
            del packed_all
        dist.barrier()

    cpu = None
    parity = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, V)
        # second half of BASELINE.json's metric: max pixel |delta| vs the PyTorch3D-semantics oracle
        # (first 2 objects of rank 0's shard, all views; the oracle is the checker, never the thing timed)
        from oracle import oracle as orc
        nb = min(1 if soft else 2, B)
        img = out.detach()[:nb].cpu().numpy()
        a_, e_, d_ = (x.detach()[:nb].cpu().numpy() for x in (azim, elev, dist_t))
        if args.workload == "points_fwd":
            want = orc.render_points(points[:nb].cpu().numpy(), a_, e_, d_, H, radius=0.006, obj_color=(0.99999,) * 3,
                                     bg_color=(0.99999,) * 3, return_fragments=False)
            parity = float(np.abs(img - want["images"]).max())
        elif soft:
            parity = soft_parity(orc, vl[0].numpy(), fl[0].numpy(), a_, e_, d_, H, renderer, img)
        else:
            vln = [vl[i].numpy() for i in range(nb)]
            fln = [fl[i].numpy() for i in range(nb)]
            ld = orc.camera_position(a_, e_, d_) if args.workload == "mesh_fwdbwd" else ((0.0, 1.0, 0.0),)
            # the oracle's switch #4 — vertex-normal sums accumulated order-independently, as the library does — ONLY where a mesh of the
            # sample has vertices whose face normals cancel (coincident reversed shells on shared vertices: a float sum there is noise
            # of its summation order; oracle/mvr_oracle_impl.inc orc_vertex_normals; tests/test_gpu_mesh_forward.py shows that only those
            # pixels differ).  Everywhere else the oracle's default — the reference's float index_add in face order — is the checker.
            exact_sum = any(bool(orc.cancelled_normal_vertices(v_, f_).any()) for v_, f_ in zip(vln, fln))
            orc.set_normals_exact_sum(exact_sum)
            want = orc.render_meshes(vln, fln, a_, e_, d_, H, light_dir=ld, obj_color=(0.99999,) * 3, bg_color=(0.99999,) * 3,
                                     return_fragments=False)
            orc.set_normals_exact_sum(False)
            parity = float(np.abs(img - want["images"]).max())

    # ---- `secondary`: the driver's command also measures the workloads the headline does not — the CAD-like and
    # ragged batches of configs[1], C3 fwd+bwd, C4 points, the per-GPU share of C5 — each from 100 untimed + --secondary-steps timed
    # steps in this process, after the headline's numbers are final (nothing here changes `value`, `config` or `roofline`)
    secondary = None
    if (plain_command and rank == 0 and world == 1 and not dist_on and wl_name == "mesh_fwd" and args.geometry == "ico" and not args.host_meshes
            and not args.no_secondary and not args.batch and not args.views and args.image_size == 224 and args.tier == "M" and not args.strip_rows):
        secondary = {}
        del out
        t_sec = time.perf_counter()
        only = [x for x in os.environ.get("MVR_BENCH_SECONDARY", "").split(",") if x]   # (developer aid: a subset, in the given order)
        for nm in (only or SECONDARY):
            try:
                secondary[nm] = measure_secondary(nm, dev, lib, H, max(1, args.secondary_steps), 100, not args.no_cpu_baseline, args.rotation_check)
            except Exception as exc:   # (a failure here must not lose the headline)
                secondary[nm] = {"error": "%s: %s" % (type(exc).__name__, str(exc)[:300])}
        secondary["seconds"] = time.perf_counter() - t_sec

    if rank == 0:
        if fixed_total:
            wl = "C5: 64 tier-L meshes x %d spherical views in total, %dx%d, mesh render fwd%s; %d..%d objects per GPU" % (
                V, H, W, " + RCCL all-gather of the rendered views" if gather_in_step else "", 64 // world, -(-64 // world))
        elif c5:
            wl = "C5 (one GPU's share): B=%d tier-L meshes x %d spherical views, %dx%d, mesh render fwd" % (B, V, H, W)
        elif soft:
            wl = "C3 shape through SoftPhongShader: B=%d tier-%s meshes x %d jittered views, %dx%d, faces_per_pixel 4, sigma 1e-4, fwd+bwd" % (B, args.tier, V, H, W)
        else:
            geo = {"ico": "tier-%s" % args.tier, "cad": "CAD-like (cad_mesh, %d faces)" % (nf_list[0] if nf_list else 0), "ragged": "ragged tier-S/M/L/M"}[args.geometry]
            wl = {"mesh_fwd": "C2: B=%d %s meshes x %d spherical views, %dx%d, mesh render fwd" % (B, geo, V, H, W),
                  "mesh_fwdbwd": "C3: B=%d %s meshes x %d jittered views, %dx%d, fwd+bwd (grad wrt azim/elev/dist, relative light)" % (B, geo, V, H, W),
                  "points_fwd": "C4: B=%d clouds x 2048 pts x %d views, %dx%d, point render fwd" % (B, V, H, W)}[args.workload]
        line = {
            "metric": "rendered views/sec (224x224 point clouds, 20 views/obj)" if args.workload == "points_fwd" else
                      "rendered views/sec (224x224, %d views/obj)" % V,
            "value": value, "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "prewarm_steps": prewarm,   # untimed steps in front of the W warm-up steps (--prewarm): untimed in all = prewarm + W
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if fixed_total else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "host_enqueue_ms_per_step": host_enqueue_ms,   # Python + dispatch + kernel launches per step, measured with a short queue (4 x 25 steps after the timed region)
            "config": {"workload": wl,
                       "objects_per_gpu": B, "views_per_object": V, "image_size": H, "rotation_check": args.rotation_check,
                       "inputs": "CPU meshes packed + copied every step" if args.host_meshes else "resident in HBM",
                       "issue": "eager: one MVRenderer.forward%s per step from Python" % (" + backward" if args.workload == "mesh_fwdbwd" else ""),
                       "autograd_node": None if args.workload == "points_fwd" or soft else
                                        ("C++ (torch.ops.mvr.render_mesh_views, csrc/mvr_torch_op.cpp)" if _lib.torch_op() is not None else "Python (functional._RenderMeshViews)"),
                       "untimed_steps_before_the_timed_region": prewarm + args.warmup,
                       "parallelism": ("objects sharded over %d GPU(s); " % world) +
                                      ("one RCCL all-gather of the rendered views per step" if gather_in_step else "no data-path collective")},
            "roofline": roofline,
            "geometry": args.geometry if args.workload.startswith("mesh") else None,
            "fast_raster_stats": fast_stats,
            "cpu_baseline": cpu,
            "max_pixel_err_vs_oracle": parity,
        }
        line.update(beside)
        if scaling_ref:
            line["scaling_vs_one_gpu"] = scaling_ref
        if gather:
            line["allgather"] = gather
        if dist_on:
            line["world_size_seen_by_rccl"] = dist.get_world_size()
        if pipelined:
            line["two_streams"] = pipelined
        if graphed:
            line["hip_graph_replay"] = graphed
        if single_thread is not None:
            line["eager_autograd_single_thread"] = single_thread
        if secondary is not None:
            line["secondary"] = secondary
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)   # C stdio of the libraries (the RCCL banner is buffered there): out to stderr now
        except OSError:
            pass
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def soft_parity(orc, v, f, az, el, di, H, renderer, img):
    """max |RGB difference| of object 0's views against the oracle's SoftPhongShader restatement (bounded: first 2 views)."""
    nv = min(2, az.shape[1])
    ld = orc.camera_position(az[:, :nv], el[:, :nv], di[:, :nv])
    want = orc.render_meshes_soft([v], [f], az[:, :nv], el[:, :nv], di[:, :nv], H, light_dir=ld, obj_color=(0.99999,) * 3,
                                  bg_color=(0.99999,) * 3, blur_radius=renderer.blur_radius, faces_per_pixel=renderer.faces_per_pixel,
                                  sigma=renderer.sigma, gamma=renderer.gamma)
    return float(np.abs(img[:, :nv] - want["images"][:, :nv, 0:3]).max())


if __name__ == "__main__":
    main()
