"""Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The path shards by OBJECT: every (b, v) view is independent (models/renderer.py:65-151 has no
cross-view term) and a mesh's V views stay on the rank that holds the mesh, so mesh bytes never
cross the fabric and the backward needs no exchange (gradients are per-(b,v) scalars).  The only
collective is the all-gather of rendered views for consumers that want the whole batch on every
rank; its autograd backward is the matching reduce-scatter.  The reference itself is single-GPU
(run_mvtn.py:56-57,87).
"""
import torch
import torch.distributed as dist


def shard_range(n_objects, world_size, rank):
    """Contiguous [lo, hi) slice of the object axis for `rank`; sizes differ by at most one."""
    base, rem = divmod(n_objects, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_counts(n_objects, world_size):
    return [shard_range(n_objects, world_size, r)[1] - shard_range(n_objects, world_size, r)[0] for r in range(world_size)]


class _AllGatherViews(torch.autograd.Function):
    @staticmethod
    def forward(ctx, local, counts, group):
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
        ctx.counts, ctx.group, ctx.rank = counts, group, rank
        local = local.contiguous()
        mx = max(counts)
        if local.shape[0] != mx:  # uneven shards: pad to the largest so ONE fixed-size all-gather serves every backend
            pad = local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))
            local = torch.cat([local, pad], 0)
        out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local, group=group)
        if len(set(counts)) == 1:
            return out
        return torch.cat([out[r * mx:r * mx + c] for r, c in enumerate(counts)], 0)

    @staticmethod
    def backward(ctx, g):
        counts, group, rank = ctx.counts, ctx.group, ctx.rank
        g = g.contiguous()
        if len(set(counts)) == 1:
            out = g.new_empty((counts[0],) + tuple(g.shape[1:]))
            try:
                dist.reduce_scatter_tensor(out, g, op=dist.ReduceOp.SUM, group=group)
                return out, None, None
            except RuntimeError:  # backend without reduce-scatter (gloo): all-reduce + slice
                pass
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
        lo = sum(counts[:rank])
        return g[lo:lo + counts[rank]].contiguous(), None, None


def all_gather_views(local_images, counts=None, group=None):
    """(B_local, V, 3, H, W) on every rank -> (B, V, 3, H, W) on every rank, object order = rank order.
    Differentiable: backward is reduce-scatter(sum) of the incoming gradient."""
    world = dist.get_world_size(group)
    if counts is None:
        counts = [local_images.shape[0]] * world
    return _AllGatherViews.apply(local_images, list(counts), group)


def _channels(renderer):
    """Channels of the renderer's images: MVRenderer.out_channels (RGBA only from the soft MESH shaders with `return_alpha`;
    the point path and the hard shader return RGB whatever the flag says)."""
    oc = getattr(renderer, "out_channels", None)
    if oc is not None:
        return int(oc)
    soft_mesh = getattr(renderer, "shader", "hard_phong") != "hard_phong" and not getattr(renderer, "pc_rendering", False)
    return 4 if (getattr(renderer, "return_alpha", False) and soft_mesh) else 3


def _image_tail(renderer, azim, group, have_local, local_tail):
    """(V, 3, H, W) of the renderer's output for ranks that render nothing.  MVRenderer knows it (`image_size`); for any
    other callable the lowest rank that did render broadcasts its tail — every rank takes this branch together because
    the shard sizes are a function of (B, world) only."""
    S = getattr(renderer, "image_size", None)
    if S is not None:
        return (azim.shape[1], _channels(renderer), int(S), int(S))
    world = dist.get_world_size(group)
    t = torch.tensor(list(local_tail) if have_local else [0, 0, 0, 0], dtype=torch.int64, device=azim.device)
    allt = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allt, t, group=group)
    for x in allt:
        if int(x[0]) > 0:
            return tuple(int(v) for v in x)
    raise ValueError("render_sharded: no rank rendered anything (empty batch)")


def _local_meshes(meshes, lo, hi, B, world):
    """The local shard of `meshes`: a list of B objects is sliced; a PackedMeshes (already on this rank's GPU) must BE the
    local shard (hi - lo meshes) — packed batches are not re-sliced on the device."""
    if meshes is None:
        return None
    if hasattr(meshes, "n_meshes"):
        if meshes.n_meshes == hi - lo:
            return meshes
        raise ValueError("render_sharded: a PackedMeshes argument must hold exactly this rank's shard (%d meshes for objects "
                         "[%d, %d) of %d over %d ranks), got %d — pass the list of objects or pack the local slice"
                         % (hi - lo, lo, hi, B, world, meshes.n_meshes))
    return [meshes[i] for i in range(lo, hi)]


def render_sharded(renderer, meshes, points, azim, elev, dist_, color=None, gather=True, group=None):
    """Render this rank's shard of a batch that every rank holds in full (meshes: list of B objects, or a PackedMeshes
    of exactly the local shard; points / azim / elev / dist: leading dimension B), optionally all-gathering the images.
    A rank whose shard is empty (B < world size) renders nothing and contributes zero rows to the gather.
    Returns (images, cameras_of_the_local_shard)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = azim.shape[0]
    lo, hi = shard_range(B, world, rank)
    counts = shard_counts(B, world)
    c = color[lo:hi] if (torch.is_tensor(color) and color.dim() >= 2) else color
    images = cams = None
    if hi > lo:
        images, cams = renderer(_local_meshes(meshes, lo, hi, B, world), None if points is None else points[lo:hi],
                                azim[lo:hi], elev[lo:hi], dist_[lo:hi], c)
    if min(counts) == 0:   # the same on every rank
        tail = _image_tail(renderer, azim, group, images is not None, () if images is None else images.shape[1:])
        if images is None:
            images = torch.zeros((0,) + tuple(tail), dtype=torch.float32, device=azim.device)
            if torch.is_grad_enabled() and any(torch.is_tensor(x) and x.requires_grad for x in (azim, elev, dist_)):
                # keep this rank inside the autograd graph of the gather (its backward is a collective every rank must enter)
                images = images + 0.0 * sum(x[lo:hi].sum() for x in (azim, elev, dist_) if x.requires_grad)
    if gather:
        images = all_gather_views(images, counts, group)
    return images, cams


def _chunk_bounds(lo, hi, chunks):
    """Split [lo, hi) into `chunks` contiguous pieces (the first ones one longer); empty pieces are kept."""
    return [tuple(lo + shard_range(hi - lo, chunks, c)[k] for k in (0, 1)) for c in range(chunks)]


def render_sharded_overlapped(renderer, meshes, points, azim, elev, dist_, color=None, chunks=2, group=None, packed_pieces=None):
    """Inference-time variant of `render_sharded(gather=True)` that hides the exchange behind the rendering
    (at C5 the gather moves 771 MB per rank, about as long as the render itself): the local shard is
    rendered in `chunks` pieces and, as soon as piece c is queued, it is sent to every peer while piece c+1 renders.
    The exchange is direct: one send and one receive per peer and piece (batched: one RCCL group), each landing at its
    final place in the (B, V, 3, H, W) result — xGMI is point-to-point, so this is the fabric's native pattern, and
    there is no staging copy and no re-layout afterwards.  No autograd through the exchange (returns detached
    images); training keeps `render_sharded`.  `packed_pieces`: optionally this rank's pieces as HBM-resident
    `PackedMeshes` (one per piece, e.g. `PackedMeshes.from_meshes(meshes[lo:hi], device)` for the bounds
    `_chunk_bounds(*shard_range(B, world, rank), chunks)`), which skips the per-call packing of `meshes`.
    Returns (images of the WHOLE batch, cameras of the local pieces)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = azim.shape[0]
    ranges = [shard_range(B, world, r) for r in range(world)]
    pieces = [_chunk_bounds(lo, hi, chunks) for lo, hi in ranges]
    out = None
    works, keep, cams = [], [], []
    # shape of a rendered view block, for ranks / pieces without objects: known from the renderer, else (stand-in
    # renderers in tests) from the first rank's first object — no throw-away render on the GPU path
    tail = None
    if getattr(renderer, "image_size", None) is not None:
        tail = (azim.shape[1], _channels(renderer), int(renderer.image_size), int(renderer.image_size))
    elif min(hi - lo for lo, hi in ranges) == 0 or min(b - a for pr in pieces for a, b in pr) == 0:
        with torch.no_grad():
            probe = renderer(None if meshes is None else [meshes[0]], None if points is None else points[:1],
                             azim[:1], elev[:1], dist_[:1], color[:1] if (torch.is_tensor(color) and color.dim() >= 2) else color)[0]
        tail = tuple(probe.shape[1:])
    for c in range(chunks):
        lo, hi = pieces[rank][c]
        local = None
        if hi > lo:
            if packed_pieces is not None:
                m = packed_pieces[c]
            else:
                m = None if meshes is None else [meshes[i] for i in range(lo, hi)]
            p = None if points is None else points[lo:hi]
            col = color[lo:hi] if (torch.is_tensor(color) and color.dim() >= 2) else color
            with torch.no_grad():
                local, cam = renderer(m, p, azim[lo:hi], elev[lo:hi], dist_[lo:hi], col)
            local = local.contiguous()
            cams.append(cam)
            if out is None:
                out = local.new_empty((B,) + tuple(local.shape[1:]))
            out[lo:hi].copy_(local)
        if out is None:     # a rank without objects in this piece still has to post its receives
            out = torch.empty((B,) + tuple(tail), dtype=torch.float32, device=azim.device)
        ops = []
        for peer in range(world):
            if peer == rank:
                continue
            if local is not None:
                ops.append(dist.P2POp(dist.isend, local, peer, group))
            plo, phi = pieces[peer][c]
            if phi > plo:
                ops.append(dist.P2POp(dist.irecv, out[plo:phi], peer, group))
        if ops:
            works += dist.batch_isend_irecv(ops)     # asynchronous: the next piece renders while this one travels
            keep.append(local)
    for w in works:
        w.wait()
    return out, cams
