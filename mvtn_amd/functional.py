"""torch.autograd bindings of the C ABI (include/mvr.h).  PyTorch supplies device memory, the
current HIP stream and autograd plumbing; all arithmetic happens in libmvr_hip.so."""
import torch

from . import _lib
from .structures import PackedMeshes


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream      # ~1 us; torch.cuda.current_stream() builds a Python object (~13 us)
except AttributeError:                                    # pragma: no cover
    _raw_stream = None


def _stream():
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t.contiguous()


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mvtn_amd: tensors must live on the GPU (the HIP library is the only path; there is no CPU fallback)")


def _ptr(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------
# cameras
# ---------------------------------------------------------------------------------------------

class _LookAt(torch.autograd.Function):
    @staticmethod
    def forward(ctx, azim, elev, dist, max_trials, seed, host_status=None):
        lib = _lib.load()
        azim, elev, dist = _f32c(azim), _f32c(elev), _f32c(dist)
        _require_cuda(azim, elev, dist)
        n = azim.numel()
        dev = azim.device
        R = torch.empty(n, 3, 3, dtype=torch.float32, device=dev)
        T = torch.empty(n, 3, dtype=torch.float32, device=dev)
        status = torch.empty(n + 2, dtype=torch.int32, device=dev)  # [0] re-draws, [1] #invalid, [2:] per-view valid
        if max_trials >= 0:
            used = torch.empty(2, n, dtype=torch.float32, device=dev)
            # host_status: 2 pinned int32 words the kernel writes directly (zero-copy) — no D2H copy on the stream
            sp = status.data_ptr() if host_status is None else host_status.data_ptr()
            _lib.check(lib.mvr_look_at_checked(azim.data_ptr(), elev.data_ptr(), dist.data_ptr(), n, int(max_trials),
                                               int(seed) & 0xFFFFFFFFFFFFFFFF, R.data_ptr(), T.data_ptr(), used[0].data_ptr(),
                                               used[1].data_ptr(), status[2:].data_ptr(), sp, _stream()),
                       "mvr_look_at_checked")
            ctx.save_for_backward(used[0], used[1], dist)
        else:
            _lib.check(lib.mvr_look_at(azim.data_ptr(), elev.data_ptr(), dist.data_ptr(), n, R.data_ptr(), T.data_ptr(),
                                       status[2:].data_ptr(), status.data_ptr(), _stream()), "mvr_look_at")
            ctx.save_for_backward(azim, elev, dist)
        ctx.mark_non_differentiable(status)
        return R, T, status

    @staticmethod
    def backward(ctx, gR, gT, _gs):
        lib = _lib.load()
        azim, elev, dist = ctx.saved_tensors
        n = azim.numel()
        gR = None if gR is None else _f32c(gR)
        gT = None if gT is None else _f32c(gT)
        ga, ge, gd = torch.empty_like(azim), torch.empty_like(elev), torch.empty_like(dist)
        _lib.check(lib.mvr_look_at_backward(azim.data_ptr(), elev.data_ptr(), dist.data_ptr(), n, _ptr(gR), _ptr(gT),
                                            ga.data_ptr(), ge.data_ptr(), gd.data_ptr(), _stream()), "mvr_look_at_backward")
        return ga, ge, gd, None, None, None


def look_at_view_transform(dist, elev, azim, return_status=False, max_trials=-1, seed=0, host_status=None):
    """R (N,3,3), T (N,3) for flat per-view (N,) distance / elevation / azimuth in degrees —
    pytorch3d look_at_view_transform as called at models/renderer.py:79-80 — differentiable.
    status (int32, N+2): [0] = remedy re-draws performed, [1] = number of views failing
    check_valid_rotation_matrix (util.py:403-420), [2:] = per-view validity; all computed on the
    device without a host sync.  max_trials >= 0 enables the device-side remedy loop of
    ops.py:156-165 (see mvr_look_at_checked in include/mvr.h); with `host_status` (2 pinned int32 words)
    the kernel writes status[0:2] straight into host memory instead of the returned device tensor."""
    R, T, status = _LookAt.apply(azim.reshape(-1), elev.reshape(-1), dist.reshape(-1), max_trials, seed, host_status)
    return (R, T, status) if return_status else (R, T)


class _CameraPosition(torch.autograd.Function):
    @staticmethod
    def forward(ctx, azim, elev, dist):
        lib = _lib.load()
        azim, elev, dist = _f32c(azim), _f32c(elev), _f32c(dist)
        _require_cuda(azim, elev, dist)
        n = azim.numel()
        C = torch.empty(n, 3, dtype=torch.float32, device=azim.device)
        _lib.check(lib.mvr_camera_position(azim.data_ptr(), elev.data_ptr(), dist.data_ptr(), n, C.data_ptr(), _stream()),
                   "mvr_camera_position")
        ctx.save_for_backward(azim, elev, dist)
        return C

    @staticmethod
    def backward(ctx, gC):
        lib = _lib.load()
        azim, elev, dist = ctx.saved_tensors
        gC = _f32c(gC)
        ga, ge, gd = torch.empty_like(azim), torch.empty_like(elev), torch.empty_like(dist)
        _lib.check(lib.mvr_camera_position_backward(azim.data_ptr(), elev.data_ptr(), dist.data_ptr(), azim.numel(),
                                                    gC.data_ptr(), ga.data_ptr(), ge.data_ptr(), gd.data_ptr(), _stream()),
                   "mvr_camera_position_backward")
        return ga, ge, gd


def camera_position_from_spherical_angles(distance, elevation, azimuth):
    """(N,3) camera positions (degrees) — models/renderer.py:168-169 — differentiable."""
    return _CameraPosition.apply(azimuth.reshape(-1), elevation.reshape(-1), distance.reshape(-1))


# ---------------------------------------------------------------------------------------------
# mesh path
# ---------------------------------------------------------------------------------------------

def mesh_params(n_views, image_size, n_light, obj_color, bg_color, cull_backfaces=False, perspective_correct=True,
                fov=60.0, ambient=0.5, diffuse=0.3, specular=0.2, shininess=64.0, strip_rows=0, exact_raster=False, z_clip=None,
                gather_backward=False, normalize=None, fov_scale=0.0, normals_three_cross=True, projection_fma=False):
    """`exact_raster`: skip the fast rasteriser (MVR_RASTER_EXACT_ONLY); the images are the same either way.
    `gather_backward`: skip the face-table backward kernel (MVR_BACKWARD_GATHER); same gradients up to summation order.
    `normalize`: (mean, std) per channel (or scalars): the trainer's Normalize, (rgb - mean) / std, applied in the
    rasteriser's stores (no pass over the frames afterwards); the backward differentiates the normalised images.
    `z_clip`: view depth of PyTorch3D >= 0.5's face clipping; None = the reference's rule (znear / 2 = 0.5 when
    perspective_correct — MeshRasterizer only clips for perspective cameras with it on —, else off); 0 = off.
    `fov_scale`: 0 = the projection scale from `fov` (1 / tan(fov / 2) evaluated in double, rounded once); > 0: used as given
    (K[0][0] of a particular PyTorch3D build).  `normals_three_cross` (default): the vertex normals from three cross products per
    face (MVR_NORMALS_THREE_CROSS: Meshes._compute_vertex_normals of the PyTorch3D releases of the reference's era); False: one
    cross product per face added to its three vertices (the later form).  ~1e-7 apart, colours only.  `projection_fma`:
    MVR_PROJECTION_FMA — world -> view accumulated like a GPU matrix product (PyTorch3D's transform_points is a torch.bmm) instead of
    the closed form; decides only which of two coincident faces wins a pixel."""
    if z_clip is None:
        z_clip = 0.5 if perspective_correct else 0.0
    H, W = (image_size, image_size) if isinstance(image_size, int) else image_size
    flags = (_lib.MVR_CULL_BACKFACES if cull_backfaces else 0) | (_lib.MVR_PERSPECTIVE_CORRECT if perspective_correct else 0)
    flags |= _lib.MVR_RASTER_EXACT_ONLY if exact_raster else 0
    flags |= _lib.MVR_BACKWARD_GATHER if gather_backward else 0
    flags |= _lib.MVR_NORMALS_THREE_CROSS if normals_three_cross else 0
    flags |= _lib.MVR_PROJECTION_FMA if projection_fma else 0
    p = _lib.MeshParams(n_views, H, W, flags, n_light, strip_rows, fov, ambient, diffuse, specular, shininess)
    for k in range(3):
        p.obj_color[k] = float(obj_color[k])
        p.bg_color[k] = float(bg_color[k])
    p.z_clip = float(z_clip)
    p.fov_scale = float(fov_scale)
    _set_normalize(p, normalize)
    return p


def with_window(params, scale=None, flip=None, crop_i=None, crop_j=None, pad=0):
    """A copy of `params` (mesh_params or points_params) that makes the rasteriser render straight onto the reference's view-regulariser window
    (ops.py:138-176): arguments as `mvtn_amd.regularize.draw_regularization` returns them — per image n = b*V + v a dropout
    scale (f32) or None, a flip flag (u8) or None, crop offsets (i32) or None, and the padding.  The frames (and, through
    the autograd nodes, their gradients) are those of render -> regualarize_rendered_views, with no pass over the frames."""
    if scale is None and flip is None and crop_i is None:
        return params
    ref = next(t for t in (scale, flip, crop_i) if t is not None)
    n, dev = ref.numel(), ref.device

    def as_is(t, dtype):   # no kernel when the draws already have the library's types (draw_regularization's do)
        return t if t.dtype == dtype and t.is_contiguous() else t.to(dtype).contiguous()
    if crop_i is not None and crop_j is not None:
        # the crop offsets go in as drawn: the library subtracts `win_pad` itself (no elementwise launch per step)
        dx, dy, win_pad = as_is(crop_j, torch.int32), as_is(crop_i, torch.int32), int(pad)
    else:
        zeros = torch.zeros(n, dtype=torch.int32, device=dev)
        dx = zeros if crop_j is None else (crop_j.to(torch.int32) - int(pad)).contiguous()
        dy = zeros if crop_i is None else (crop_i.to(torch.int32) - int(pad)).contiguous()
        win_pad = 0
    fl = torch.zeros(n, dtype=torch.uint8, device=dev) if flip is None else as_is(flip, torch.uint8)
    sc = None if scale is None else as_is(scale, torch.float32)
    _require_cuda(dx, dy, fl, sc)
    p = type(params).from_buffer_copy(params)   # (MeshParams or PointsParams: the same fields)
    p.win_dx, p.win_dy, p.win_flip, p.win_scale = dx.data_ptr(), dy.data_ptr(), fl.data_ptr(), (None if sc is None else sc.data_ptr())
    p.win_pad = win_pad
    p._keep = (dx, dy, fl, sc)   # the struct only holds their addresses
    return p


def _copy_params(params, extra_flags):
    """`params` with extra flag bits; the variants are remembered on the params object (a training step asks for the same two
    every time: copying the ctypes struct costs ~4 us of host time per call)."""
    cache = params.__dict__.setdefault("_variants", {})
    p = cache.get(extra_flags)
    src = bytes(params)   # (a caller may have changed a field of `params` since the variant was made)
    if p is None or p._src_bytes != src:
        p = _lib.MeshParams.from_buffer_copy(params)
        p._src_bytes = src
        p.flags |= extra_flags
        p._keep = getattr(params, "_keep", None)
        cache[extra_flags] = p
    return p


class _RenderMeshes(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, T, light, packed, params, need_grad):
        lib = _lib.load()
        R, T, light = _f32c(R), _f32c(T), _f32c(light)
        _require_cuda(R, T, light, packed.verts)
        B, V, H, W = packed.n_meshes, params.n_views, params.image_h, params.image_w
        N = B * V
        if R.shape[0] != N or T.shape[0] != N:
            raise ValueError("R/T must have B*V = %d rows" % N)
        dev = R.device
        mb = packed.c_struct()
        if need_grad:   # the rasteriser leaves its lists of covered pixels / winning faces for the backward
            params = _copy_params(params, _lib.MVR_KEEP_PIXEL_LISTS)
        ws_bytes = lib.mvr_mesh_workspace_bytes(mb, params)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        images = torch.empty(B, V, 3, H, W, dtype=torch.float32, device=dev)
        p2f = torch.empty(N, H, W, dtype=torch.int32, device=dev) if need_grad else None
        _lib.check(lib.mvr_render_meshes_forward(mb, params, R.data_ptr(), T.data_ptr(), light.data_ptr(), images.data_ptr(),
                                                 _ptr(p2f), ws.data_ptr(), ws_bytes, _stream()), "mvr_render_meshes_forward")
        if need_grad:
            ctx.save_for_backward(R, T, light, p2f, ws)  # ws: the backward reuses the vertex attributes in it
            ctx.packed, ctx.params = packed, params
        return images

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        R, T, light, p2f, ws = ctx.saved_tensors
        packed, params = ctx.packed, ctx.params
        g = _f32c(g)
        N = R.shape[0]
        mb = packed.c_struct()
        ws_bytes = ws.numel()
        bparams = _copy_params(params, _lib.MVR_WORKSPACE_FROM_FORWARD)
        gR, gT = torch.empty_like(R), torch.empty_like(T)
        per_view_light = params.n_light == N and N > 0 and ctx.needs_input_grad[2]
        gL = torch.empty(N, 3, dtype=torch.float32, device=R.device) if per_view_light else None
        _lib.check(lib.mvr_render_meshes_backward(mb, bparams, R.data_ptr(), T.data_ptr(), light.data_ptr(), p2f.data_ptr(),
                                                  g.data_ptr(), gR.data_ptr(), gT.data_ptr(), _ptr(gL), ws.data_ptr(), ws_bytes,
                                                  _stream()), "mvr_render_meshes_backward")
        return gR, gT, gL, None, None, None


def render_meshes(packed: PackedMeshes, R, T, light_dir, params):
    """images (B,V,3,H,W) — MeshRenderer(MeshRasterizer, HardPhongShader) of models/renderer.py:89-107
    plus the re-layout of :109-113; differentiable w.r.t. R, T and a per-view light_dir."""
    need_grad = torch.is_grad_enabled() and (R.requires_grad or T.requires_grad or light_dir.requires_grad)
    return _RenderMeshes.apply(R, T, light_dir, packed, params, need_grad)


class _RenderMeshViews(torch.autograd.Function):
    """models/renderer.py:79-113 as ONE autograd node and ONE library call per direction
    (mvr_render_mesh_views_forward / _backward): look_at (+ the device-side remedy of ops.py:156-165) ->
    [camera-position light, :168-169] -> rasterise + shade.  One node instead of three cuts the host cost of a training
    step (graph construction, three backward dispatches, the gradient-accumulation kernels between them) below the GPU time."""

    @staticmethod
    def forward(ctx, azim, elev, dist, packed, params, light, max_trials, seed, host_status, need_grad):
        lib = _lib.load()
        azim, elev, dist = _f32c(azim), _f32c(elev), _f32c(dist)
        _require_cuda(azim, elev, dist, packed.verts)
        B, V, H, W = packed.n_meshes, params.n_views, params.image_h, params.image_w
        N = B * V
        if azim.numel() != N:
            raise ValueError("azim/elev/dist must have B*V = %d elements" % N)
        dev = azim.device
        ctx.set_materialize_grads(False)   # unused outputs (R, T of the returned cameras) arrive as None, not zeros
        R = torch.empty(N, 3, 3, dtype=torch.float32, device=dev)
        T = torch.empty(N, 3, dtype=torch.float32, device=dev)
        status = torch.empty(N + 2, dtype=torch.int32, device=dev)
        aux = torch.empty(5, N, dtype=torch.float32, device=dev)   # rows 0-1: angles used | rows 2-4 (as (N,3)): relative light
        aux_p = aux.data_ptr()
        relative = light is None
        if not relative:
            light = _f32c(light)
        sp = status.data_ptr() if host_status is None else host_status.data_ptr()
        mb = packed.c_struct()
        if need_grad:   # the rasteriser leaves its lists of covered pixels / winning faces for the backward
            params = _copy_params(params, _lib.MVR_KEEP_PIXEL_LISTS)
        ws_bytes = packed.workspace_bytes(params)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        images = torch.empty(B, V, 3, H, W, dtype=torch.float32, device=dev)
        p2f = torch.empty(N, H, W, dtype=torch.int32, device=dev) if need_grad else None
        _lib.check(lib.mvr_render_mesh_views_forward(mb, params, azim.data_ptr(), elev.data_ptr(), dist.data_ptr(), int(max_trials),
                                                     int(seed) & 0xFFFFFFFFFFFFFFFF, _ptr(None if relative else light),
                                                     R.data_ptr(), T.data_ptr(), aux_p, aux_p + 4 * N, aux_p + 8 * N,
                                                     status.data_ptr() + 8, sp, images.data_ptr(), _ptr(p2f), ws.data_ptr(),
                                                     ws_bytes, _stream()), "mvr_render_mesh_views_forward")
        if need_grad:
            ctx.save_for_backward(azim, elev, dist, aux, R, T, aux if relative else light, p2f, ws)
            ctx.packed, ctx.params, ctx.relative = packed, params, relative
        ctx.mark_non_differentiable(status)
        return images, R, T, status

    @staticmethod
    def backward(ctx, g, gR_ext, gT_ext, _gs):
        lib = _lib.load()
        azim, elev, dist, aux, R, T, light, p2f, ws = ctx.saved_tensors
        packed, params = ctx.packed, ctx.params
        N = R.shape[0]
        st = _stream()
        dev = R.device
        aux_p = aux.data_ptr()
        light_p = aux_p + 8 * N if ctx.relative else light.data_ptr()
        bparams = _copy_params(params, _lib.MVR_WORKSPACE_FROM_FORWARD)
        grads = torch.empty(3, N, dtype=torch.float32, device=dev)
        gp = grads.data_ptr()
        if g is not None and gR_ext is None and gT_ext is None:   # the training step: everything in one call
            _lib.check(lib.mvr_render_mesh_views_backward(packed.c_struct(), bparams, azim.data_ptr(), elev.data_ptr(), dist.data_ptr(),
                                                          aux_p, aux_p + 4 * N, R.data_ptr(), T.data_ptr(), light_p,
                                                          1 if ctx.relative else 0, p2f.data_ptr(), _f32c(g).data_ptr(), gp, gp + 4 * N,
                                                          gp + 8 * N, ws.data_ptr(), ws.numel(), st), "mvr_render_mesh_views_backward")
            ga, ge, gd = grads.unbind(0)   # (one call for the three views)
            return ga, ge, gd, None, None, None, None, None, None, None
        # somebody also differentiated through the returned cameras (or only through them): chain the three backwards
        gR_c, gT_c = torch.zeros(N, 3, 3, dtype=torch.float32, device=dev), torch.zeros(N, 3, dtype=torch.float32, device=dev)
        gL = torch.zeros(N, 3, dtype=torch.float32, device=dev) if ctx.relative else None
        if g is not None:
            _lib.check(lib.mvr_render_meshes_backward(packed.c_struct(), bparams, R.data_ptr(), T.data_ptr(), light_p, p2f.data_ptr(),
                                                      _f32c(g).data_ptr(), gR_c.data_ptr(), gT_c.data_ptr(), _ptr(gL), ws.data_ptr(),
                                                      ws.numel(), st), "mvr_render_meshes_backward")
        if gR_ext is not None:
            gR_c = gR_c + gR_ext
        if gT_ext is not None:
            gT_c = gT_c + gT_ext
        _lib.check(lib.mvr_look_at_backward(aux_p, aux_p + 4 * N, dist.data_ptr(), N, gR_c.data_ptr(), gT_c.data_ptr(), gp, gp + 4 * N,
                                            gp + 8 * N, st), "mvr_look_at_backward")
        if gL is not None:
            g2 = torch.empty(3, N, dtype=torch.float32, device=dev)
            _lib.check(lib.mvr_camera_position_backward(azim.data_ptr(), elev.data_ptr(), dist.data_ptr(), N, gL.data_ptr(),
                                                        g2[0].data_ptr(), g2[1].data_ptr(), g2[2].data_ptr(), st),
                       "mvr_camera_position_backward")
            grads += g2
        return grads[0], grads[1], grads[2], None, None, None, None, None, None, None


def render_mesh_views(packed: PackedMeshes, azim, elev, dist, light_dir, params, max_trials=-1, seed=0, host_status=None):
    """(images (B,V,3,H,W), R (N,3,3), T (N,3), status) for flat (N,) azim / elev / dist in degrees: the whole mesh
    path of models/renderer.py:65-114 as one differentiable call.  light_dir: (1,3) / (N,3) tensor, or None for the
    reference's "relative" light (camera position of every view, differentiable).  status as in look_at_view_transform."""
    need_grad = torch.is_grad_enabled() and (azim.requires_grad or elev.requires_grad or dist.requires_grad)
    if azim.dim() != 1:   # (the renderer hands over flat views already: a second reshape is another tensor object each)
        azim, elev, dist = azim.reshape(-1), elev.reshape(-1), dist.reshape(-1)
    op = _lib.torch_op()
    if op is not None:
        # the same two library calls from a C++ autograd node (csrc/mvr_torch_op.cpp): no Python in the backward, no ctypes
        # marshalling, the buffers allocated natively — the host cost of a training step stays well below its GPU time
        _require_cuda(azim, elev, dist, packed.verts)
        # zero-copy view of the ctypes struct, made per call (1.3 us).  NOT cached on `params`: the tensor's storage holds a
        # reference to `params` inside C++, so params -> tensor -> params would be a cycle the garbage collector cannot see,
        # and every `with_window` copy (one per training step, with four device tensors in `_keep`) would live forever
        pt = torch.frombuffer(params, dtype=torch.uint8)
        keep = getattr(params, "_keep", None)
        seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        return tuple(op.render_mesh_views(_f32c(azim), _f32c(elev), _f32c(dist), None if light_dir is None else _f32c(light_dir), getattr(host_status, "tensor", host_status),
                                          packed.verts, packed.faces, packed.vert_off, packed.face_off, packed.vert_colors,
                                          packed.max_faces, packed.max_verts, pt, [t for t in keep if t is not None] if keep else [],
                                          int(max_trials), seed - (1 << 64) if seed >= (1 << 63) else seed, _stream(), need_grad))
    return _RenderMeshViews.apply(azim, elev, dist, packed, params, light_dir, max_trials, seed, host_status, need_grad)


def render_meshes_fragments(packed: PackedMeshes, R, T, light_dir, params, return_redo=False, return_straddlers=False, return_stats=False):
    """(images, pix_to_face (N,H,W) int32 LOCAL face index / -1) without autograd — for parity tests.
    `return_redo`: also the number of (view, strip) tiles the fast rasteriser left to the exact kernel (a device int);
    `return_straddlers`: also the number of faces that straddled the z-clip plane (and were cut there);
    `return_stats`: also mvr_mesh_fast_stats' four device ints (strips redone, ambiguous pixels, strip-list entries rasterised
    by a whole wave, strip-list entries in all)."""
    lib = _lib.load()
    R, T, light = _f32c(R), _f32c(T), _f32c(light_dir)
    _require_cuda(R, T, light, packed.verts)
    B, V, H, W = packed.n_meshes, params.n_views, params.image_h, params.image_w
    mb = packed.c_struct()
    ws_bytes = lib.mvr_mesh_workspace_bytes(mb, params)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=R.device)
    images = torch.empty(B, V, 3, H, W, dtype=torch.float32, device=R.device)
    p2f = torch.empty(B * V, H, W, dtype=torch.int32, device=R.device)
    _lib.check(lib.mvr_render_meshes_forward(mb, params, R.data_ptr(), T.data_ptr(), light.data_ptr(), images.data_ptr(),
                                             p2f.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "mvr_render_meshes_forward")
    extra = ()
    if return_redo:
        cnt = torch.zeros(1, dtype=torch.int32, device=R.device)
        _lib.check(lib.mvr_mesh_redo_count(mb, params, ws.data_ptr(), ws_bytes, cnt.data_ptr(), _stream()), "mvr_mesh_redo_count")
        extra += (cnt,)
    if return_straddlers:
        cnt = torch.zeros(1, dtype=torch.int32, device=R.device)
        _lib.check(lib.mvr_mesh_clip_straddlers(mb, params, ws.data_ptr(), ws_bytes, cnt.data_ptr(), _stream()), "mvr_mesh_clip_straddlers")
        extra += (cnt,)
    if return_stats:
        cnt = torch.zeros(4, dtype=torch.int32, device=R.device)
        _lib.check(lib.mvr_mesh_fast_stats(mb, params, ws.data_ptr(), ws_bytes, cnt.data_ptr(), _stream()), "mvr_mesh_fast_stats")
        extra += (cnt,)
    return (images, p2f) + extra


# ---------------------------------------------------------------------------------------------
# soft mesh path: blur_radius > 0, faces_per_pixel = K, SoftPhongShader / SoftSilhouetteShader
# ---------------------------------------------------------------------------------------------

def soft_params(blur_radius, faces_per_pixel, shader="soft_phong", sigma=1e-4, gamma=1e-4, clip_barycentric_coords=None):
    """RasterizationSettings(blur_radius, faces_per_pixel, clip_barycentric_coords) + BlendParams(sigma, gamma) of the
    reference's imports (models/renderer.py:3-12,:42,:89-93).  clip_barycentric_coords None = PyTorch3D's rule
    (True when blur_radius > 0)."""
    if clip_barycentric_coords is None:
        clip_barycentric_coords = blur_radius > 0
    sh = {"soft_phong": _lib.MVR_SHADER_SOFT_PHONG, "soft_silhouette": _lib.MVR_SHADER_SOFT_SILHOUETTE}[shader]
    return _lib.SoftParams(float(blur_radius), float(sigma), float(gamma), int(faces_per_pixel), sh, int(bool(clip_barycentric_coords)))


class _RenderMeshesSoft(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, T, light, packed, params, soft):
        lib = _lib.load()
        R, T, light = _f32c(R), _f32c(T), _f32c(light)
        _require_cuda(R, T, light, packed.verts)
        B, V, H, W = packed.n_meshes, params.n_views, params.image_h, params.image_w
        N, K = B * V, soft.faces_per_pixel
        if R.shape[0] != N or T.shape[0] != N:
            raise ValueError("R/T must have B*V = %d rows" % N)
        dev = R.device
        mb = packed.c_struct()
        ws_bytes = lib.mvr_mesh_soft_workspace_bytes(mb, params, soft)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        images = torch.empty(B, V, 4, H, W, dtype=torch.float32, device=dev)
        p2f = torch.empty(N, H, W, K, dtype=torch.int32, device=dev)
        _lib.check(lib.mvr_render_meshes_soft_forward(mb, params, soft, R.data_ptr(), T.data_ptr(), light.data_ptr(), images.data_ptr(),
                                                      p2f.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "mvr_render_meshes_soft_forward")
        ctx.save_for_backward(R, T, light, p2f, ws)
        ctx.packed, ctx.params, ctx.soft = packed, params, soft
        ctx.mark_non_differentiable(p2f)
        return images, p2f

    @staticmethod
    def backward(ctx, g, _gp):
        lib = _lib.load()
        R, T, light, p2f, ws = ctx.saved_tensors
        packed, params, soft = ctx.packed, ctx.params, ctx.soft
        g = _f32c(g)
        N = R.shape[0]
        bparams = _copy_params(params, _lib.MVR_WORKSPACE_FROM_FORWARD)
        gR, gT = torch.empty_like(R), torch.empty_like(T)
        per_view_light = params.n_light == N and N > 0 and ctx.needs_input_grad[2]
        gL = torch.empty(N, 3, dtype=torch.float32, device=R.device) if per_view_light else None
        _lib.check(lib.mvr_render_meshes_soft_backward(packed.c_struct(), bparams, soft, R.data_ptr(), T.data_ptr(), light.data_ptr(),
                                                       p2f.data_ptr(), g.data_ptr(), gR.data_ptr(), gT.data_ptr(), _ptr(gL),
                                                       ws.data_ptr(), ws.numel(), _stream()), "mvr_render_meshes_soft_backward")
        return gR, gT, gL, None, None, None


def render_meshes_soft(packed: PackedMeshes, R, T, light_dir, params, soft, return_fragments=False):
    """RGBA images (B,V,4,H,W) of MeshRenderer(MeshRasterizer(blur_radius, faces_per_pixel), SoftPhongShader /
    SoftSilhouetteShader); differentiable w.r.t. R, T and a per-view light_dir — including the coverage gradient
    through the signed distances (alpha and the softmax weights).  `return_fragments`: also pix_to_face (N,H,W,K)."""
    images, p2f = _RenderMeshesSoft.apply(R, T, light_dir, packed, params, soft)
    return (images, p2f) if return_fragments else images


def render_points_fragments(points, colors, R, T, dist, params):
    """(images, pix_to_point int32: (N,H,W) for points_per_pixel == 1, else (N,H,W,K)) without autograd — for parity tests."""
    lib = _lib.load()
    R, T, dist = _f32c(R), _f32c(T), _f32c(dist)
    B, V, H, W = params.n_clouds, params.n_views, params.image_h, params.image_w
    images = torch.empty(B, V, 3, H, W, dtype=torch.float32, device=R.device)
    p2p = torch.empty(_p2p_shape(params), dtype=torch.int32, device=R.device)
    ws_bytes = lib.mvr_points_workspace_bytes(params)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=R.device)
    _lib.check(lib.mvr_render_points_forward(params, points.data_ptr(), _ptr(colors), R.data_ptr(), T.data_ptr(),
                                             dist.data_ptr(), images.data_ptr(), p2p.data_ptr(), ws.data_ptr(), ws_bytes, _stream()),
               "mvr_render_points_forward")
    return images, p2p


def vertex_normals(packed: PackedMeshes):
    lib = _lib.load()
    mb = packed.c_struct()
    p = mesh_params(1, 8, 1, (1, 1, 1), (1, 1, 1))
    ws_bytes = lib.mvr_mesh_workspace_bytes(mb, p)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=packed.device)
    out = torch.empty(packed.total_verts, 3, dtype=torch.float32, device=packed.device)
    _lib.check(lib.mvr_vertex_normals(mb, out.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "mvr_vertex_normals")
    return out


# ---------------------------------------------------------------------------------------------
# point-cloud path
# ---------------------------------------------------------------------------------------------

def _set_normalize(p, normalize):
    """(mean, std) per channel (or scalars) -> out_mean / out_std of a params struct: Normalize in the rasteriser's stores."""
    if normalize is None:
        return
    mean, std = normalize
    mean = [float(mean)] * 3 if not hasattr(mean, "__len__") else [float(v) for v in mean]
    std = [float(std)] * 3 if not hasattr(std, "__len__") else [float(v) for v in std]
    if len(mean) != 3 or len(std) != 3 or min(std) <= 0:
        raise ValueError("normalize: (mean, std) with three (or one) values each, std > 0")
    for k in range(3):
        p.out_mean[k], p.out_std[k] = mean[k], std[k]


def points_params(n_clouds, n_points, n_views, image_size, radius, obj_color, bg_color, strip_rows=0, points_per_pixel=1, normalize=None):
    """`normalize`: (mean, std): the trainer's Normalize, (rgb - mean) / std, in the rasteriser's stores (see mesh_params)."""
    H, W = (image_size, image_size) if isinstance(image_size, int) else image_size
    p = _lib.PointsParams(n_clouds, n_points, n_views, H, W, strip_rows, int(points_per_pixel), 0, radius)
    for k in range(3):
        p.obj_color[k] = float(obj_color[k])
        p.bg_color[k] = float(bg_color[k])
    _set_normalize(p, normalize)
    return p


def _p2p_shape(params):
    K = max(1, params.points_per_pixel)
    shape = (params.n_clouds * params.n_views, params.image_h, params.image_w)
    return shape if K == 1 else shape + (K,)


class _RenderPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, R, T, dist, points, colors, params, need_grad):
        lib = _lib.load()
        R, T, dist = _f32c(R), _f32c(T), _f32c(dist)
        _require_cuda(R, T, dist, points, colors)
        B, V, H, W = params.n_clouds, params.n_views, params.image_h, params.image_w
        N = B * V
        if R.shape[0] != N or dist.numel() != N:
            raise ValueError("R/T/dist must have B*V = %d rows" % N)
        images = torch.empty(B, V, 3, H, W, dtype=torch.float32, device=R.device)
        p2p = torch.empty(_p2p_shape(params), dtype=torch.int32, device=R.device) if need_grad else None
        ws_bytes = lib.mvr_points_workspace_bytes(params)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=R.device)
        _lib.check(lib.mvr_render_points_forward(params, points.data_ptr(), _ptr(colors), R.data_ptr(), T.data_ptr(),
                                                 dist.data_ptr(), images.data_ptr(), _ptr(p2p), ws.data_ptr(), ws_bytes, _stream()),
                   "mvr_render_points_forward")
        if need_grad:
            ctx.save_for_backward(R, T, dist, points, p2p)
            ctx.colors, ctx.params = colors, params
        return images

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        R, T, dist, points, p2p = ctx.saved_tensors
        params, colors = ctx.params, ctx.colors
        g = _f32c(g)
        ws_bytes = lib.mvr_points_workspace_bytes(params)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=R.device)
        gR, gT, gD = torch.empty_like(R), torch.empty_like(T), torch.empty_like(dist)
        _lib.check(lib.mvr_render_points_backward(params, points.data_ptr(), _ptr(colors), R.data_ptr(), T.data_ptr(),
                                                  dist.data_ptr(), p2p.data_ptr(), g.data_ptr(), gR.data_ptr(), gT.data_ptr(),
                                                  gD.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "mvr_render_points_backward")
        return gR, gT, gD, None, None, None, None


POINT_VIEWS_FUSED_MAX = 2048   # MVR_POINT_VIEWS_FUSED_MAX: views one fused call accepts


def render_point_views(points, colors, azim, elev, dist, params, max_trials=-1, seed=0, host_status=None):
    """(images (B,V,3,H,W), R (N,3,3), T (N,3), status) for flat (N,) azim / elev / dist in degrees: the forward of
    the point path of models/renderer.py:116-151 as ONE library call (look-at inside the projection pre-pass).
    Inference only (no autograd graph): with gradients use look_at_view_transform + render_points."""
    lib = _lib.load()
    azim, elev, dist = _f32c(azim.detach().reshape(-1)), _f32c(elev.detach().reshape(-1)), _f32c(dist.detach().reshape(-1))
    _require_cuda(azim, elev, dist, points, colors)
    B, V, H, W = params.n_clouds, params.n_views, params.image_h, params.image_w
    N = B * V
    if azim.numel() != N or dist.numel() != N:
        raise ValueError("azim/elev/dist must have B*V = %d elements" % N)
    dev = azim.device
    R = torch.empty(N, 3, 3, dtype=torch.float32, device=dev)
    T = torch.empty(N, 3, dtype=torch.float32, device=dev)
    status = torch.empty(N + 2, dtype=torch.int32, device=dev)
    images = torch.empty(B, V, 3, H, W, dtype=torch.float32, device=dev)
    ws_bytes = lib.mvr_points_workspace_bytes(params)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    sp = status.data_ptr() if host_status is None else host_status.data_ptr()
    _lib.check(lib.mvr_render_point_views_forward(params, points.data_ptr(), _ptr(colors), azim.data_ptr(), elev.data_ptr(),
                                                  dist.data_ptr(), int(max_trials), int(seed) & 0xFFFFFFFFFFFFFFFF, R.data_ptr(),
                                                  T.data_ptr(), None, None, status[2:].data_ptr(), sp, images.data_ptr(), None,
                                                  ws.data_ptr(), ws_bytes, _stream()), "mvr_render_point_views_forward")
    return images, R, T, status


def render_points(points, colors, R, T, dist, params):
    """images (B,V,3,H,W) — PointsRenderer(PointsRasterizer, NormWeightedCompositor) over the
    1/dist-scaled clouds of models/renderer.py:129-145 plus the re-layout of :146-150."""
    need_grad = torch.is_grad_enabled() and (R.requires_grad or T.requires_grad or dist.requires_grad)
    return _RenderPoints.apply(R, T, dist, points, colors, params, need_grad)
