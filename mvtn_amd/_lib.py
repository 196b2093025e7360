"""ctypes binding of libmvr_hip.so (include/mvr.h).  There is NO fallback: if the HIP library
is missing the import of any rendering entry point fails loudly."""
import ctypes
import os

from . import build as _build

c_float_p = ctypes.POINTER(ctypes.c_float)
c_int32_p = ctypes.POINTER(ctypes.c_int32)
_vp = ctypes.c_void_p

MVR_OK = 0
MVR_ABI_VERSION = 103   # == MVR_ABI_VERSION of include/mvr.h == mvr_version() of the library
MVR_CULL_BACKFACES = 1
MVR_PERSPECTIVE_CORRECT = 2
MVR_WORKSPACE_FROM_FORWARD = 4
MVR_RASTER_EXACT_ONLY = 8
MVR_BACKWARD_GATHER = 16
MVR_KEEP_PIXEL_LISTS = 32
MVR_NORMALS_THREE_CROSS = 64
MVR_PROJECTION_FMA = 128
MVR_SHADER_SOFT_PHONG, MVR_SHADER_SOFT_SILHOUETTE = 0, 1


class MeshBatch(ctypes.Structure):  # struct mvr_mesh_batch
    _fields_ = [("n_meshes", ctypes.c_int32), ("total_verts", ctypes.c_int32), ("total_faces", ctypes.c_int32),
                ("max_faces", ctypes.c_int32), ("max_verts", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("verts", _vp), ("faces", _vp), ("vert_off", _vp), ("face_off", _vp), ("vert_colors", _vp)]


class MeshParams(ctypes.Structure):  # struct mvr_mesh_params
    _fields_ = [("n_views", ctypes.c_int32), ("image_h", ctypes.c_int32), ("image_w", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("n_light", ctypes.c_int32), ("strip_rows", ctypes.c_int32),
                ("fov_deg", ctypes.c_float), ("ambient", ctypes.c_float), ("diffuse", ctypes.c_float),
                ("specular", ctypes.c_float), ("shininess", ctypes.c_float),
                ("obj_color", ctypes.c_float * 3), ("bg_color", ctypes.c_float * 3), ("z_clip", ctypes.c_float),
                ("out_mean", ctypes.c_float * 3), ("out_std", ctypes.c_float * 3),
                ("win_dx", ctypes.c_void_p), ("win_dy", ctypes.c_void_p), ("win_flip", ctypes.c_void_p), ("win_scale", ctypes.c_void_p),
                ("win_pad", ctypes.c_int32), ("fov_scale", ctypes.c_float)]


class SoftParams(ctypes.Structure):  # struct mvr_soft_params
    _fields_ = [("blur_radius", ctypes.c_float), ("sigma", ctypes.c_float), ("gamma", ctypes.c_float),
                ("faces_per_pixel", ctypes.c_int32), ("shader", ctypes.c_int32), ("clip_barycentric_coords", ctypes.c_int32)]


class PointsParams(ctypes.Structure):  # struct mvr_points_params
    _fields_ = [("n_clouds", ctypes.c_int32), ("n_points", ctypes.c_int32), ("n_views", ctypes.c_int32),
                ("image_h", ctypes.c_int32), ("image_w", ctypes.c_int32), ("strip_rows", ctypes.c_int32),
                ("points_per_pixel", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("radius", ctypes.c_double), ("obj_color", ctypes.c_float * 3), ("bg_color", ctypes.c_float * 3),
                ("out_mean", ctypes.c_float * 3), ("out_std", ctypes.c_float * 3),
                ("win_dx", ctypes.c_void_p), ("win_dy", ctypes.c_void_p), ("win_flip", ctypes.c_void_p), ("win_scale", ctypes.c_void_p),
                ("win_pad", ctypes.c_int32), ("reserved2_", ctypes.c_int32)]


class PackLayout(ctypes.Structure):  # struct mvr_pack_layout
    _fields_ = [("n_meshes", ctypes.c_int32), ("total_verts", ctypes.c_int32), ("total_faces", ctypes.c_int32),
                ("max_verts", ctypes.c_int32), ("max_faces", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("verts_at", ctypes.c_uint64), ("faces_at", ctypes.c_uint64), ("vert_off_at", ctypes.c_uint64),
                ("face_off_at", ctypes.c_uint64), ("bytes", ctypes.c_uint64)]


# name -> (restype, argtypes); mirrors include/mvr.h one to one
_MB, _MP, _PP = ctypes.POINTER(MeshBatch), ctypes.POINTER(MeshParams), ctypes.POINTER(PointsParams)
_PL = ctypes.POINTER(PackLayout)
PROTOTYPES = {
    "mvr_regularize_views_forward": (ctypes.c_int, [_vp] + [ctypes.c_int32] * 4 + [_vp] * 4 + [ctypes.c_int32, _vp, _vp]),
    "mvr_regularize_views_backward": (ctypes.c_int, [_vp] + [ctypes.c_int32] * 4 + [_vp] * 4 + [ctypes.c_int32, _vp, _vp]),
    "mvr_regularize_normalize_views_forward": (ctypes.c_int, [_vp] + [ctypes.c_int32] * 4 + [_vp] * 4 + [ctypes.c_int32, c_float_p, c_float_p, _vp, _vp]),
    "mvr_regularize_normalize_views_backward": (ctypes.c_int, [_vp] + [ctypes.c_int32] * 4 + [_vp] * 4 + [ctypes.c_int32, c_float_p, c_float_p, _vp, _vp]),
    "mvr_pack_meshes_layout": (ctypes.c_int, [ctypes.c_int32, _vp, _vp, _PL]),
    "mvr_pack_meshes_host": (ctypes.c_int, [_PL, _vp, _vp, _vp, _vp, _vp]),
    "mvr_version": (ctypes.c_int, []),
    "mvr_source_hash": (ctypes.c_char_p, []),
    "mvr_last_error": (ctypes.c_char_p, []),
    "mvr_render_mesh_views_forward": (ctypes.c_int, [_MB, _MP, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                                     ctypes.c_uint64] + [ctypes.c_void_p] * 11 + [ctypes.c_size_t, ctypes.c_void_p]),
    "mvr_render_mesh_views_backward": (ctypes.c_int, [_MB, _MP] + [ctypes.c_void_p] * 8 + [ctypes.c_int32] + [ctypes.c_void_p] * 6 +
                                       [ctypes.c_size_t, ctypes.c_void_p]),
    "mvr_selftest_rcp": (ctypes.c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "mvr_profile_begin": (ctypes.c_int, []),
    "mvr_profile_end": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32)]),
    "mvr_look_at": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "mvr_look_at_checked": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvr_look_at_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mvr_camera_position": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp]),
    "mvr_camera_position_backward": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "mvr_mesh_workspace_bytes": (ctypes.c_size_t, [_MB, _MP]),
    "mvr_mesh_soft_workspace_bytes": (ctypes.c_size_t, [_MB, _MP, ctypes.POINTER(SoftParams)]),
    "mvr_render_meshes_soft_forward": (ctypes.c_int, [_MB, _MP, ctypes.POINTER(SoftParams)] + [_vp] * 6 + [ctypes.c_size_t, _vp]),
    "mvr_render_meshes_soft_backward": (ctypes.c_int, [_MB, _MP, ctypes.POINTER(SoftParams)] + [_vp] * 9 + [ctypes.c_size_t, _vp]),
    "mvr_mesh_clip_straddlers": (ctypes.c_int, [_MB, _MP, _vp, ctypes.c_size_t, _vp, _vp]),
    "mvr_mesh_redo_count": (ctypes.c_int, [_MB, _MP, _vp, ctypes.c_size_t, _vp, _vp]),
    "mvr_mesh_fast_stats": (ctypes.c_int, [_MB, _MP, _vp, ctypes.c_size_t, _vp, _vp]),
    "mvr_vertex_normals": (ctypes.c_int, [_MB, _vp, _vp, ctypes.c_size_t, _vp]),
    "mvr_render_meshes_forward": (ctypes.c_int, [_MB, _MP, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "mvr_render_meshes_backward": (ctypes.c_int, [_MB, _MP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "mvr_points_workspace_bytes": (ctypes.c_size_t, [_PP]),
    "mvr_render_points_forward": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
    "mvr_render_point_views_forward": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, ctypes.c_uint64] + [_vp] * 9 +
                                       [ctypes.c_size_t, _vp]),
    "mvr_render_points_backward": (ctypes.c_int, [_PP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_size_t, _vp]),
}

_LIB = None


def lib_path():
    # MVR_LIB: developer override used by tools/ (diagnostic builds); the default is the in-tree product library
    return os.environ.get("MVR_LIB") or _build.LIB_PATH


def load():
    """Load libmvr_hip.so (does not touch the GPU).  Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    unverified = False
    if not os.environ.get("MVR_LIB") and not _build.is_current():
        # the in-tree library is missing or was built from other sources than the ones on disk: (re)build it now — hipcc
        # cross-compiles without a GPU; concurrent ranks serialise on a file lock and only the first one compiles
        try:
            _build.build_hip()
        except Exception as e:  # noqa: BLE001
            # a deployed tree may ship the library without its side file, on a node without hipcc or with a read-only
            # package directory: the library carries the hash of its sources itself (mvr_source_hash) — checked below
            if not os.path.exists(path):
                raise ImportError(
                    "mvtn_amd: %s is missing and could not be built (%s) — the HIP extension is the only rendering path "
                    "(no CPU fallback).  Build it with `python -c 'import __graft_entry__ as g; g.build()'`." % (path, e))
            unverified = True
    if not os.path.exists(path):
        raise ImportError("mvtn_amd: %s does not exist" % path)
    lib = ctypes.CDLL(path)
    # ABI generation: bumped whenever a struct layout or an entry point's contract changes (include/mvr.h: MVR_ABI_VERSION);
    # a library of another generation is never used, whoever built it
    ver_fn = getattr(lib, "mvr_version", None)
    ver = ver_fn() if ver_fn is not None else -1
    if ver != MVR_ABI_VERSION:
        raise ImportError("mvtn_amd: %s has ABI version %d, this package binds version %d (include/mvr.h) — rebuild it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'`" % (path, ver, MVR_ABI_VERSION))
    if unverified:
        # the library could not be rebuilt here: it must carry the hash of exactly the sources beside it.  A stale library is
        # refused (struct layouts move between builds) unless the caller opts in with MVR_ALLOW_STALE_LIB=1.
        hash_fn = getattr(lib, "mvr_source_hash", None)   # (a library that predates mvr_source_hash is stale by definition)
        have = ""
        if hash_fn is not None:
            hash_fn.restype = ctypes.c_char_p
            have = (hash_fn() or b"").decode()
        if have != _build.source_hash():
            msg = ("mvtn_amd: %s was built from other sources than the ones beside it (library %s..., tree %s...) and could not be "
                   "rebuilt here" % (path, have[:12] or "<none>", _build.source_hash()[:12]))
            if os.environ.get("MVR_ALLOW_STALE_LIB") != "1":
                raise ImportError(msg + "; set MVR_ALLOW_STALE_LIB=1 to load it anyway")
            import warnings
            warnings.warn(msg + "; loading it as it is (MVR_ALLOW_STALE_LIB=1)", RuntimeWarning)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


_TORCH_OP = False   # False: not tried yet; None: unavailable (the Python autograd node of functional.py is used)


def torch_op():
    """`torch.ops.mvr` — the mesh step as ONE torch operator with a C++ autograd node (csrc/mvr_torch_op.cpp: host-side
    plumbing over the same C ABI calls) — or None: MVR_TORCH_OP=0, or the extension is not there and cannot be built here
    (no g++ / read-only tree).  Then the Python `torch.autograd.Function` of functional.py issues the same calls."""
    global _TORCH_OP
    if _TORCH_OP is not False:
        return _TORCH_OP
    _TORCH_OP = None
    if os.environ.get("MVR_TORCH_OP", "1") == "0":
        return None
    import torch
    try:
        path = _build.build_torch_op()
    except Exception as e:  # noqa: BLE001
        import warnings
        warnings.warn("mvtn_amd: %s could not be %s (%s); the slower Python autograd node is used (MVR_TORCH_OP=0 silences this)" % (
            _build.TORCH_OP_PATH, "rebuilt" if os.path.exists(_build.TORCH_OP_PATH) else "built", e), RuntimeWarning)
        return None
    lib = load()
    try:
        torch.ops.load_library(path)
        addr = [ctypes.cast(getattr(lib, n), ctypes.c_void_p).value for n in
                ("mvr_render_mesh_views_forward", "mvr_render_mesh_views_backward", "mvr_render_meshes_backward", "mvr_look_at_backward",
                 "mvr_camera_position_backward", "mvr_mesh_workspace_bytes", "mvr_last_error")]
        torch.ops.mvr.init(addr)
        _TORCH_OP = torch.ops.mvr
    except Exception as e:  # noqa: BLE001
        import warnings
        warnings.warn("mvtn_amd: %s did not load (%s); the Python autograd node is used" % (path, e), RuntimeWarning)
    return _TORCH_OP


def check(rc, what=""):
    if rc != MVR_OK:
        msg = load().mvr_last_error()
        raise RuntimeError("mvr %s failed (code %d): %s" % (what, rc, msg.decode() if msg else ""))
