"""Boundary data types of the renderer.

The reference hands `MVRenderer.forward` a Python list of single-mesh PyTorch3D `Meshes`
(custom_dataset.py:143 + collate_fn :149-188) or one batched `Meshes` (run_mvtn.py:517-525),
and gets back a PyTorch3D cameras object.  PyTorch3D is not a dependency here: `Meshes` is
the duck type the callers need, `PackedMeshes` is the HBM layout the kernels read, `Cameras`
carries R, T for the one consumer of the second return value (util.py:604-618 plot_cameras).
"""
import ctypes

import torch

from . import _lib


class Meshes:
    """Minimal stand-in for pytorch3d.structures.Meshes: a list of (verts (Vn,3) float,
    faces (F,3) int64) pairs with verts_list()/faces_list()/len()/indexing/iteration."""

    def __init__(self, verts, faces, textures=None):
        if torch.is_tensor(verts):
            verts = list(verts)
        if torch.is_tensor(faces):
            faces = list(faces)
        if len(verts) != len(faces):
            raise ValueError("verts and faces lists differ in length")
        self._verts = [torch.as_tensor(v, dtype=torch.float32) for v in verts]
        self._faces = [torch.as_tensor(f, dtype=torch.int64) for f in faces]
        self.textures = textures

    def verts_list(self):
        return self._verts

    def faces_list(self):
        return self._faces

    def __len__(self):
        return len(self._verts)

    def __getitem__(self, i):
        if isinstance(i, int):
            return Meshes([self._verts[i]], [self._faces[i]])
        idx = range(len(self))[i] if isinstance(i, slice) else i
        return Meshes([self._verts[j] for j in idx], [self._faces[j] for j in idx])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def verts_padded(self):
        mx = max(v.shape[0] for v in self._verts)
        out = torch.zeros(len(self), mx, 3)
        for i, v in enumerate(self._verts):
            out[i, : v.shape[0]] = v
        return out

    def extend(self, n):
        return Meshes([v for v in self._verts for _ in range(n)], [f for f in self._faces for _ in range(n)])


def _mesh_pairs(meshes):
    """Accept a list of single-mesh objects, or one batched object (both as in the reference)."""
    if hasattr(meshes, "verts_list") and hasattr(meshes, "faces_list") and not isinstance(meshes, (list, tuple)):
        return list(zip(meshes.verts_list(), meshes.faces_list()))
    return [(m.verts_list()[0], m.faces_list()[0]) for m in meshes]


def _host_array(x, dtype):
    """(n,3) contiguous CPU tensor of `dtype` (no copy when the caller's tensor already is one)."""
    if isinstance(x, torch.Tensor) and x.dtype == dtype and x.device.type == "cpu" and x.dim() == 2 and x.is_contiguous():
        return x        # what a DataLoader hands over: nothing to do
    t = x.detach() if torch.is_tensor(x) else torch.as_tensor(x)
    t = t.to(device="cpu", dtype=dtype).reshape(-1, 3)
    return t if t.is_contiguous() else t.contiguous()


_STAGING = {}   # device -> [ring of (pinned uint8 block, event recorded after its last H2D copy), next slot]


_COPY_STREAMS = {}


def _copy_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _COPY_STREAMS.get(key)
    if st is None:
        st = _COPY_STREAMS[key] = torch.cuda.Stream(device)
    return st


def _staging_block(nbytes, device, ring=3):
    """Host block the packer writes.  GPU: a small ring of pinned blocks, each guarded by the event of the
    copy that last read it (a block is rewritten only after that copy has finished).  CPU: a fresh tensor."""
    if device.type == "cpu":
        return torch.empty(nbytes, dtype=torch.uint8), None
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    st = _STAGING.get(key)
    if st is None:
        st = _STAGING[key] = [[None] * ring, 0]
    slot = st[1]
    st[1] = (slot + 1) % ring
    ent = st[0][slot]
    if ent is None or ent[0].numel() < nbytes:
        ent = st[0][slot] = (torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True), torch.cuda.Event())
    else:
        ent[1].synchronize()
    return ent


class PackedMeshes:
    """A batch of meshes resident in HBM in the layout `struct mvr_mesh_batch` describes:
    verts (sum Vn, 3) f32, faces (sum F, 3) i32 LOCAL indices, prefix-sum offsets.  One H2D copy
    per array (the reference does 2*B per-mesh `.cuda()` copies, models/renderer.py:67-68)."""

    def __init__(self, verts, faces, vert_off, face_off, n_meshes, max_verts, max_faces, vert_colors=None):
        self.verts, self.faces, self.vert_off, self.face_off = verts, faces, vert_off, face_off
        self.n_meshes, self.max_verts, self.max_faces = n_meshes, max_verts, max_faces
        self.vert_colors = vert_colors
        self.total_verts, self.total_faces = verts.shape[0], faces.shape[0]

    def __len__(self):
        return self.n_meshes

    @property
    def device(self):
        return self.verts.device

    @staticmethod
    def from_meshes(meshes, device, vert_colors=None):
        """The host half of models/renderer.py:66-77: `mvr_pack_meshes_host` writes verts, faces (int64 ->
        int32, range-checked) and both prefix sums into ONE pinned staging block; one H2D copy moves it and
        the four arrays are views of that device block (the reference issues 2*B per-mesh `.cuda()` copies)."""
        pairs = _mesh_pairs(meshes)
        if len(pairs) == 0:
            raise ValueError("empty mesh batch")
        n = len(pairs)
        vs = [_host_array(v, torch.float32) for v, _ in pairs]
        fs = [_host_array(f, torch.int64) for _, f in pairs]
        nv = (ctypes.c_int32 * n)(*[v.shape[0] for v in vs])
        nf = (ctypes.c_int32 * n)(*[f.shape[0] for f in fs])
        lib = _lib.load()
        lay = _lib.PackLayout()
        if lib.mvr_pack_meshes_layout(n, nv, nf, ctypes.byref(lay)) != _lib.MVR_OK:
            raise ValueError(lib.mvr_last_error().decode())
        device = torch.device(device)
        block, done = _staging_block(int(lay.bytes), device)
        vp = (ctypes.c_void_p * n)(*[v.data_ptr() for v in vs])
        fp = (ctypes.c_void_p * n)(*[f.data_ptr() for f in fs])
        if lib.mvr_pack_meshes_host(ctypes.byref(lay), vp, nv, fp, nf, block.data_ptr()) != _lib.MVR_OK:
            raise ValueError(lib.mvr_last_error().decode())
        if device.type == "cpu":
            dev_block = block          # a fresh tensor per call (see _staging_block)
        else:
            # the copy runs on a side stream, so that it overlaps the kernels of the PREVIOUS forward still queued
            # on the caller's stream (the host runs ahead of the GPU); the caller's stream then waits for it
            cur, side = torch.cuda.current_stream(device), _copy_stream(device)
            with torch.cuda.stream(side):
                dev_block = block[:int(lay.bytes)].to(device, non_blocking=True)
                done.record(side)                  # the staging block is reusable once this copy has run
            cur.wait_event(done)
            dev_block.record_stream(cur)           # allocated on the side stream's pool, read on the caller's

        def view(at, count, dtype):
            return dev_block[at:at + 4 * count].view(dtype)
        verts = view(int(lay.verts_at), 3 * lay.total_verts, torch.float32).view(-1, 3)
        faces = view(int(lay.faces_at), 3 * lay.total_faces, torch.int32).view(-1, 3)
        vo = view(int(lay.vert_off_at), n + 1, torch.int32)
        fo = view(int(lay.face_off_at), n + 1, torch.int32)
        vc = None
        if vert_colors is not None:  # (B, >=max Vn, 3) padded, as models/renderer.py:76-77 builds it
            vcs = vert_colors.detach().to("cpu", torch.float32)
            vc = torch.cat([vcs[b, :v.shape[0]].reshape(-1, 3) for b, v in enumerate(vs)], 0).contiguous().to(device, non_blocking=True)
        return PackedMeshes(verts, faces, vo, fo, n, int(lay.max_verts), int(lay.max_faces), vc)

    def with_vert_colors(self, vert_colors):
        """A view of this batch (same geometry tensors, nothing copied) carrying per-vertex colours: `vert_colors` is the
        padded (B, >= max Vn, 3) tensor models/renderer.py:76-77 builds, or an already packed (sum Vn, 3) one."""
        vc = vert_colors.detach().to(self.verts.device, torch.float32)
        if vc.dim() == 3:
            off = self.vert_off.cpu().tolist()
            if vc.shape[0] != self.n_meshes or vc.shape[1] < self.max_verts:
                raise ValueError("vert_colors must be (B, >= max Vn, 3) = (%d, >= %d, 3), got %s" % (self.n_meshes, self.max_verts, tuple(vc.shape)))
            vc = torch.cat([vc[b, :off[b + 1] - off[b]] for b in range(self.n_meshes)], 0)
        if tuple(vc.shape) != (self.total_verts, 3):
            raise ValueError("packed vert_colors must be (%d, 3), got %s" % (self.total_verts, tuple(vc.shape)))
        return PackedMeshes(self.verts, self.faces, self.vert_off, self.face_off, self.n_meshes, self.max_verts, self.max_faces,
                            vc.contiguous())

    def c_struct(self):
        cs = getattr(self, "_cs", None)
        if cs is None:   # the tensors of a PackedMeshes are never re-assigned: build the C view once
            cs = self._cs = _lib.MeshBatch(self.n_meshes, self.total_verts, self.total_faces, self.max_faces, self.max_verts, 0,
                                           self.verts.data_ptr(), self.faces.data_ptr(), self.vert_off.data_ptr(),
                                           self.face_off.data_ptr(),
                                           self.vert_colors.data_ptr() if self.vert_colors is not None else None)
        return cs

    def workspace_bytes(self, params):
        """mvr_mesh_workspace_bytes for this batch, remembered per (views, image, strip) shape."""
        key = (params.n_views, params.image_h, params.image_w, params.strip_rows, params.flags & _lib.MVR_KEEP_PIXEL_LISTS)
        cache = self.__dict__.setdefault("_ws_bytes", {})
        n = cache.get(key)
        if n is None:
            n = cache[key] = _lib.load().mvr_mesh_workspace_bytes(self.c_struct(), params)
        return n


class _Transform:
    """Row-vector 4x4 transform, just enough of pytorch3d Transform3d for util.py:604-618."""

    def __init__(self, matrix):
        self._m = matrix  # (N,4,4), points_out = [p 1] @ M

    def get_matrix(self):
        return self._m

    def inverse(self):
        return _Transform(torch.inverse(self._m))

    def transform_points(self, points):
        if points.dim() == 2:
            points = points[None].expand(self._m.shape[0], -1, -1)
        ones = torch.ones_like(points[..., :1])
        out = torch.cat([points, ones], -1).bmm(self._m)
        return out[..., :3] / out[..., 3:]


class Cameras:
    """Second return value of MVRenderer.forward: R (N,3,3), T (N,3) on the device."""

    def __init__(self, R, T, kind="perspective", fov=60.0, znear=1.0, zfar=100.0):
        self.R, self.T, self.kind, self.fov, self.znear, self.zfar = R, T, kind, fov, znear, zfar
        self.device = R.device

    def __len__(self):
        return self.R.shape[0]

    def get_world_to_view_transform(self):
        N = self.R.shape[0]
        M = torch.zeros(N, 4, 4, dtype=self.R.dtype, device=self.R.device)
        M[:, :3, :3] = self.R
        M[:, 3, :3] = self.T
        M[:, 3, 3] = 1.0
        return _Transform(M)

    def get_camera_center(self):
        return -torch.bmm(self.T[:, None, :], self.R.transpose(1, 2))[:, 0, :]
