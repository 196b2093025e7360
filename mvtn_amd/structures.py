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
        """Pack on the host with numpy (zero-copy views of the CPU tensors; torch's CPU reductions cost tens of
        milliseconds on small integer tensors), then one H2D copy per array."""
        import numpy as np
        pairs = _mesh_pairs(meshes)
        if len(pairs) == 0:
            raise ValueError("empty mesh batch")
        vs = [np.asarray(v.detach().cpu().numpy() if torch.is_tensor(v) else v, dtype=np.float32).reshape(-1, 3) for v, _ in pairs]
        fs = [np.asarray(f.detach().cpu().numpy() if torch.is_tensor(f) else f).reshape(-1, 3) for _, f in pairs]
        nv = [int(v.shape[0]) for v in vs]
        nf = [int(f.shape[0]) for f in fs]
        if min(nv) <= 0 or min(nf) <= 0:
            raise ValueError("every mesh needs at least one vertex and one face")
        for f, n in zip(fs, nv):
            if int(f.min()) < 0 or int(f.max()) >= n:
                raise ValueError("face index out of range")
        verts = torch.from_numpy(np.ascontiguousarray(np.concatenate(vs, 0)))
        faces = torch.from_numpy(np.concatenate(fs, 0).astype(np.int32))
        vo = np.zeros(len(pairs) + 1, np.int32)
        fo = np.zeros(len(pairs) + 1, np.int32)
        vo[1:] = np.cumsum(nv)
        fo[1:] = np.cumsum(nf)
        vc = None
        if vert_colors is not None:  # (B, >=max Vn, 3) padded, as models/renderer.py:76-77 builds it
            vcn = vert_colors.detach().cpu().numpy() if torch.is_tensor(vert_colors) else np.asarray(vert_colors)
            vc = torch.from_numpy(np.ascontiguousarray(np.concatenate([vcn[b, :n].reshape(-1, 3) for b, n in enumerate(nv)], 0),
                                                       dtype=np.float32)).to(device, non_blocking=True)
        return PackedMeshes(verts.to(device, non_blocking=True), faces.to(device, non_blocking=True),
                            torch.from_numpy(vo).to(device, non_blocking=True), torch.from_numpy(fo).to(device, non_blocking=True),
                            len(pairs), max(nv), max(nf), vc)

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
        key = (params.n_views, params.image_h, params.image_w, params.strip_rows)
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
