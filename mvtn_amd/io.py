"""Mesh ingestion on the host side of the boundary: what custom_dataset.py obtains from
`trimesh.load` / `pytorch3d.io.load_obj` before it builds the single-mesh `Meshes` objects the renderer receives
(custom_dataset.py:100-143, 250-270) — triangle soup from OFF / OBJ files, the dataset's fixed rotation and the
centre-and-normalise of util.py:437-451.  numpy only; nothing here touches the GPU.
"""
import math

import numpy as np
import torch

from .structures import Meshes


def _fan(poly):
    """Triangle fan of one polygon (what trimesh / PyTorch3D do for the convex polygons of ModelNet / ShapeNet)."""
    return [(poly[0], poly[k], poly[k + 1]) for k in range(1, len(poly) - 1)]


def load_off(path):
    """(verts (n,3) float64, faces (m,3) int64) of an Object File Format mesh.  Accepts the ModelNet quirk of a header
    glued to the counts ("OFF490 518 0"), comments, blank lines, colour columns after the indices and polygons of any
    size (triangulated as a fan)."""
    with open(path, "r") as fh:
        tokens = []
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if line:
                tokens.append(line)
    if not tokens or not tokens[0].upper().startswith("OFF"):
        raise ValueError("%s: not an OFF file" % path)
    head = tokens[0][3:].split()
    rest = tokens[1:]
    if len(head) < 3:
        head = rest[0].split()
        rest = rest[1:]
    nv, nf = int(head[0]), int(head[1])
    if len(rest) < nv + nf:
        raise ValueError("%s: truncated OFF file" % path)
    verts = np.array([[float(x) for x in rest[i].split()[:3]] for i in range(nv)], np.float64).reshape(nv, 3)
    faces = []
    for i in range(nv, nv + nf):
        parts = rest[i].split()
        k = int(parts[0])
        faces.extend(_fan([int(x) for x in parts[1:1 + k]]))
    faces = np.array(faces, np.int64).reshape(-1, 3)
    if faces.size and (faces.min() < 0 or faces.max() >= nv):
        raise ValueError("%s: face index out of range" % path)
    return verts, faces


def load_obj(path):
    """(verts (n,3) float64, faces (m,3) int64) of a Wavefront OBJ: `v` and `f` records only (what custom_dataset.py's
    ShapeNet branch keeps: `faces.verts_idx`), `a/b/c` index triples, negative (relative) indices, polygons as fans."""
    verts, faces = [], []
    with open(path, "r") as fh:
        for line in fh:
            if line.startswith("v "):
                verts.append([float(x) for x in line.split()[1:4]])
            elif line.startswith("f "):
                idx = []
                for tok in line.split()[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                faces.extend(_fan(idx))
    verts = np.array(verts, np.float64).reshape(-1, 3)
    faces = np.array(faces, np.int64).reshape(-1, 3)
    if faces.size and (faces.min() < 0 or faces.max() >= len(verts)):
        raise ValueError("%s: face index out of range" % path)
    return verts, faces


def load_mesh(path):
    ext = path.rsplit(".", 1)[-1].lower()
    if ext == "off":
        return load_off(path)
    if ext == "obj":
        return load_obj(path)
    raise ValueError("unsupported mesh format: %s" % path)


def rotation_matrix(axis, theta, in_degrees=True):
    """The matrix util.py:491-506 returns — the rotation by `theta` about `axis`, counterclockwise when the axis points at the
    viewer — written from Rodrigues' formula R = cos(theta) I + sin(theta) [k]x + (1 - cos(theta)) k k^T with k the unit axis
    (the reference expands the same rotation from Euler parameters: equal to ~1e-16, tests/golden/rotation_matrix.json)."""
    if in_degrees:
        theta = math.radians(theta)
    k = np.asarray(axis, np.float64)
    k = k / np.linalg.norm(k)
    cross = np.array([[0.0, -k[2], k[1]], [k[2], 0.0, -k[0]], [-k[1], k[0], 0.0]])
    return math.cos(theta) * np.eye(3) + math.sin(theta) * cross + (1.0 - math.cos(theta)) * np.outer(k, k)


def torch_center_and_normalize(points, p="inf"):
    """util.py:437-451: subtract the mean, divide by max_i ||x_i||_p (p numeric), the Frobenius norm ("fro") or 1 ("no")."""
    center = points.mean(0)
    if p != "fro" and p != "no":
        scale = torch.max(torch.norm(points - center, p=float(p), dim=1))
    elif p == "fro":
        scale = torch.norm(points - center, p=p)
    else:
        scale = 1.0
    return (points - center) * (1.0 / float(scale))


def mesh_from_file(path, angle=-90.0, rot_axis=(1, 0, 0), dset_norm=2, flip_faces=False):
    """One dataset item as custom_dataset.py:100-132 builds it: load, (optionally) swap the winding, rotate by the
    dataset's fixed angle about x, centre + normalise, wrap as a single-mesh `Meshes` (float32 verts, int64 faces)."""
    verts, faces = load_mesh(path)
    if flip_faces:
        faces = faces[:, ::-1].copy()
    verts = rotation_matrix(rot_axis, angle).dot(verts.T).T
    verts = torch_center_and_normalize(torch.from_numpy(verts).to(torch.float), p=dset_norm)
    return Meshes([verts], [torch.from_numpy(faces)])


# ---- ScanObjectNN point clouds (custom_dataset.py:485-575) ------------------------------------------------------------

def load_scanobjectnn_bin(path, variant="obj_only", suncg=False):
    """custom_dataset.py:553-575 `ScanObjectNN.load_pc_file`: a .bin record is float32 values — the point count, then one
    row per point: x y z nx ny nz r g b instance semantic (11 columns; 3 for the SUNCG export).  "with_bg": every point's
    xyz.  Otherwise: the points of the LARGEST foreground instance (last column, instances 0 / 1 / 2 are background;
    ties go to the smallest instance id, as np.unique + argmax do)."""
    pc = np.fromfile(path, dtype=np.float32)
    pc = pc[1:].reshape((-1, 3 if suncg else 11))
    if variant == "with_bg":
        return np.array(pc[:, 0:3])
    lab = pc[:, -1]
    fg = lab[(lab != 0) & (lab != 1) & (lab != 2)]
    values, counts = np.unique(fg, return_counts=True)
    return np.array(pc[lab == values[np.argmax(counts)], 0:3])


def load_scanobjectnn_h5(path):
    """custom_dataset.py:515-520 (the "hardest" variant): (data (n, P, 3) float32, label (n,)) of an
    *_objectdataset_augmentedrot_scale75.h5 file.  Needs h5py (not part of this image: the error says so)."""
    try:
        import h5py
    except ImportError as e:
        raise ImportError("load_scanobjectnn_h5 needs h5py, which is not installed here; the .bin variants need numpy only") from e
    with h5py.File(path, "r") as f:
        return np.array(f["data"]), np.array(f["label"])


def scanobjectnn_points(points, nb_points=2048, dset_norm=2, rng=None):
    """custom_dataset.py:524-545 `__getitem__`: draw nb_points indices WITH replacement (np.random.randint), convert to a
    float tensor, centre + normalise (util.py:437-451).  `rng`: None = numpy's global generator like the reference, or a
    numpy Generator / RandomState for a reproducible draw."""
    points = np.asarray(points)
    n = points.shape[0]
    if rng is None:
        idx = np.random.randint(n, size=nb_points)
    elif hasattr(rng, "integers"):
        idx = rng.integers(0, n, size=nb_points)
    else:
        idx = rng.randint(n, size=nb_points)
    return torch_center_and_normalize(torch.from_numpy(points[idx, :]).to(torch.float), p=dset_norm)
