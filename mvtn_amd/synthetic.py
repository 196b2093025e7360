"""Seeded synthetic inputs for tests and bench.py (there is no network for datasets).

Meshes: icosphere of subdivision level L with a smooth radial perturbation, centred on
the vertex mean and scaled so max ||v||_2 = 1 — the normalisation the reference applies
to every object (util.py:437-451 torch_center_and_normalize with config.yaml:15
dset_norm "2").  Tiers: S = L3 (642 v / 1 280 f), M = L4 (2 562 v /
5 120 f, the "ModelNet-simplified-sized" assumption), L = L5 (10 242 v / 20 480 f).

View grids: restatement of util.py:359-391 unit_spherical_grid and of the
Circular/Spherical selectors (models/mvtn.py:13-80); checked against golden fixtures
captured from the importable reference code (tests/golden/).
"""
import numpy as np

_TIER_LEVEL = {"S": 3, "M": 4, "L": 5}
_ICO_CACHE = {}


def icosphere(level):
    """Unit icosphere: (verts float64 (Vn,3), faces int64 (F,3)), outward CCW winding."""
    if level in _ICO_CACHE:
        v, f = _ICO_CACHE[level]
        return v.copy(), f.copy()
    t = (1.0 + 5.0 ** 0.5) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t),
             (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
             (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
             (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache = {}

        def mid(a, b):
            key = (a, b) if a < b else (b, a)
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        new_faces = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            new_faces += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = new_faces
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    _ICO_CACHE[level] = (v, f)
    return v.copy(), f.copy()


def center_and_normalize(points):
    """util.py:437-451 with p="2": subtract the mean, divide by the max L2 norm."""
    points = points - points.mean(0, keepdims=True)
    return points / np.linalg.norm(points, axis=1).max()


def synthetic_mesh(tier="M", seed=0, amplitude=0.25):
    """(verts float32 (Vn,3), faces int64 (F,3)) — a lumpy unit-normalised blob."""
    level = _TIER_LEVEL[tier] if isinstance(tier, str) else int(tier)
    v, f = icosphere(level)
    rng = np.random.default_rng(seed)
    noise = np.zeros(len(v))
    for _ in range(6):
        w = rng.normal(size=3) * rng.uniform(1.0, 3.0)
        noise += rng.uniform(0.3, 1.0) * np.sin(v @ w + rng.uniform(0, 2 * np.pi))
    noise /= np.abs(noise).max() + 1e-12
    v = v * (1.0 + amplitude * noise)[:, None]
    v = center_and_normalize(v)
    return v.astype(np.float32), f


def cad_soup(seed=0, n_quads=24):
    """"CAD-like" soup: double-sided coincident quads (two triangles each, both windings),
    exercising the (z, face index) tie rule.  (verts float32, faces int64)."""
    rng = np.random.default_rng(seed)
    verts, faces = [], []
    for _ in range(n_quads):
        c = rng.uniform(-0.5, 0.5, 3)
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        w = np.cross(u, rng.normal(size=3))
        w /= np.linalg.norm(w)
        su, sw = rng.uniform(0.1, 0.5, 2)
        base = len(verts)
        verts += [c - su * u - sw * w, c + su * u - sw * w, c + su * u + sw * w, c - su * u + sw * w]
        faces += [(base, base + 1, base + 2), (base, base + 2, base + 3),
                  (base + 2, base + 1, base), (base + 3, base + 2, base)]
    v = center_and_normalize(np.asarray(verts, np.float64))
    return v.astype(np.float32), np.asarray(faces, np.int64)


def surface_points(verts, faces, n_points, seed=0):
    """n_points uniform on the mesh surface, then centre+normalise (ScanObjectNN-like input)."""
    rng = np.random.default_rng(seed)
    v = verts.astype(np.float64)
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    fi = rng.choice(len(faces), size=n_points, p=area / area.sum())
    r1, r2 = np.sqrt(rng.uniform(size=n_points)), rng.uniform(size=n_points)
    pts = (1 - r1)[:, None] * a[fi] + (r1 * (1 - r2))[:, None] * b[fi] + (r1 * r2)[:, None] * c[fi]
    return center_and_normalize(pts).astype(np.float32)


def unit_spherical_grid(nb_points):
    """(azim, elev) in degrees, float64 — util.py:359-391."""
    alpha = 4.0 * np.pi / nb_points
    d = np.sqrt(alpha)
    m_nu = int(np.round(np.pi / d))
    d_nu = np.pi / m_nu
    d_phi = alpha / d_nu
    azim, elev = [], []
    for m in range(m_nu):
        nu = np.pi * (m + 0.5) / m_nu
        m_phi = int(np.round(2 * np.pi * np.sin(nu) / d_phi))
        for n in range(m_phi):
            azim.append(2 * np.pi * n / m_phi)
            elev.append(nu - np.pi * 0.5)
    return np.rad2deg(azim)[:nb_points], np.rad2deg(elev)[:nb_points]


def circular_views(nb_views, elevation=30.0, distance=2.2):
    """models/mvtn.py:13-28 CircularViewSelector buffers: (azim, elev, dist) float32 (V,)."""
    azim = (np.linspace(-180.0, 180.0, nb_views + 1, dtype=np.float32)[:-1] - np.float32(90.0))
    return (azim.astype(np.float32), np.full(nb_views, elevation, np.float32),
            np.full(nb_views, distance, np.float32))


def spherical_views(nb_views, distance=2.2):
    """models/mvtn.py:48-63 SphericalViewSelector buffers: (azim, elev, dist) float32 (V,)."""
    azim, elev = unit_spherical_grid(nb_views)
    return azim.astype(np.float32), elev.astype(np.float32), np.full(nb_views, distance, np.float32)
