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


def _rot(rng, upright):
    """A part's orientation: axis-aligned up to a quarter turn (what CAD assemblies mostly are), or arbitrary."""
    if upright:
        p = rng.permutation(3)
        m = np.zeros((3, 3))
        m[np.arange(3), p] = rng.choice([-1.0, 1.0], 3)
    else:
        m, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(m) < 0:   # a proper rotation: the part keeps its outward winding
        m[:, 0] = -m[:, 0]
    return m


def _box(h):
    """Axis-aligned box of half-extents h: 8 vertices, every flat side as TWO big triangles (outward winding)."""
    v = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64) * h
    f = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    return v, np.asarray(f, np.int64)


def _cylinder(r, h, nseg, rings, rng):
    """Cylinder along z: the side as nseg x rings quads of two LONG SLIVERS each, the caps as fans from one rim vertex
    (what CAD tessellators emit: no centre vertex) or left open."""
    ang = 2.0 * np.pi * np.arange(nseg) / nseg
    zs = np.linspace(-h, h, rings + 1)
    v = np.array([[r * np.cos(a), r * np.sin(a), z] for z in zs for a in ang], np.float64)
    f = []
    for k in range(rings):
        for i in range(nseg):
            a, b = k * nseg + i, k * nseg + (i + 1) % nseg
            c, d = a + nseg, b + nseg
            f += [(a, b, d), (a, d, c)]
    if rng.uniform() < 0.7:
        top = rings * nseg
        for i in range(1, nseg - 1):
            f += [(0, i + 1, i), (top, top + i, top + i + 1)]
    return v, np.asarray(f, np.int64)


def _panel(hx, hy, nx, ny, bulge):
    """Tessellated (slightly curved) sheet: nx x ny quads — the small-triangle parts of a CAD export (fillets, seats, shells)."""
    xs, ys = np.linspace(-hx, hx, nx + 1), np.linspace(-hy, hy, ny + 1)
    v = np.array([[x, y, bulge * (1.0 - (x / hx) ** 2) * (1.0 - (y / hy) ** 2)] for y in ys for x in xs], np.float64)
    f = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i
            f += [(a, a + 1, a + nx + 2), (a, a + nx + 2, a + nx + 1)]
    return v, np.asarray(f, np.int64)


def cad_mesh(seed=0, n_faces=5120, double_sided=0.3, interior=0.15):
    """A "CAD-like" object of about the reference's simplified-ModelNet40 size (README.md:54,86: decimated CAD models): an
    assembly of boxes, thin plates, cylinders, tessellated panels and knobs around a big body.  What it has that the lumpy
    icospheres of `synthetic_mesh` lack — and what ModelNet40 is made of: flat sides as TWO screen-filling triangles next to
    long slivers (cylinder sides, rim-vertex cap fans) and small tessellated patches; parts that intersect each other; parts
    fully INSIDE the body (never visible); and `double_sided` of the parts present twice with reversed winding — coincident
    double-sided shells, half of them on the same vertex indices, half on a duplicated vertex set with identical coordinates
    (the reference ships lists of wrongly wound meshes: custom_dataset.py:92-98,118-119).  Centred and unit-normalised like
    custom_dataset.py:120-121.  Exactly n_faces faces (the last part is truncated).  (verts float32, faces int64)."""
    rng = np.random.default_rng(seed)
    verts, faces = [], []
    nv = nf = 0

    def add(v, f, centre, upright=True, both=None):
        nonlocal nv, nf
        v = v @ _rot(rng, upright).T + centre
        verts.append(v)
        faces.append(f + nv)
        nf += len(f)
        if (rng.uniform() < double_sided) if both is None else both:
            if rng.uniform() < 0.5:      # the same vertices, reversed winding
                faces.append(f[:, ::-1] + nv)
            else:                        # a second copy of the vertices (bit-identical coordinates), reversed winding
                verts.append(v.copy())
                faces.append(f[:, ::-1] + nv + len(v))
                nv += len(v)
            nf += len(f)
        nv += len(v)

    body = rng.uniform(0.25, 0.6, 3)
    add(*_box(body), np.zeros(3), both=False)
    while nf < n_faces:
        kind = rng.choice(5, p=[0.2, 0.15, 0.4, 0.15, 0.1])
        inside = rng.uniform() < interior
        centre = rng.uniform(-1, 1, 3) * (body * 0.5 if inside else body * 1.6)
        shrink = 0.3 if inside else 1.0
        upright = rng.uniform() < 0.7
        if kind == 0:      # box
            add(*_box(rng.uniform(0.05, 0.5, 3) * shrink), centre, upright)
        elif kind == 1:    # thin plate
            h = rng.uniform(0.15, 0.7, 3) * shrink
            h[rng.integers(3)] = rng.uniform(0.004, 0.02)
            add(*_box(h), centre, upright)
        elif kind == 2:    # cylinder / rod
            add(*_cylinder(rng.uniform(0.02, 0.25) * shrink, rng.uniform(0.1, 0.8) * shrink, int(rng.integers(8, 49)), int(rng.integers(1, 4)), rng),
                centre, upright)
        elif kind == 3:    # tessellated panel
            add(*_panel(rng.uniform(0.1, 0.6) * shrink, rng.uniform(0.1, 0.6) * shrink, int(rng.integers(3, 13)), int(rng.integers(3, 13)),
                        rng.uniform(0.0, 0.15)), centre, upright)
        else:              # knob
            v, f = icosphere(int(rng.integers(1, 3)))
            add(v * rng.uniform(0.03, 0.15) * shrink, f, centre)
    v = center_and_normalize(np.concatenate(verts))
    f = np.concatenate(faces)[:n_faces]
    return v.astype(np.float32), np.ascontiguousarray(f, np.int64)


def ragged_tiers(batch=16):
    """Tier letters of the ragged batch SURVEY.md section 8(d) asks for: S, M and L meshes mixed 1 : 2 : 1, interleaved."""
    return [("S", "M", "L", "M")[i % 4] for i in range(batch)]


def geometry_mesh(geometry, tier, seed):
    """One object of bench.py's --geometry: "ico" (synthetic_mesh of `tier`), "cad" (cad_mesh with the tier's face count),
    "ragged" (synthetic_mesh, the tier following ragged_tiers)."""
    if geometry == "cad":
        import os   # (developer knob for tools/: the share of doubled parts, to separate the cost of coincident shells from the rest)
        return cad_mesh(seed, n_faces=20 * 4 ** _TIER_LEVEL[tier], double_sided=float(os.environ.get("MVR_CAD_DOUBLE_SIDED", "0.3")))
    if geometry == "ragged":
        return synthetic_mesh(ragged_tiers(seed + 1)[seed], seed)
    return synthetic_mesh(tier, seed)


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
