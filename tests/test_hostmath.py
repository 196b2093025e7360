"""CPU (-m "not gpu"): the product's own per-pixel arithmetic (mvtn_amd/csrc/mvr_math.h — the code the HIP
kernels inline), built for the host with g++ in float and double, against the oracle (forward parity,
bit-exact pixel->face map) and against fp64 finite differences of the oracle (analytic backward, KA9)."""
import ctypes

import numpy as np
import pytest

from mvtn_amd import synthetic as syn


def test_markstein_division_is_ieee_division():
    """div_by(a, b, RN(1/b)) == a / b bit for bit — the identity the rasteriser's shared-reciprocal barycentrics
    rest on.  Random + adversarial operands (mantissas of all ones, powers of two, tiny numerators)."""
    rng = np.random.default_rng(0)
    n = 4_000_000
    a = (rng.standard_normal(n) * np.exp2(rng.integers(-40, 4, n))).astype(np.float32)
    b = (rng.standard_normal(n) * np.exp2(rng.integers(-27, 4, n))).astype(np.float32)
    adv = np.array([np.float32(1) - np.float32(2 ** -24), np.float32(1) + np.float32(2 ** -23), 0.99999994, 3.0, 1.9999999,
                    224.0, 128.0, 1e-8, 8.0], np.float32)
    b[:adv.size * 1000] = np.tile(adv, 1000)
    b[b == 0] = 1
    y = (np.float32(1) / b).astype(np.float32)
    q = (a * y).astype(np.float32)
    # r = fma(-b, q, a) is exact: evaluate in float64 (24x24-bit product + add fits) and round once
    r = (a.astype(np.float64) - b.astype(np.float64) * q.astype(np.float64)).astype(np.float32)
    got = (q.astype(np.float64) + r.astype(np.float64) * y.astype(np.float64)).astype(np.float32)
    want = (a / b).astype(np.float32)
    ok = np.isfinite(want) & (np.abs(want) > 1e-30)
    assert np.array_equal(got[ok], want[ok])


@pytest.mark.parametrize("persp", [True, False])
@pytest.mark.parametrize("cull", [False, True])
def test_forward_parity_with_oracle(oracle, hostmath, persp, cull):
    vl, fl = zip(syn.synthetic_mesh("S", 1), syn.cad_soup(3))
    az = np.array([[10.0, 130.0, 250.0], [33.0, -70.0, 190.0]], np.float32)
    el = np.array([[-40.0, 20.0, 65.0], [0.0, -15.0, 80.0]], np.float32)
    di = np.full((2, 3), 2.2, np.float32)
    o = oracle.render_meshes(vl, fl, az, el, di, 64, perspective_correct=persp, cull_backfaces=cull, mode=1)
    R, T, valid = hostmath.look_at(az, el, di)
    assert np.array_equal(R, o["R"]) and np.array_equal(T, o["T"]) and valid.all()
    nrm = np.concatenate([oracle.vertex_normals(v, f) for v, f in zip(vl, fl)], 0)
    img, p2f = hostmath.mesh_forward(vl, fl, nrm, R, T, [[0, 1, 0]], [1, 1, 1], [1, 1, 1], 3, 64,
                                     flags=(2 if persp else 0) | (1 if cull else 0))
    assert np.array_equal(p2f, o["pix_to_face"]), "pixel->face map differs"
    assert np.abs(img - o["images"]).max() < 2e-6
    assert (p2f >= 0).mean() > 0.2


def test_look_at_valid_flags_and_camera_position(oracle, hostmath):
    az, el, di = [0.0, 10.0, 0.0, 300.0], [89.9999, -89.9999, 90.0, 12.0], [2.2, 1.0, 3.0, 0.7]
    R, T, valid = hostmath.look_at(az, el, di)
    Ro, To = oracle.look_at(az, el, di)
    assert np.array_equal(R, Ro) and np.array_equal(T, To)
    assert np.array_equal(valid, oracle.check_rotation(Ro)[1]) and list(valid) == [False, False, True, True]
    assert np.array_equal(hostmath.camera_position(az, el, di), oracle.camera_position(az, el, di))


def _fd(fn, x, h):
    g = np.zeros_like(x)
    for idx in np.ndindex(*x.shape):
        p, m = x.copy(), x.copy()
        p[idx] += h
        m[idx] -= h
        g[idx] = (fn(p) - fn(m)) / (2 * h)
    return g


@pytest.mark.parametrize("persp,relative", [(True, True), (True, False), (False, True)])
def test_ka9_mesh_backward_vs_finite_differences(oracle, hostmath, persp, relative):
    vl = [syn.synthetic_mesh("S", 1)[0].astype(np.float64), syn.synthetic_mesh(2, 5)[0].astype(np.float64)]
    fl = [syn.synthetic_mesh("S", 1)[1], syn.synthetic_mesh(2, 5)[1]]
    az = np.array([[10.0, 130.0], [33.0, -70.0]])
    el = np.array([[-40.0, 20.0], [5.0, -15.0]])
    di = np.array([[2.2, 1.9], [2.5, 2.0]])
    S, V, N = 40, 2, 4
    obj = (0.9, 0.8, 0.7)
    R, T = oracle.look_at(az, el, di, np.float64)
    light = oracle.camera_position(az, el, di, np.float64) if relative else np.array([[0.3, 1.0, -0.2]])
    o = oracle.render_meshes(vl, fl, az, el, di, S, light_dir=light, perspective_correct=persp, dtype=np.float64, R=R, T=T,
                             obj_color=obj)
    p2f = o["pix_to_face"]
    G = np.random.default_rng(0).normal(size=o["images"].shape)

    def L(R_, T_, l_):
        oo = oracle.render_meshes(vl, fl, az, el, di, S, light_dir=l_, perspective_correct=persp, dtype=np.float64, R=R_,
                                  T=T_, fixed_p2f=p2f, obj_color=obj, return_fragments=False)
        return float((oo["images"] * G).sum())
    assert abs(L(R, T, light) - float((o["images"] * G).sum())) < 1e-9
    nrm = np.concatenate([oracle.vertex_normals(x, y) for x, y in zip(vl, fl)], 0)
    gR, gT, gL = hostmath.mesh_backward(vl, fl, nrm, R, T, light, obj, V, S, p2f, G, flags=2 if persp else 0)
    h = 1e-6
    fdR = _fd(lambda x: L(x, T, light), R, h)
    fdT = _fd(lambda x: L(R, x, light), T, h)
    assert np.abs(gR - fdR).max() < 1e-6 * np.abs(fdR).max()
    assert np.abs(gT - fdT).max() < 1e-6 * np.abs(fdT).max()
    if relative:
        fdL = _fd(lambda x: L(R, T, x), light, h)
        assert np.abs(gL - fdL).max() < 1e-6 * np.abs(fdL).max()


def test_look_at_and_camera_position_backward(oracle, hostmath):
    rng = np.random.default_rng(1)
    az, el, di = np.array([10.0, 130.0, 33.0, -70.0, 0.0]), np.array([-40.0, 20.0, 5.0, -15.0, 90.0]), np.array([2.2, 1.9, 2.5, 2.0, 2.2])
    n = az.size
    gR, gT, gC = rng.normal(size=(n, 3, 3)), rng.normal(size=(n, 3)), rng.normal(size=(n, 3))

    def LL(a, e, d):
        R, T = oracle.look_at(a, e, d, np.float64)
        return float((R * gR).sum() + (T * gT).sum())

    def LC(a, e, d):
        return float((oracle.camera_position(a, e, d, np.float64) * gC).sum())
    ga, ge, gd = hostmath.look_at_bwd(az, el, di, gR, gT)
    ca, ce, cd = hostmath.camera_position_bwd(az, el, di, gC)
    h = 1e-6
    for got, got_c, k in ((ga, ca, 0), (ge, ce, 1), (gd, cd, 2)):
        def wrap(fn):
            def g(x):
                args = [az, el, di]
                args[k] = x
                return fn(*args)
            return g
        fd = _fd(wrap(LL), [az, el, di][k], h)
        fdc = _fd(wrap(LC), [az, el, di][k], h)
        # index 4 is the pole (elev = 90, x-axis repair branch): finite differences straddle the branch there
        assert np.abs(got[:4] - fd[:4]).max() < 1e-6 * max(1.0, np.abs(fd).max())
        assert np.abs(got_c - fdc).max() < 1e-6 * max(1.0, np.abs(fdc).max())
    assert np.isfinite(ga).all() and np.isfinite(ge).all()
