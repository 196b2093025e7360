// mvr_math.h — per-view / per-face / per-pixel arithmetic of the renderer, shared by every
// kernel in mvr_kernels.hip.  Templated on the scalar so the same code is instantiated as
// float on the device and (tests only, host build) as double for finite-difference checks.
//
// Everything that decides COVERAGE or DEPTH ORDER (view transform, projection, edge
// functions, barycentrics, interpolated z) is written as explicit IEEE operations in a fixed
// order and compiled with -ffp-contract=off: the pixel->face map must not depend on FMA
// contraction.  The semantics are PyTorch3D 0.7.x as called from
// models/renderer.py:65-151.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MVR_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define MVR_HD inline
#endif

namespace mvr {

constexpr float kEps = 1e-8f;          // PyTorch3D rasteriser kEpsilon
constexpr float kNormalizeEps = 1e-6f; // F.normalize eps in shading / vertex normals
constexpr float kLookAtEps = 1e-5f;    // F.normalize eps in look_at_rotation
constexpr double kPi = 3.14159265358979323846;

template <typename T> MVR_HD T tmin(T a, T b) { return a < b ? a : b; }
template <typename T> MVR_HD T tmax(T a, T b) { return a > b ? a : b; }
MVR_HD float tsqrt(float x) { return sqrtf(x); }
MVR_HD double tsqrt(double x) { return sqrt(x); }
MVR_HD float tabs(float x) { return fabsf(x); }
MVR_HD double tabs(double x) { return fabs(x); }

MVR_HD float tfma(float a, float b, float c) { return fmaf(a, b, c); }
MVR_HD double tfma(double a, double b, double c) { return fma(a, b, c); }

// a / b given y = RN(1/b) (the correctly rounded reciprocal, e.g. from one IEEE division shared by
// several numerators): q = RN(a y); r = a - b q (exact, one FMA); RN(q + r y) = RN(a / b) — Markstein's
// theorem; bit-identical to the IEEE quotient whenever no intermediate underflows (checked by brute force
// in tests/test_hostmath.py).  3 VALU ops instead of the ~14 of a correctly rounded f32 division.
template <typename T> MVR_HD T div_by(T a, T b, T y) {
  T q = a * y;
  T r = tfma(-b, q, a);
  return tfma(r, y, q);
}

// Reciprocal / reciprocal square root for arithmetic that does NOT decide coverage or depth order (shading
// colours, gradients): the 1-ulp hardware approximations on the device, exact on the host.
MVR_HD float trcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
MVR_HD double trcp(double x) { return 1.0 / x; }
MVR_HD float trsq(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsqf(x);
#else
  return 1.0f / sqrtf(x);
#endif
}
MVR_HD double trsq(double x) { return 1.0 / sqrt(x); }

template <typename T> struct V3 { T x, y, z; };

template <typename T> MVR_HD V3<T> cross(const V3<T> &a, const V3<T> &b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <typename T> MVR_HD T dot(const V3<T> &a, const V3<T> &b) {
  return (a.x * b.x + a.y * b.y) + a.z * b.z;
}

// torch.nn.functional.normalize: v / max(||v||, eps); also returns the clamped norm.
template <typename T> MVR_HD V3<T> normalize(const V3<T> &v, T eps, T *norm_out = nullptr, T *raw_norm = nullptr) {
  T n = tsqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
  T d = tmax(n, eps);
  if (norm_out) *norm_out = d;
  if (raw_norm) *raw_norm = n;
  return {v.x / d, v.y / d, v.z / d};
}

// Shading-only variant: one reciprocal shared by the three components (div_by keeps it equal to the
// three IEEE divisions of F.normalize).
template <typename T> MVR_HD V3<T> normalize_shared(const V3<T> &v, T eps, T *norm_out = nullptr, T *raw_norm = nullptr) {
  T n = tsqrt((v.x * v.x + v.y * v.y) + v.z * v.z);
  T d = tmax(n, eps);
  if (norm_out) *norm_out = d;
  if (raw_norm) *raw_norm = n;
  T y = (T)1 / d;
  return {div_by(v.x, d, y), div_by(v.y, d, y), div_by(v.z, d, y)};
}

// Gradient-only arithmetic (the backward kernels; no coverage or depth decision depends on it): the library is built with
// -ffp-contract=off for the exact paths, and a helper defined outside a `#pragma clang fp contract(fast)` scope keeps its
// separate multiplies and adds even when it is inlined into one — the backward's pixel loop had 124 v_mul + 65 v_add / v_sub
// against 34 fused multiply-adds.  These variants carry the pragma themselves.
template <typename T> MVR_HD T dot_c(const V3<T> &a, const V3<T> &b) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  return (a.x * b.x + a.y * b.y) + a.z * b.z;
}
template <typename T> MVR_HD T edge_fn_c(T px, T py, T ax, T ay, T bx, T by) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

// v / max(||v||, eps) through one reciprocal square root: inv_d = 1/max(||v||, eps); clamped <=> ||v|| <= eps.
template <typename T> MVR_HD V3<T> normalize_rs(const V3<T> &v, T eps, T *inv_d, bool *clamped) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  T n2 = (v.x * v.x + v.y * v.y) + v.z * v.z;
  *clamped = !(n2 > eps * eps);
  T r = trsq(tmax(n2, eps * eps));
  *inv_d = r;
  return {v.x * r, v.y * r, v.z * r};
}
template <typename T> MVR_HD V3<T> normalize_rs_bwd(const V3<T> &out, T inv_d, bool clamped, const V3<T> &g) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  if (clamped) return {g.x * inv_d, g.y * inv_d, g.z * inv_d};
  T og = dot_c(out, g);
  return {(g.x - out.x * og) * inv_d, (g.y - out.y * og) * inv_d, (g.z - out.z * og) * inv_d};
}

// gradient of normalize: out = v/d, d = max(n, eps).  n > eps: (g - out (out.g))/d ; else g/d.
template <typename T> MVR_HD V3<T> normalize_bwd(const V3<T> &out, T d, bool clamped, const V3<T> &g) {
  if (clamped) return {g.x / d, g.y / d, g.z / d};
  T og = dot(out, g);
  return {(g.x - out.x * og) / d, (g.y - out.y * og) / d, (g.z - out.z * og) / d};
}

// ---- cameras --------------------------------------------------------------------------

#if defined(__HIP_DEVICE_COMPILE__)
// sin and cos of a double on the device, for |x| < 1.5e6 (any angle a view selector can produce, in radians): Cody-Waite
// reduction by pi/2 in three parts + the fdlibm kernel polynomials (errors < 1 ulp of a DOUBLE).  The device library's
// generic sincos(double) spends several hundred instructions per call (it carries the Payne-Hanek path for huge
// arguments); it was the largest single consumer of VALU time in the pre-pass kernels, where every view workgroup
// evaluates the look-at of all views.  The result is only ever rounded to float: like the library's it equals the
// correctly rounded float unless the true value lies within ~1e-16 (relative) of a float rounding boundary.
__device__ __forceinline__ void sincos_f64_small(double x, double *sn, double *cs) {
  const double k = __builtin_rint(x * 6.36619772367581382433e-01);   // 2 / pi
  double r = __builtin_fma(-k, 1.57079632673412561417e+00, x);      // pi/2, first 33 bits
  r = __builtin_fma(-k, 6.07710050630396597660e-11, r);             // next 33 bits
  r = __builtin_fma(-k, 2.02226624871116645580e-21, r);             // next 33 bits
  const double tail = __builtin_fma(-k, 8.47842766036889956997e-32, 0.0);   // what is left of pi/2 beyond 99 bits
  const double z = r * r;
  // __kernel_sin / __kernel_cos of fdlibm on [-pi/4, pi/4]
  const double ps = z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                    z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  const double s_ = r + (r * z) * (-1.66666666666666324348e-01 + ps) + tail;
  const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                    z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  const double hz = 0.5 * z;
  const double w = 1.0 - hz;
  const double c_ = w + (((1.0 - w) - hz) + (z * pc - r * tail));
  const int q = (int)k & 3;
  const double ss = (q & 1) ? c_ : s_, cc = (q & 1) ? s_ : c_;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}
#endif

// sin/cos of an angle in degrees: rad = float(pi/180) * deg in T, then correctly rounded
// sin/cos (evaluate in double, round once).
template <typename T> MVR_HD void sincos_deg(T deg, T *s, T *c) {
  T rad = (T)(kPi / 180.0) * deg;
#if defined(__HIP_DEVICE_COMPILE__)
  double sd, cd;
  const double xd = (double)rad;
  if (__builtin_fabs(xd) < 1.5e6) sincos_f64_small(xd, &sd, &cd);
  else sincos(xd, &sd, &cd);    // (huge or non-finite angles: the library's full-range path)
  *s = (T)sd; *c = (T)cd;
#else
  *s = (T)sin((double)rad);
  *c = (T)cos((double)rad);
#endif
}

// camera_position_from_spherical_angles (models/renderer.py:168-169)
// ... from the sines / cosines of the two angles (so that two lanes can evaluate them side by side)
template <typename T> MVR_HD V3<T> camera_position_sc(T se, T ce, T sa, T ca, T dist) {
  return {(dist * ce) * sa, dist * se, (dist * ce) * ca};
}
template <typename T> MVR_HD V3<T> camera_position(T azim, T elev, T dist) {
  T se, ce, sa, ca;
  sincos_deg(elev, &se, &ce);
  sincos_deg(azim, &sa, &ca);
  return camera_position_sc(se, ce, sa, ca, dist);
}

// d(position)/d(azim, elev, dist) contracted with g (angles in degrees).
template <typename T> MVR_HD void camera_position_bwd(T azim, T elev, T dist, const V3<T> &g, T *g_azim, T *g_elev, T *g_dist) {
  T se, ce, sa, ca;
  sincos_deg(elev, &se, &ce);
  sincos_deg(azim, &sa, &ca);
  const T k = (T)(kPi / 180.0);
  *g_dist = (g.x * (ce * sa) + g.y * se) + g.z * (ce * ca);
  *g_elev = k * (dist * ((g.x * (-se * sa) + g.y * ce) + g.z * (-se * ca)));
  *g_azim = k * (dist * (g.x * (ce * ca) - g.z * (ce * sa)));
}

// look_at_view_transform (models/renderer.py:79-80): R row-major with axes as columns, T = -(R^T C).
template <typename T> struct LookAtState {  // what the backward needs
  V3<T> C, x, y, z, tx, ty;
  T dz, dx, dy;           // clamped norms
  bool cz, cx, cy;        // clamp active
  bool repaired;
};

template <typename T> MVR_HD void look_at_from_position(const V3<T> &C, T R[9], T Tr[3], LookAtState<T> *st = nullptr);
template <typename T> MVR_HD void look_at(T azim, T elev, T dist, T R[9], T Tr[3], LookAtState<T> *st = nullptr) {
  look_at_from_position(camera_position(azim, elev, dist), R, Tr, st);
}
template <typename T> MVR_HD void look_at_from_position(const V3<T> &C, T R[9], T Tr[3], LookAtState<T> *st) {
  V3<T> up = {(T)0, (T)1, (T)0};
  V3<T> neg = {(T)0 - C.x, (T)0 - C.y, (T)0 - C.z};
  T dz, nz, dx, nx, dy, ny;
  V3<T> z = normalize(neg, (T)kLookAtEps, &dz, &nz);
  V3<T> tx = cross(up, z);
  V3<T> x = normalize(tx, (T)kLookAtEps, &dx, &nx);
  V3<T> ty = cross(z, x);
  V3<T> y = normalize(ty, (T)kLookAtEps, &dy, &ny);
  bool repaired = false;
  if (tabs(x.x) <= (T)5e-3 && tabs(x.y) <= (T)5e-3 && tabs(x.z) <= (T)5e-3) {
    tx = cross(y, z);
    x = normalize(tx, (T)kLookAtEps, &dx, &nx);
    repaired = true;
  }
  R[0] = x.x; R[1] = y.x; R[2] = z.x;
  R[3] = x.y; R[4] = y.y; R[5] = z.y;
  R[6] = x.z; R[7] = y.z; R[8] = z.z;
  Tr[0] = -((R[0] * C.x + R[3] * C.y) + R[6] * C.z);
  Tr[1] = -((R[1] * C.x + R[4] * C.y) + R[7] * C.z);
  Tr[2] = -((R[2] * C.x + R[5] * C.y) + R[8] * C.z);
  if (st) {
    st->C = C; st->x = x; st->y = y; st->z = z; st->tx = tx; st->ty = ty;
    st->dz = dz; st->dx = dx; st->dy = dy;
    st->cz = !(nz > (T)kLookAtEps); st->cx = !(nx > (T)kLookAtEps); st->cy = !(ny > (T)kLookAtEps);
    st->repaired = repaired;
  }
}

// check_valid_rotation_matrix (util.py:403-420) for one matrix.
template <typename T> MVR_HD bool rotation_valid(const T M[9], T tol) {
  bool ok = true;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      T v = (M[3 * r] * M[3 * c] + M[3 * r + 1] * M[3 * c + 1]) + M[3 * r + 2] * M[3 * c + 2];
      T e = (r == c) ? (T)1 : (T)0;
      T lim = tol + (T)1e-5 * e;
      if (!(tabs(v - e) <= lim)) ok = false;
    }
  T det = (M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6])) +
          M[2] * (M[3] * M[7] - M[4] * M[6]);
  if (!(tabs(det - (T)1) <= (T)1e-8 + (T)1e-5)) ok = false;
  return ok;
}

// backward of look_at: gR[9] (dL/dR row-major), gT[3] -> g_azim, g_elev, g_dist (degrees).
// The part of look_at_bwd that does not need the upstream gradients (the forward recomputed: two sin / cos pairs, three
// normalisations) and the part that does: a caller whose gradients arrive late (k_mesh_views_backward_finalize: they are sums
// over the strips' partials) evaluates the first while it waits for them.  look_at_bwd below is the two in a row.
template <typename T> struct LookAtFwd { LookAtState<T> st; T se, ce, sa, ca; };
template <typename T> MVR_HD LookAtFwd<T> look_at_fwd_state(T azim, T elev, T dist) {
  LookAtFwd<T> f;
  sincos_deg(elev, &f.se, &f.ce);
  sincos_deg(azim, &f.sa, &f.ca);
  T R[9], Tr[3];
  look_at_from_position(camera_position_sc(f.se, f.ce, f.sa, f.ca, dist), R, Tr, &f.st);
  return f;
}
template <typename T> MVR_HD void look_at_bwd_from_state(const LookAtFwd<T> &fw, T dist, const T gR[9], const T gT[3], T *g_azim, T *g_elev, T *g_dist);
template <typename T> MVR_HD void look_at_bwd(T azim, T elev, T dist, const T gR[9], const T gT[3], T *g_azim, T *g_elev, T *g_dist) {
  look_at_bwd_from_state(look_at_fwd_state(azim, elev, dist), dist, gR, gT, g_azim, g_elev, g_dist);
}
template <typename T> MVR_HD void look_at_bwd_from_state(const LookAtFwd<T> &fw, T dist, const T gR[9], const T gT[3], T *g_azim, T *g_elev, T *g_dist) {
  const LookAtState<T> &st = fw.st;
  V3<T> gx = {gR[0], gR[3], gR[6]}, gy = {gR[1], gR[4], gR[7]}, gz = {gR[2], gR[5], gR[8]};
  // T_j = -(axis_j . C)
  V3<T> gC = {-((gT[0] * st.x.x + gT[1] * st.y.x) + gT[2] * st.z.x),
              -((gT[0] * st.x.y + gT[1] * st.y.y) + gT[2] * st.z.y),
              -((gT[0] * st.x.z + gT[1] * st.y.z) + gT[2] * st.z.z)};
  gx = {gx.x - gT[0] * st.C.x, gx.y - gT[0] * st.C.y, gx.z - gT[0] * st.C.z};
  gy = {gy.x - gT[1] * st.C.x, gy.y - gT[1] * st.C.y, gy.z - gT[1] * st.C.z};
  gz = {gz.x - gT[2] * st.C.x, gz.y - gT[2] * st.C.y, gz.z - gT[2] * st.C.z};
  V3<T> up = {(T)0, (T)1, (T)0};
  if (st.repaired) {
    // x = normalize(cross(y, z)) replaced the first x (which still fed y)
    V3<T> gtx = normalize_bwd(st.x, st.dx, st.cx, gx);
    // c = y x z : g_y += z x gtx ; g_z += gtx x y
    V3<T> a = cross(st.z, gtx), b = cross(gtx, st.y);
    gy = {gy.x + a.x, gy.y + a.y, gy.z + a.z};
    gz = {gz.x + b.x, gz.y + b.y, gz.z + b.z};
    // recover the first x (what y was built from): x0 = normalize(up x z)
    V3<T> tx0 = cross(up, st.z);
    T d0, n0;
    V3<T> x0 = normalize(tx0, (T)kLookAtEps, &d0, &n0);
    V3<T> gty = normalize_bwd(st.y, st.dy, st.cy, gy);
    V3<T> gz2 = cross(x0, gty), gx0 = cross(gty, st.z);
    gz = {gz.x + gz2.x, gz.y + gz2.y, gz.z + gz2.z};
    V3<T> gtx0 = normalize_bwd(x0, d0, !(n0 > (T)kLookAtEps), gx0);
    V3<T> gz3 = cross(gtx0, up);
    gz = {gz.x + gz3.x, gz.y + gz3.y, gz.z + gz3.z};
  } else {
    // y = normalize(z x x)
    V3<T> gty = normalize_bwd(st.y, st.dy, st.cy, gy);
    V3<T> gz2 = cross(st.x, gty), gx2 = cross(gty, st.z);
    gz = {gz.x + gz2.x, gz.y + gz2.y, gz.z + gz2.z};
    gx = {gx.x + gx2.x, gx.y + gx2.y, gx.z + gx2.z};
    // x = normalize(up x z)
    V3<T> gtx = normalize_bwd(st.x, st.dx, st.cx, gx);
    V3<T> gz3 = cross(gtx, up);
    gz = {gz.x + gz3.x, gz.y + gz3.y, gz.z + gz3.z};
  }
  // z = normalize(-C)
  V3<T> gneg = normalize_bwd(st.z, st.dz, st.cz, gz);
  gC = {gC.x - gneg.x, gC.y - gneg.y, gC.z - gneg.z};
  // camera_position_bwd with the forward's sines and cosines (the same values: sincos_deg of the same angles)
  const T se = fw.se, ce = fw.ce, sa = fw.sa, ca = fw.ca, k = (T)(kPi / 180.0);
  *g_dist = (gC.x * (ce * sa) + gC.y * se) + gC.z * (ce * ca);
  *g_elev = k * (dist * ((gC.x * (-se * sa) + gC.y * ce) + gC.z * (-se * ca)));
  *g_azim = k * (dist * (gC.x * (ce * ca) - gC.z * (ce * sa)));
}

// Camera centre the shader uses: cameras.get_camera_center() = -T R^T.
template <typename T> MVR_HD V3<T> camera_center(const T R[9], const T Tr[3]) {
  return {-((Tr[0] * R[0] + Tr[1] * R[1]) + Tr[2] * R[2]),
          -((Tr[0] * R[3] + Tr[1] * R[4]) + Tr[2] * R[5]),
          -((Tr[0] * R[6] + Tr[1] * R[7]) + Tr[2] * R[8])};
}

// X_view = X_world R + T
template <typename T> MVR_HD V3<T> world_to_view(const T R[9], const T Tr[3], const V3<T> &X) {
  return {((X.x * R[0] + X.y * R[3]) + X.z * R[6]) + Tr[0],
          ((X.x * R[1] + X.y * R[4]) + X.z * R[7]) + Tr[1],
          ((X.x * R[2] + X.y * R[5]) + X.z * R[8]) + Tr[2]};
}

// FoVPerspectiveCameras (znear=1, aspect=1): s = 1/tan(fov/2)
MVR_HD float fov_scale(float fov_deg) { return (float)(1.0 / tan(((double)fov_deg * (kPi / 180.0)) / 2.0)); }
MVR_HD double fov_scale(double fov_deg) { return 1.0 / tan((fov_deg * (kPi / 180.0)) / 2.0); }

// MVR_PROJECTION_FMA (round 6): the same sum the way a batched 4 x 4 matrix product accumulates it on a GPU — ascending k, every
// product after the first fused into the running sum; PyTorch3D projects vertices through Transform3d.transform_points =
// torch.bmm([X 1], [[R, 0], [T, 1]]) (the k = 3 term is 1 * T: fma(1, T, acc) = acc + T; the homogeneous coordinate is exactly 1).
// The two forms differ in the last bit of a view coordinate; that decides which of two COINCIDENT faces (a shell and its reversed
// copy) has the smaller interpolated depth, nothing else (oracle switch #5, tests/test_oracle.py).
MVR_HD float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
MVR_HD double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
template <typename T> MVR_HD V3<T> world_to_view_fma(const T R[9], const T Tr[3], const V3<T> &X) {
  return {fma_t(X.z, R[6], fma_t(X.y, R[3], X.x * R[0])) + Tr[0],
          fma_t(X.z, R[7], fma_t(X.y, R[4], X.x * R[1])) + Tr[1],
          fma_t(X.z, R[8], fma_t(X.y, R[5], X.x * R[2])) + Tr[2]};
}

// NDC x, y and VIEW z of a world vertex (MeshRasterizer.transform).  ([x y z 1] K has one nonzero product per component — x s,
// y s, z —, so a fused chain rounds it as this closed form does.)
template <typename T> MVR_HD V3<T> project_vertex(const T R[9], const T Tr[3], T s, const V3<T> &X, bool fma_sum = false) {
  V3<T> v = fma_sum ? world_to_view_fma(R, Tr, X) : world_to_view(R, Tr, X);
  return {(s * v.x) / v.z, (s * v.y) / v.z, v.z};
}

// PixToNdc(i, S) = -1 + (2 i + 1)/S ; output column j <-> i = S-1-j (and rows likewise).
template <typename T> MVR_HD T pix_to_ndc(int i, int S) { return (T)-1 + ((T)2 * (T)i + (T)1) / (T)S; }

// ---- rasteriser: one (pixel, face) test ------------------------------------------------

template <typename T> MVR_HD T edge_fn(T px, T py, T ax, T ay, T bx, T by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

template <typename T> struct FaceSetup {
  V3<T> v0, v1, v2;            // NDC x, y ; view z
  T xmin, xmax, ymin, ymax;
  T area;                      // edge(v2; v0, v1) + kEps  (barycentric denominator)
  T inv_area;                  // RN(1 / area): shared by the three barycentric divisions (div_by)
  bool drawable;               // passes the per-face culls
};

template <typename T> MVR_HD FaceSetup<T> face_setup(const V3<T> &v0, const V3<T> &v1, const V3<T> &v2, bool cull_backfaces) {
  FaceSetup<T> fs;
  fs.v0 = v0; fs.v1 = v1; fs.v2 = v2;
  fs.xmin = tmin(tmin(v0.x, v1.x), v2.x); fs.xmax = tmax(tmax(v0.x, v1.x), v2.x);
  fs.ymin = tmin(tmin(v0.y, v1.y), v2.y); fs.ymax = tmax(tmax(v0.y, v1.y), v2.y);
  T zmin = tmin(tmin(v0.z, v1.z), v2.z), zmax = tmax(tmax(v0.z, v1.z), v2.z);
  T face_area = edge_fn(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
  bool back_face = face_area < (T)0;
  bool zero_area = (face_area <= (T)kEps && face_area >= -(T)kEps);
  bool z_invalid = zmin < (T)kEps;
  fs.drawable = !(zmax < (T)0 || (cull_backfaces && back_face) || z_invalid || zero_area);
  fs.area = edge_fn(v2.x, v2.y, v0.x, v0.y, v1.x, v1.y) + (T)kEps;
  fs.inv_area = (T)1 / fs.area;
  return fs;
}

// Barycentrics (optionally perspective-corrected) and interpolated z of pixel centre (px,py) from the
// three edge values; no inside test.  Every quotient equals the IEEE division of the oracle (div_by).
template <typename T> MVR_HD void bary_from_edges(const FaceSetup<T> &fs, T e0, T e1, T e2, bool persp, T *w0, T *w1, T *w2, T *pz) {
  T a0 = div_by(e0, fs.area, fs.inv_area);
  T a1 = div_by(e1, fs.area, fs.inv_area);
  T a2 = div_by(e2, fs.area, fs.inv_area);
  if (persp) {
    T t0 = (a0 * fs.v1.z) * fs.v2.z, t1 = (fs.v0.z * a1) * fs.v2.z, t2 = (fs.v0.z * fs.v1.z) * a2;
    T den = tmax((t0 + t1) + t2, (T)kEps);
    T yd = (T)1 / den;
    a0 = div_by(t0, den, yd); a1 = div_by(t1, den, yd); a2 = div_by(t2, den, yd);
  }
  *w0 = a0; *w1 = a1; *w2 = a2;
  *pz = (a0 * fs.v0.z + a1 * fs.v1.z) + a2 * fs.v2.z;
}

template <typename T> MVR_HD void face_bary(const FaceSetup<T> &fs, T px, T py, bool persp, T *w0, T *w1, T *w2, T *pz) {
  T e0 = edge_fn(px, py, fs.v1.x, fs.v1.y, fs.v2.x, fs.v2.y);
  T e1 = edge_fn(px, py, fs.v2.x, fs.v2.y, fs.v0.x, fs.v0.y);
  T e2 = edge_fn(px, py, fs.v0.x, fs.v0.y, fs.v1.x, fs.v1.y);
  bary_from_edges(fs, e0, e1, e2, persp, w0, w1, w2, pz);
}

// CheckPixelInsideFace with blur_radius = 0: true iff the pixel centre is strictly inside.
template <typename T> MVR_HD bool face_test(const FaceSetup<T> &fs, T px, T py, bool persp, T *pz, T *w0, T *w1, T *w2) {
  if (px > fs.xmax || px < fs.xmin || py > fs.ymax || py < fs.ymin) return false;
  // cheap sign pre-filter: e_i / area > 0 requires e_i != 0 with the sign of area
  T e0 = edge_fn(px, py, fs.v1.x, fs.v1.y, fs.v2.x, fs.v2.y);
  T e1 = edge_fn(px, py, fs.v2.x, fs.v2.y, fs.v0.x, fs.v0.y);
  T e2 = edge_fn(px, py, fs.v0.x, fs.v0.y, fs.v1.x, fs.v1.y);
  bool pos = fs.area > (T)0;
  bool maybe = pos ? (e0 > (T)0 && e1 > (T)0 && e2 > (T)0) : (e0 < (T)0 && e1 < (T)0 && e2 < (T)0);
  if (!maybe) return false;
  bary_from_edges(fs, e0, e1, e2, persp, w0, w1, w2, pz);
  if (*pz < (T)0) return false;
  return *w0 > (T)0 && *w1 > (T)0 && *w2 > (T)0;
}

// ---- face clipping at view depth z = zc (PyTorch3D >= 0.5 clip_faces, SURVEY A.9 ii; perspective-correct rasterisation) ------
// A face with one or two vertices nearer than zc is cut: one vertex behind (p1; p2, p3 the next two in cyclic order) leaves
// the quad (p4, p2, p5) + (p5, p2, p3), two behind (p1 the one in front) leave (p1, p4, p5); p4 on edge p1 p2, p5 on p1 p3 at
// the view-depth parameter w = (z1 - zc) / (z1 - z_other); their NDC x, y are interpolated in x z, y z (linear in view space)
// and divided by the new vertex's own interpolated z.  The rasteriser tests the cut triangles; PyTorch3D maps their barycentrics back to the original face
// (beta).  Same operations in the same order as the oracle's clip_face.  straddles(): 1 or 2 vertices behind.
template <typename T> MVR_HD bool straddles(T z0, T z1, T z2, T zc) {
  const int nb = (z0 < zc ? 1 : 0) + (z1 < zc ? 1 : 0) + (z2 < zc ? 1 : 0);
  return nb == 1 || nb == 2;
}
// (No dynamically indexed arrays: private arrays that the compiler cannot keep in registers would give every kernel that
// contains this rare path a scratch allocation.  The barycentric map `beta` of the reference is not needed here: colours
// and gradients of cut faces use bary_ray below.)
template <typename T> struct ClippedFace { int nt; V3<T> t0[3], t1[3]; };   // nt cut triangles: t0 (and t1)
template <typename T> MVR_HD ClippedFace<T> clip_face(const V3<T> &v0, const V3<T> &v1, const V3<T> &v2, T zc) {
  ClippedFace<T> o;
  const bool h0 = v0.z < zc, h1 = v1.z < zc, h2 = v2.z < zc;   // behind the plane
  const int nb = (h0 ? 1 : 0) + (h1 ? 1 : 0) + (h2 ? 1 : 0);
  o.t0[0] = v0; o.t0[1] = v1; o.t0[2] = v2;
  o.t1[0] = v0; o.t1[1] = v1; o.t1[2] = v2;
  o.nt = nb == 3 ? 0 : 1;
  if (nb == 0 || nb == 3) return o;
  // i1: the vertex that is alone on its side of the plane (the LAST such index, as the reference's loop leaves it);
  // p1, p2, p3: the vertices from there in cyclic order
  const bool alone = nb == 1;
  const int i1 = (h2 == alone) ? 2 : ((h1 == alone) ? 1 : 0);
  const V3<T> p1 = i1 == 0 ? v0 : (i1 == 1 ? v1 : v2), p2 = i1 == 0 ? v1 : (i1 == 1 ? v2 : v0), p3 = i1 == 0 ? v2 : (i1 == 1 ? v0 : v1);
  const T w2 = (p1.z - zc) / (p1.z - p2.z), w3 = (p1.z - zc) / (p1.z - p3.z);
  V3<T> p4, p5;
  // (upstream projects a new vertex back by ITS OWN interpolated depth, which equals zc only up to rounding)
  p4.z = p1.z * ((T)1 - w2) + p2.z * w2;
  p5.z = p1.z * ((T)1 - w3) + p3.z * w3;
  p4.x = ((p1.x * p1.z) * ((T)1 - w2) + (p2.x * p2.z) * w2) / p4.z;
  p5.x = ((p1.x * p1.z) * ((T)1 - w3) + (p3.x * p3.z) * w3) / p5.z;
  p4.y = ((p1.y * p1.z) * ((T)1 - w2) + (p2.y * p2.z) * w2) / p4.z;
  p5.y = ((p1.y * p1.z) * ((T)1 - w3) + (p3.y * p3.z) * w3) / p5.z;
  if (nb == 1) {   // p1 behind: (p4, p2, p5) and (p5, p2, p3)
    o.t0[0] = p4; o.t0[1] = p2; o.t0[2] = p5;
    o.t1[0] = p5; o.t1[1] = p2; o.t1[2] = p3;
    o.nt = 2;
  } else {         // p2, p3 behind: (p1, p4, p5)
    o.t0[0] = p1; o.t0[1] = p4; o.t0[2] = p5;
    o.nt = 1;
  }
  return o;
}

// Barycentrics of the point where the ray of pixel (px, py) meets the plane of a triangle given in VIEW space (camera at the
// origin, ray direction d = (px / s, py / s, 1)): b_i = c_i / sum c, c_0 = d . (X1 x X2), c_1 = d . (X2 x X0), c_2 = d . (X0 x X1).
// In exact arithmetic these are the perspective-correct barycentrics of the rasteriser, and the ones PyTorch3D's clipped
// faces map back to — without dividing by a vertex depth, so they stay finite for vertices at or behind the camera plane.
// Shading and gradients of pixels on CUT faces use them (colours and gradients only: coverage and depth come from the
// cut triangles, bit for bit as in the oracle).
template <typename T> MVR_HD void bary_ray(const V3<T> &d, const V3<T> &X0, const V3<T> &X1, const V3<T> &X2, T *b0, T *b1, T *b2, T *inv_sum = nullptr) {
  const T c0 = dot(d, cross(X1, X2)), c1 = dot(d, cross(X2, X0)), c2 = dot(d, cross(X0, X1));
  const T r = (T)1 / ((c0 + c1) + c2);
  *b0 = c0 * r; *b1 = c1 * r; *b2 = c2 * r;
  if (inv_sum) *inv_sum = r;
}
// gradient of bary_ray w.r.t. the three view-space vertices, given g_i = dL/db_i
template <typename T> MVR_HD void bary_ray_bwd(const V3<T> &d, const V3<T> &X0, const V3<T> &X1, const V3<T> &X2, T b0, T b1, T b2, T inv_sum,
                                               T g0, T g1, T g2, V3<T> *gX0, V3<T> *gX1, V3<T> *gX2) {
  const T gs = (g0 * b0 + g1 * b1) + g2 * b2;
  const T h0 = (g0 - gs) * inv_sum, h1 = (g1 - gs) * inv_sum, h2 = (g2 - gs) * inv_sum;   // dL/dc_i
  // c0 = d.(X1 x X2) = X1.(X2 x d) = X2.(d x X1); c1 = d.(X2 x X0) = X2.(X0 x d) = X0.(d x X2); c2 = d.(X0 x X1) = X0.(X1 x d) = X1.(d x X0)
  const V3<T> dX0 = cross(d, X0), dX1 = cross(d, X1), dX2 = cross(d, X2);
  *gX0 = {h1 * dX2.x - h2 * dX1.x, h1 * dX2.y - h2 * dX1.y, h1 * dX2.z - h2 * dX1.z};
  *gX1 = {h2 * dX0.x - h0 * dX2.x, h2 * dX0.y - h0 * dX2.y, h2 * dX0.z - h0 * dX2.z};
  *gX2 = {h0 * dX1.x - h1 * dX0.x, h0 * dX1.y - h1 * dX0.y, h0 * dX1.z - h1 * dX0.z};
}

// ---- shading: HardPhongShader + DirectionalLights (models/renderer.py:102-103,190-191) ----

template <typename T> struct Phong { T ambient, diffuse, specular, shininess; };

// a^shininess for a in [0, 1].  The reference's value (64, models/renderer.py uses the Materials default) is six
// squarings.  Any other exponent: on the device exp2(s log2 a) with the hardware log/exp (relative error ~ s * 1e-7,
// far inside the 1e-4 colour tolerance; a double-precision pow would drag ~200 f64 instructions and their registers
// into the shading kernels for a path the reference never takes), on the host the libm pow.
template <typename T> MVR_HD T pow_shininess(T a, T shininess) {
  if (shininess == (T)64) { T a2 = a * a, a4 = a2 * a2, a8 = a4 * a4, a16 = a8 * a8, a32 = a16 * a16; return a32 * a32; }
#if defined(__HIP_DEVICE_COMPILE__)
  if (!(a > (T)0)) return shininess == (T)0 ? (T)1 : (T)0;
  return (T)__builtin_amdgcn_exp2f((float)shininess * __builtin_amdgcn_logf((float)a));
#else
  return (T)pow((double)a, (double)shininess);
#endif
}

// colour of one covered pixel.  X*, N*: world position / vertex normal of the face's
// vertices; c*: their colours; l: light direction (towards the light); cc: camera centre.
template <typename T> MVR_HD V3<T> phong_pixel(T w0, T w1, T w2, const V3<T> &X0, const V3<T> &X1, const V3<T> &X2,
                                              const V3<T> &N0, const V3<T> &N1, const V3<T> &N2,
                                              const V3<T> &c0, const V3<T> &c1, const V3<T> &c2,
                                              const V3<T> &l, const V3<T> &cc, const Phong<T> &ph) {
  V3<T> p = {(w0 * X0.x + w1 * X1.x) + w2 * X2.x, (w0 * X0.y + w1 * X1.y) + w2 * X2.y, (w0 * X0.z + w1 * X1.z) + w2 * X2.z};
  V3<T> n = {(w0 * N0.x + w1 * N1.x) + w2 * N2.x, (w0 * N0.y + w1 * N1.y) + w2 * N2.y, (w0 * N0.z + w1 * N1.z) + w2 * N2.z};
  V3<T> tex = {(w0 * c0.x + w1 * c1.x) + w2 * c2.x, (w0 * c0.y + w1 * c1.y) + w2 * c2.y, (w0 * c0.z + w1 * c1.z) + w2 * c2.z};
  V3<T> nh = normalize_shared(n, (T)kNormalizeEps);
  V3<T> lh = normalize_shared(l, (T)kNormalizeEps);
  T cosang = dot(nh, lh);
  T diff = cosang > (T)0 ? cosang : (T)0;
  T mask = cosang > (T)0 ? (T)1 : (T)0;
  V3<T> view = {cc.x - p.x, cc.y - p.y, cc.z - p.z};
  V3<T> vh = normalize_shared(view, (T)kNormalizeEps);
  V3<T> refl = {-lh.x + (T)2 * (cosang * nh.x), -lh.y + (T)2 * (cosang * nh.y), -lh.z + (T)2 * (cosang * nh.z)};
  T dvr = dot(vh, refl);
  T alpha = (dvr > (T)0 ? dvr : (T)0) * mask;
  T spec = ph.specular * pow_shininess(alpha, ph.shininess);
  T sc = ph.ambient + ph.diffuse * diff;
  return {sc * tex.x + spec, sc * tex.y + spec, sc * tex.z + spec};
}

// Per-pixel backward.  Inputs: the face's projected verts (fs), world attributes, upstream
// gradient G (rgb).  Accumulates into acc[18]: gR[9], gT[3], gCc[3] (camera centre), gL[3].
// pix_to_face is constant; relu / indicator kinks use subgradient 0.
template <typename T> struct PixelGradIn {
  T R[9], Tr[3], s;     // camera of the view, projection scale
  V3<T> l, cc;          // light direction, camera centre
  Phong<T> ph;
  bool persp;
  bool l_is_unit = false;   // `l` is already normalised; acc[15..17] then receives the gradient w.r.t. the UNIT vector and
                            // the caller applies normalize_rs_bwd once per view (it is linear in that gradient)
};

// What the per-pixel backward needs of a face that does not depend on the pixel: the projected vertices, the
// reciprocals of their depths and of the (eps-shifted) area.  k_mesh_backward_tab computes it once per face and strip.
template <typename T> struct FaceProjected { T u0, v0, z0, rz0, u1, v1, z1, rz1, u2, v2, z2, rz2, ra; };
template <typename T> MVR_HD FaceProjected<T> face_project_for_bwd(const PixelGradIn<T> &cam, const V3<T> &X0, const V3<T> &X1, const V3<T> &X2) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  FaceProjected<T> f;
  const V3<T> Xv0 = world_to_view(cam.R, cam.Tr, X0), Xv1 = world_to_view(cam.R, cam.Tr, X1), Xv2 = world_to_view(cam.R, cam.Tr, X2);
  f.rz0 = trcp(Xv0.z); f.rz1 = trcp(Xv1.z); f.rz2 = trcp(Xv2.z);
  f.u0 = (cam.s * Xv0.x) * f.rz0; f.v0 = (cam.s * Xv0.y) * f.rz0; f.z0 = Xv0.z;
  f.u1 = (cam.s * Xv1.x) * f.rz1; f.v1 = (cam.s * Xv1.y) * f.rz1; f.z1 = Xv1.z;
  f.u2 = (cam.s * Xv2.x) * f.rz2; f.v2 = (cam.s * Xv2.y) * f.rz2; f.z2 = Xv2.z;
  const T area = edge_fn(f.u2, f.v2, f.u0, f.v0, f.u1, f.v1) + (T)kEps;
  f.ra = trcp(area);
  return f;
}

// The shading half of the per-pixel backward: barycentrics w -> (p, n, tex) -> Phong -> upstream gradient G back to
// g_p, g_n, g_tex (gradients of the interpolated position / normal / texel), the camera centre and the light.
// FLAT: the three vertex colours are the same (a constant object colour): the texel is that colour for any barycentrics that
// sum to one — what the fast rasteriser's shading assumes too — so it is not interpolated and carries no gradient.
template <typename T, bool FLAT = false> MVR_HD void phong_shade_bwd(const PixelGradIn<T> &cam, T w0, T w1, T w2,
                                                  const V3<T> &X0, const V3<T> &X1, const V3<T> &X2,
                                                  const V3<T> &N0, const V3<T> &N1, const V3<T> &N2,
                                                  const V3<T> &c0, const V3<T> &c1, const V3<T> &c2, const V3<T> &G,
                                                  V3<T> *g_p_out, V3<T> *g_n_out, V3<T> *g_tex_out, V3<T> *g_cc_out, V3<T> *g_l_out) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  V3<T> p = {(w0 * X0.x + w1 * X1.x) + w2 * X2.x, (w0 * X0.y + w1 * X1.y) + w2 * X2.y, (w0 * X0.z + w1 * X1.z) + w2 * X2.z};
  V3<T> n = {(w0 * N0.x + w1 * N1.x) + w2 * N2.x, (w0 * N0.y + w1 * N1.y) + w2 * N2.y, (w0 * N0.z + w1 * N1.z) + w2 * N2.z};
  V3<T> tex = c0;
  if (!FLAT) tex = {(w0 * c0.x + w1 * c1.x) + w2 * c2.x, (w0 * c0.y + w1 * c1.y) + w2 * c2.y, (w0 * c0.z + w1 * c1.z) + w2 * c2.z};
  T idn, idl, idv;
  bool cn, cl, cv;
  V3<T> nh = normalize_rs(n, (T)kNormalizeEps, &idn, &cn);
  V3<T> lh = cam.l;
  idl = (T)1; cl = true;
  if (!cam.l_is_unit) lh = normalize_rs(cam.l, (T)kNormalizeEps, &idl, &cl);
  T cosang = dot_c(nh, lh);
  bool lit = cosang > (T)0;
  V3<T> view = {cam.cc.x - p.x, cam.cc.y - p.y, cam.cc.z - p.z};
  V3<T> vh = normalize_rs(view, (T)kNormalizeEps, &idv, &cv);
  V3<T> refl = {-lh.x + (T)2 * (cosang * nh.x), -lh.y + (T)2 * (cosang * nh.y), -lh.z + (T)2 * (cosang * nh.z)};
  T dvr = dot_c(vh, refl);
  bool spec_on = lit && dvr > (T)0;
  T alpha = spec_on ? dvr : (T)0;
  T sc = cam.ph.ambient + cam.ph.diffuse * (lit ? cosang : (T)0);
  // ---- backward ------------------------------------------------------------
  V3<T> g_tex = {G.x * sc, G.y * sc, G.z * sc};
  T g_sc = (G.x * tex.x + G.y * tex.y) + G.z * tex.z;
  T g_cos = lit ? cam.ph.diffuse * g_sc : (T)0;
  V3<T> g_nh = {(T)0, (T)0, (T)0}, g_lh = {(T)0, (T)0, (T)0}, g_p = {(T)0, (T)0, (T)0};
  V3<T> g_cc = {(T)0, (T)0, (T)0};
  if (spec_on) {
    T g_spec = (G.x + G.y) + G.z;
    // d(alpha^s)/d alpha = s * alpha^(s-1)
    T dpow;
    if (cam.ph.shininess == (T)64) {
      T a2_ = alpha * alpha, a4 = a2_ * a2_, a8 = a4 * a4, a16 = a8 * a8, a32 = a16 * a16;
      dpow = (T)64 * (((a32 * a16) * (a8 * a4)) * (a2_ * alpha));
    } else {
      dpow = cam.ph.shininess * pow_shininess(alpha, cam.ph.shininess - (T)1);
    }
    T g_dvr = cam.ph.specular * dpow * g_spec;
    V3<T> g_vh = {g_dvr * refl.x, g_dvr * refl.y, g_dvr * refl.z};
    V3<T> g_r = {g_dvr * vh.x, g_dvr * vh.y, g_dvr * vh.z};
    g_lh = {-g_r.x, -g_r.y, -g_r.z};
    g_cos += (T)2 * dot_c(g_r, nh);
    g_nh = {(T)2 * cosang * g_r.x, (T)2 * cosang * g_r.y, (T)2 * cosang * g_r.z};
    V3<T> g_view = normalize_rs_bwd(vh, idv, cv, g_vh);
    g_cc = g_view;
    g_p = {-g_view.x, -g_view.y, -g_view.z};
  }
  g_nh = {g_nh.x + g_cos * lh.x, g_nh.y + g_cos * lh.y, g_nh.z + g_cos * lh.z};
  g_lh = {g_lh.x + g_cos * nh.x, g_lh.y + g_cos * nh.y, g_lh.z + g_cos * nh.z};
  V3<T> g_n = normalize_rs_bwd(nh, idn, cn, g_nh);
  V3<T> g_l = cam.l_is_unit ? g_lh : normalize_rs_bwd(lh, idl, cl, g_lh);
  *g_p_out = g_p; *g_n_out = g_n; *g_tex_out = g_tex; *g_cc_out = g_cc; *g_l_out = g_l;
}

// `mid` runs between the shading part and the geometric part.  The face's data (fp, X, N) are not used across it: a caller
// that holds them in fast memory may let `mid` re-read them there (they are passed by reference and read again after it)
// instead of keeping ~40 registers alive through the shading arithmetic (k_mesh_backward_tab).
struct NoMid { MVR_HD void operator()() const {} };
template <typename T, typename Mid = NoMid> MVR_HD void phong_pixel_bwd_pre(const PixelGradIn<T> &cam, T px, T py, const FaceProjected<T> &fp,
                                                      const V3<T> &X0, const V3<T> &X1, const V3<T> &X2,
                                                      const V3<T> &N0, const V3<T> &N1, const V3<T> &N2,
                                                      const V3<T> &c0, const V3<T> &c1, const V3<T> &c2,
                                                      const V3<T> &G, T acc[18], Mid mid = Mid()) {
  // Gradient arithmetic: no coverage decision depends on it, so quotients go through shared reciprocals
  // (trcp / trsq) and the compiler may contract multiply-adds.
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  // ---- forward recompute -------------------------------------------------
  T rz0 = fp.rz0, rz1 = fp.rz1, rz2 = fp.rz2;
  T u0 = fp.u0, v0 = fp.v0, z0 = fp.z0, u1 = fp.u1, v1 = fp.v1, z1 = fp.z1, u2 = fp.u2, v2 = fp.v2, z2 = fp.z2;
  T ra = fp.ra;
  T e0 = edge_fn(px, py, u1, v1, u2, v2), e1 = edge_fn(px, py, u2, v2, u0, v0), e2 = edge_fn(px, py, u0, v0, u1, v1);
  T a0 = e0 * ra, a1 = e1 * ra, a2 = e2 * ra;
  T w0 = a0, w1 = a1, w2 = a2, rden = (T)1;
  bool den_clamped = false;
  if (cam.persp) {
    T t0 = (a0 * z1) * z2, t1 = (z0 * a1) * z2, t2 = (z0 * z1) * a2;
    T sum = (t0 + t1) + t2;
    rden = trcp(tmax(sum, (T)kEps));
    den_clamped = !(sum > (T)kEps);
    w0 = t0 * rden; w1 = t1 * rden; w2 = t2 * rden;
  }
  V3<T> g_p, g_n, g_tex, g_cc, g_l;
  phong_shade_bwd(cam, w0, w1, w2, X0, X1, X2, N0, N1, N2, c0, c1, c2, G, &g_p, &g_n, &g_tex, &g_cc, &g_l);
  mid();
  rz0 = fp.rz0; rz1 = fp.rz1; rz2 = fp.rz2;
  u0 = fp.u0; v0 = fp.v0; z0 = fp.z0; u1 = fp.u1; v1 = fp.v1; z1 = fp.z1; u2 = fp.u2; v2 = fp.v2; z2 = fp.z2;
  ra = fp.ra;
  // barycentric gradients
  T gw0 = (dot_c(g_p, X0) + dot_c(g_n, N0)) + dot_c(g_tex, c0);
  T gw1 = (dot_c(g_p, X1) + dot_c(g_n, N1)) + dot_c(g_tex, c1);
  T gw2 = (dot_c(g_p, X2) + dot_c(g_n, N2)) + dot_c(g_tex, c2);
  T ga0 = gw0, ga1 = gw1, ga2 = gw2, gz0 = (T)0, gz1 = (T)0, gz2 = (T)0;
  if (cam.persp) {
    T gt0, gt1, gt2;
    if (den_clamped) {
      gt0 = gw0 * rden; gt1 = gw1 * rden; gt2 = gw2 * rden;
    } else {
      T s_ = (gw0 * w0 + gw1 * w1) + gw2 * w2;
      gt0 = (gw0 - s_) * rden; gt1 = (gw1 - s_) * rden; gt2 = (gw2 - s_) * rden;
    }
    ga0 = gt0 * (z1 * z2); gz1 += gt0 * (a0 * z2); gz2 += gt0 * (a0 * z1);
    ga1 = gt1 * (z0 * z2); gz0 += gt1 * (a1 * z2); gz2 += gt1 * (z0 * a1);
    ga2 = gt2 * (z0 * z1); gz0 += gt2 * (z1 * a2); gz1 += gt2 * (z0 * a2);
  }
  T ge0 = ga0 * ra, ge1 = ga1 * ra, ge2 = ga2 * ra;
  T g_area = -((ga0 * a0 + ga1 * a1) + ga2 * a2) * ra;
  // edge(p; a, b): d/dax = py - by ; d/day = bx - px ; d/dbx = ay - py ; d/dby = px - ax
  T gu0 = (T)0, gv0 = (T)0, gu1 = (T)0, gv1 = (T)0, gu2 = (T)0, gv2 = (T)0;
  // e0 = edge(p; V1, V2)
  gu1 += ge0 * (py - v2); gv1 += ge0 * (u2 - px); gu2 += ge0 * (v1 - py); gv2 += ge0 * (px - u1);
  // e1 = edge(p; V2, V0)
  gu2 += ge1 * (py - v0); gv2 += ge1 * (u0 - px); gu0 += ge1 * (v2 - py); gv0 += ge1 * (px - u2);
  // e2 = edge(p; V0, V1)
  gu0 += ge2 * (py - v1); gv0 += ge2 * (u1 - px); gu1 += ge2 * (v0 - py); gv1 += ge2 * (px - u0);
  // area = edge(V2; V0, V1): q = V2 -> d/dqx = (v1 - v0), d/dqy = -(u1 - u0); a = V0, b = V1 with p := V2
  gu2 += g_area * (v1 - v0); gv2 += g_area * (-(u1 - u0));
  gu0 += g_area * (v2 - v1); gv0 += g_area * (u1 - u2);
  gu1 += g_area * (v0 - v2); gv1 += g_area * (u2 - u0);
  // projection u = s x / z, v = s y / z, zv = z
  auto proj_bwd = [&](T rz, T u, T v, T gu, T gv, T gz, const V3<T> &Xw) {
    T gx = gu * cam.s * rz, gy = gv * cam.s * rz;
    T gzz = gz - (gu * u + gv * v) * rz;
    acc[0] += Xw.x * gx; acc[1] += Xw.x * gy; acc[2] += Xw.x * gzz;
    acc[3] += Xw.y * gx; acc[4] += Xw.y * gy; acc[5] += Xw.y * gzz;
    acc[6] += Xw.z * gx; acc[7] += Xw.z * gy; acc[8] += Xw.z * gzz;
    acc[9] += gx; acc[10] += gy; acc[11] += gzz;
  };
  proj_bwd(rz0, u0, v0, gu0, gv0, gz0, X0);
  proj_bwd(rz1, u1, v1, gu1, gv1, gz1, X1);
  proj_bwd(rz2, u2, v2, gu2, gv2, gz2, X2);
  acc[12] += g_cc.x; acc[13] += g_cc.y; acc[14] += g_cc.z;
  acc[15] += g_l.x; acc[16] += g_l.y; acc[17] += g_l.z;
}

// The perspective-correct case in RAY form.  The perspective-correct barycentrics of pixel (px, py) are the barycentrics
// of the point where its ray d = (px / s, py / s, 1) meets the face's plane in VIEW space (bary_ray above):
//     w_i = k_i / (k_0 + k_1 + k_2),  k_0 = d . (v1 x v2), k_1 = d . (v2 x v0), k_2 = d . (v0 x v1)
// (evaluated as 2 x 2 determinants of ray-relative vectors, below), and their gradient goes to the view-space vertices directly:
//     h_i = (dL/dw_i - sum_j dL/dw_j w_j) / sum k ;  dL/dv0 = d x (h1 v2 - h2 v1), dL/dv1 = d x (h2 v0 - h0 v2), dL/dv2 = d x (h0 v1 - h1 v0)
// — no projected vertices, no edge functions, no division by a vertex depth (finite for vertices at or behind the camera
// plane): ~55 operations per pixel where the projected form (phong_pixel_bwd_pre) spends ~125 on the same two steps.  The
// kEps clamps of the forward's quotients are not reproduced (they act on faces the rasteriser culls); a ray parallel to
// the face's plane (sum k = 0) contributes nothing.
template <typename T> struct FaceView { V3<T> v0, v1, v2; };   // view-space vertices (rounds 3-5 also carried their cross products: the barycentrics' old form)
template <typename T> MVR_HD FaceView<T> face_view_for_bwd(const PixelGradIn<T> &cam, const V3<T> &X0, const V3<T> &X1, const V3<T> &X2) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  FaceView<T> f;
  f.v0 = world_to_view(cam.R, cam.Tr, X0); f.v1 = world_to_view(cam.R, cam.Tr, X1); f.v2 = world_to_view(cam.R, cam.Tr, X2);
  return f;
}
// (`mid` as in phong_pixel_bwd_pre: a scheduling hook between the shading half and the geometric half)
// FLAT (constant object colour): a term common to the three dL/dw_i cancels in h_i when the barycentrics sum to one, and the
// texel's contribution dot(g_tex, c) is such a term — it is left out, with the texel's interpolation (phong_shade_bwd).
template <typename T, typename Mid = NoMid, bool FLAT = false> MVR_HD void phong_pixel_bwd_ray(const PixelGradIn<T> &cam, T px, T py, const FaceView<T> &fv,
                                                      const V3<T> &X0, const V3<T> &X1, const V3<T> &X2,
                                                      const V3<T> &N0, const V3<T> &N1, const V3<T> &N2,
                                                      const V3<T> &c0, const V3<T> &c1, const V3<T> &c2,
                                                      const V3<T> &G, T acc[18], Mid mid = Mid()) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  const T inv_s = trcp(cam.s);
  const T dx = px * inv_s, dy = py * inv_s;
  // k_i = d . (v_j x v_k) = det[d, v_j, v_k] as the 2 x 2 determinant of the RAY-RELATIVE vectors a_i = v_i - z_i d (adding multiples of
  // d to the other columns leaves the determinant alone; a_i.z = 0): the differences come first, as in the reference's edge functions,
  // instead of three products of absolute view-space coordinates that cancel against each other when the face is nearly edge-on
  const T a0x = fv.v0.x - fv.v0.z * dx, a0y = fv.v0.y - fv.v0.z * dy, a1x = fv.v1.x - fv.v1.z * dx, a1y = fv.v1.y - fv.v1.z * dy,
          a2x = fv.v2.x - fv.v2.z * dx, a2y = fv.v2.y - fv.v2.z * dy;
  const T k0 = a1x * a2y - a1y * a2x, k1 = a2x * a0y - a2y * a0x, k2 = a0x * a1y - a0y * a1x;
  const T sum = (k0 + k1) + k2;
  const T r = sum != (T)0 ? trcp(sum) : (T)0;
  const T w0 = k0 * r, w1 = k1 * r, w2 = k2 * r;
  V3<T> g_p, g_n, g_tex, g_cc, g_l;
  phong_shade_bwd<T, FLAT>(cam, w0, w1, w2, X0, X1, X2, N0, N1, N2, c0, c1, c2, G, &g_p, &g_n, &g_tex, &g_cc, &g_l);
  mid();
  T gw0 = dot_c(g_p, X0) + dot_c(g_n, N0), gw1 = dot_c(g_p, X1) + dot_c(g_n, N1), gw2 = dot_c(g_p, X2) + dot_c(g_n, N2);
  if (!FLAT) { gw0 += dot_c(g_tex, c0); gw1 += dot_c(g_tex, c1); gw2 += dot_c(g_tex, c2); }
  const T gs = (gw0 * w0 + gw1 * w1) + gw2 * w2;
  const T h0 = (gw0 - gs) * r, h1 = (gw1 - gs) * r, h2 = (gw2 - gs) * r;
  auto add_vert = [&](const T gx, const T gy, const T gz, const V3<T> &Xw) {   // R / T through Xv = Xw R + T
    acc[0] += Xw.x * gx; acc[1] += Xw.x * gy; acc[2] += Xw.x * gz;
    acc[3] += Xw.y * gx; acc[4] += Xw.y * gy; acc[5] += Xw.y * gz;
    acc[6] += Xw.z * gx; acc[7] += Xw.z * gy; acc[8] += Xw.z * gz;
    acc[9] += gx; acc[10] += gy; acc[11] += gz;
  };
  // d L / d v_i = d x Y_i with Y_0 = h1 v2 - h2 v1 (cyclic).  d x v_i = d x a_i (v_i = a_i + z_i d and d x d = 0), and with a_i.z = 0,
  // d = (dx, dy, 1): d x a_i = (-a_iy, a_ix, dx a_iy - dy a_ix) =: c_i — two operations per vertex instead of a cross product of
  // absolute coordinates, and the gradient is h_j c_k - h_k c_j: 24 operations where Y_i and d x Y_i took 36, on the small vectors.
  const T c0z = dx * a0y - dy * a0x, c1z = dx * a1y - dy * a1x, c2z = dx * a2y - dy * a2x;
  add_vert(h2 * a1y - h1 * a2y, h1 * a2x - h2 * a1x, h1 * c2z - h2 * c1z, X0);
  add_vert(h0 * a2y - h2 * a0y, h2 * a0x - h0 * a2x, h2 * c0z - h0 * c2z, X1);
  add_vert(h1 * a0y - h0 * a1y, h0 * a1x - h1 * a0x, h0 * c1z - h1 * c0z, X2);
  acc[12] += g_cc.x; acc[13] += g_cc.y; acc[14] += g_cc.z;
  acc[15] += g_l.x; acc[16] += g_l.y; acc[17] += g_l.z;
}

template <typename T> MVR_HD void phong_pixel_bwd(const PixelGradIn<T> &cam, T px, T py,
                                                  const V3<T> &X0, const V3<T> &X1, const V3<T> &X2,
                                                  const V3<T> &N0, const V3<T> &N1, const V3<T> &N2,
                                                  const V3<T> &c0, const V3<T> &c1, const V3<T> &c2,
                                                  const V3<T> &G, T acc[18]) {
  if (cam.persp) phong_pixel_bwd_ray(cam, px, py, face_view_for_bwd(cam, X0, X1, X2), X0, X1, X2, N0, N1, N2, c0, c1, c2, G, acc);
  else phong_pixel_bwd_pre(cam, px, py, face_project_for_bwd(cam, X0, X1, X2), X0, X1, X2, N0, N1, N2, c0, c1, c2, G, acc);
}


// ---- soft rasterisation (blur_radius > 0, faces_per_pixel = K) ------------------------------
// PyTorch3D 0.7.x csrc/utils/geometry_utils.cuh + csrc/rasterize_meshes (CheckPixelInsideFace), recalled; the CPU
// oracle restates the same operations independently (oracle/mvr_oracle_impl.inc, face_test_soft).

// PointLineDistanceForward: squared distance from p to the SEGMENT a b; *t_out: the clamped parameter (or -1 when the
// segment is degenerate and the distance is taken to b)
template <typename T> MVR_HD T point_segment_dist2(T px, T py, T ax, T ay, T bx, T by, T *t_out = nullptr) {
  const T bax = bx - ax, bay = by - ay;
  const T l2 = bax * bax + bay * bay;
  if (l2 <= (T)kEps) {
    if (t_out) *t_out = (T)-1;
    return (px - bx) * (px - bx) + (py - by) * (py - by);
  }
  T t = (bax * (px - ax) + bay * (py - ay)) / l2;
  t = t < (T)0 ? (T)0 : (t > (T)1 ? (T)1 : t);
  if (t_out) *t_out = t;
  const T qx = ax + t * bax, qy = ay + t * bay;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

// PointTriangleDistanceForward: min over the edges (v0 v1), (v0 v2), (v1 v2); *edge: 0, 1, 2 in that order
template <typename T> MVR_HD T point_triangle_dist2(T px, T py, const V3<T> &v0, const V3<T> &v1, const V3<T> &v2, int *edge = nullptr,
                                                    T *t_out = nullptr) {
  T t01, t02, t12;
  const T e01 = point_segment_dist2(px, py, v0.x, v0.y, v1.x, v1.y, &t01);
  const T e02 = point_segment_dist2(px, py, v0.x, v0.y, v2.x, v2.y, &t02);
  const T e12 = point_segment_dist2(px, py, v1.x, v1.y, v2.x, v2.y, &t12);
  const int k = (e01 <= e02 && e01 <= e12) ? 0 : ((e02 <= e01 && e02 <= e12) ? 1 : 2);
  if (edge) *edge = k;
  if (t_out) *t_out = k == 0 ? t01 : (k == 1 ? t02 : t12);
  return k == 0 ? e01 : (k == 1 ? e02 : e12);
}

template <typename T> struct SoftFrag {
  T z, dist;          // interpolated depth (with the clipped barycentrics), SIGNED squared distance (negative inside)
  T b0, b1, b2;       // barycentrics as stored in the fragments (clipped + renormalised when clip_bary)
  T w0, w1, w2;       // (perspective-corrected) barycentrics before clipping
  bool inside;
};

// soft_frag_eval(..., with_dist = false) with its nine IEEE divisions — three each by the face's area, by the perspective
// denominator and by the clip sum — through div_by and ONE correctly rounded reciprocal per divisor (r_area = RN(1 / area) comes
// from the caller: it is a per-face constant): the same quotients bit for bit (Markstein: RN(q + (a - b q) y) = RN(a / b) for
// y = RN(1 / b), q = RN(a y), absent underflow — numerators here are edge values and their products with vertex depths), 51
// operations instead of 108.  The soft forward's candidate walk is bound by exactly this arithmetic.
template <typename T> MVR_HD SoftFrag<T> soft_frag_eval_shared(const V3<T> &v0, const V3<T> &v1, const V3<T> &v2, T area, T r_area, T px, T py,
                                                               bool persp, bool clip_bary) {
  SoftFrag<T> f;
  T w0 = div_by(edge_fn(px, py, v1.x, v1.y, v2.x, v2.y), area, r_area);
  T w1 = div_by(edge_fn(px, py, v2.x, v2.y, v0.x, v0.y), area, r_area);
  T w2 = div_by(edge_fn(px, py, v0.x, v0.y, v1.x, v1.y), area, r_area);
  if (persp) {
    const T t0 = (w0 * v1.z) * v2.z, t1 = (v0.z * w1) * v2.z, t2 = (v0.z * v1.z) * w2;
    const T den = tmax((t0 + t1) + t2, (T)kEps);
    const T rden = (T)1 / den;
    w0 = div_by(t0, den, rden); w1 = div_by(t1, den, rden); w2 = div_by(t2, den, rden);
  }
  f.w0 = w0; f.w1 = w1; f.w2 = w2;
  T c0 = w0, c1 = w1, c2 = w2;
  if (clip_bary) {   // BarycentricClipForward
    c0 = tmax(tmin(w0, (T)1), (T)0); c1 = tmax(tmin(w1, (T)1), (T)0); c2 = tmax(tmin(w2, (T)1), (T)0);
    const T sm = tmax((c0 + c1) + c2, (T)1e-5);
    const T rsm = (T)1 / sm;
    c0 = div_by(c0, sm, rsm); c1 = div_by(c1, sm, rsm); c2 = div_by(c2, sm, rsm);
  }
  f.b0 = c0; f.b1 = c1; f.b2 = c2;
  f.z = (c0 * v0.z + c1 * v1.z) + c2 * v2.z;
  f.inside = w0 > (T)0 && w1 > (T)0 && w2 > (T)0;
  f.dist = (T)0;
  return f;
}
// The fragment of (pixel, face) without the acceptance tests (its nine quotients through soft_frag_eval_shared: one IEEE division
// per divisor — area, perspective denominator, clip sum — instead of three, the same results bit for bit).
template <typename T> MVR_HD SoftFrag<T> soft_frag_eval(const V3<T> &v0, const V3<T> &v1, const V3<T> &v2, T px, T py, bool persp, bool clip_bary, bool with_dist = true) {
  const T area = edge_fn(v2.x, v2.y, v0.x, v0.y, v1.x, v1.y) + (T)kEps;
  SoftFrag<T> f = soft_frag_eval_shared(v0, v1, v2, area, (T)1 / area, px, py, persp, clip_bary);
  if (with_dist) {
    const T d = point_triangle_dist2(px, py, v0, v1, v2);
    f.dist = f.inside ? -d : d;
  }
  return f;
}
// the signed squared distance of a fragment evaluated without it (soft_frag_eval(..., with_dist = false))
template <typename T> MVR_HD void soft_frag_add_dist(SoftFrag<T> *f, const V3<T> &v0, const V3<T> &v1, const V3<T> &v2, T px, T py) {
  const T d = point_triangle_dist2(px, py, v0, v1, v2);
  f->dist = f->inside ? -d : d;
}

// CheckPixelInsideFace with blur_radius > 0: true when the pixel is inside the face or nearer than sqrt(blur_radius).
template <typename T> MVR_HD bool face_test_soft(const V3<T> &v0, const V3<T> &v1, const V3<T> &v2, T px, T py, T blur_radius, T blur,
                                                 bool persp, bool clip_bary, bool cull_backfaces, SoftFrag<T> *out) {
  const T xmin = tmin(tmin(v0.x, v1.x), v2.x) - blur, xmax = tmax(tmax(v0.x, v1.x), v2.x) + blur;
  const T ymin = tmin(tmin(v0.y, v1.y), v2.y) - blur, ymax = tmax(tmax(v0.y, v1.y), v2.y) + blur;
  const T zmin = tmin(tmin(v0.z, v1.z), v2.z), zmax = tmax(tmax(v0.z, v1.z), v2.z);
  const bool outside = px > xmax || px < xmin || py > ymax || py < ymin || zmin < (T)kEps;
  const T face_area = edge_fn(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
  const bool zero_area = face_area <= (T)kEps && face_area >= -(T)kEps;
  if (zmax < (T)0 || (cull_backfaces && face_area < (T)0) || outside || zero_area) return false;
  const SoftFrag<T> f = soft_frag_eval(v0, v1, v2, px, py, persp, clip_bary);
  if (f.z < (T)0) return false;
  if (!f.inside && f.dist >= blur_radius) return false;
  *out = f;
  return true;
}

// Backward of one soft fragment: upstream gradients of its colour (gc), depth (gz) and signed distance (gd) ->
// acc[18] (gR 9, gT 3, g camera centre 3, g light 3), like phong_pixel_bwd.  shade: 0 = Phong colour, 1 = texel only.
template <typename T> MVR_HD void soft_fragment_bwd(const PixelGradIn<T> &cam, T px, T py, bool clip_bary, int shade,
                                                    const V3<T> &X0, const V3<T> &X1, const V3<T> &X2,
                                                    const V3<T> &N0, const V3<T> &N1, const V3<T> &N2,
                                                    const V3<T> &c0, const V3<T> &c1, const V3<T> &c2,
                                                    const V3<T> &G, T gzbuf, T gdist, T acc[18]) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  // ---- forward recompute ---------------------------------------------------------------------------------------
  const V3<T> Xv0 = world_to_view(cam.R, cam.Tr, X0), Xv1 = world_to_view(cam.R, cam.Tr, X1), Xv2 = world_to_view(cam.R, cam.Tr, X2);
  const T rz0 = trcp(Xv0.z), rz1 = trcp(Xv1.z), rz2 = trcp(Xv2.z);
  const T u0 = (cam.s * Xv0.x) * rz0, v0 = (cam.s * Xv0.y) * rz0, z0 = Xv0.z;
  const T u1 = (cam.s * Xv1.x) * rz1, v1 = (cam.s * Xv1.y) * rz1, z1 = Xv1.z;
  const T u2 = (cam.s * Xv2.x) * rz2, v2 = (cam.s * Xv2.y) * rz2, z2 = Xv2.z;
  const T area = edge_fn(u2, v2, u0, v0, u1, v1) + (T)kEps;
  const T ra = trcp(area);
  const T e0 = edge_fn(px, py, u1, v1, u2, v2), e1 = edge_fn(px, py, u2, v2, u0, v0), e2 = edge_fn(px, py, u0, v0, u1, v1);
  const T a0 = e0 * ra, a1 = e1 * ra, a2 = e2 * ra;
  T w0 = a0, w1 = a1, w2 = a2, rden = (T)1;
  bool den_clamped = false;
  if (cam.persp) {
    const T t0 = (a0 * z1) * z2, t1 = (z0 * a1) * z2, t2 = (z0 * z1) * a2;
    const T sum = (t0 + t1) + t2;
    rden = trcp(tmax(sum, (T)kEps));
    den_clamped = !(sum > (T)kEps);
    w0 = t0 * rden; w1 = t1 * rden; w2 = t2 * rden;
  }
  const bool inside = w0 > (T)0 && w1 > (T)0 && w2 > (T)0;
  T x0 = w0, x1 = w1, x2 = w2, rS = (T)1;
  bool s_clamped = false;
  if (clip_bary) {
    x0 = tmax(tmin(w0, (T)1), (T)0); x1 = tmax(tmin(w1, (T)1), (T)0); x2 = tmax(tmin(w2, (T)1), (T)0);
    const T sm = (x0 + x1) + x2;
    s_clamped = !(sm > (T)1e-5);
    rS = trcp(tmax(sm, (T)1e-5));
  }
  const T b0 = x0 * rS, b1 = x1 * rS, b2 = x2 * rS;
  // ---- colour backward -> gradients of the (clipped) barycentrics, of the camera centre and of the light --------------
  T gb0, gb1, gb2;
  V3<T> g_cc = {(T)0, (T)0, (T)0}, g_l = {(T)0, (T)0, (T)0};
  if (shade == 0) {
    const V3<T> p = {(b0 * X0.x + b1 * X1.x) + b2 * X2.x, (b0 * X0.y + b1 * X1.y) + b2 * X2.y, (b0 * X0.z + b1 * X1.z) + b2 * X2.z};
    const V3<T> n = {(b0 * N0.x + b1 * N1.x) + b2 * N2.x, (b0 * N0.y + b1 * N1.y) + b2 * N2.y, (b0 * N0.z + b1 * N1.z) + b2 * N2.z};
    const V3<T> tex = {(b0 * c0.x + b1 * c1.x) + b2 * c2.x, (b0 * c0.y + b1 * c1.y) + b2 * c2.y, (b0 * c0.z + b1 * c1.z) + b2 * c2.z};
    T idn, idl, idv;
    bool cn, cl, cv;
    const V3<T> nh = normalize_rs(n, (T)kNormalizeEps, &idn, &cn);
    V3<T> lh = cam.l;
    idl = (T)1; cl = true;
    if (!cam.l_is_unit) lh = normalize_rs(cam.l, (T)kNormalizeEps, &idl, &cl);
    const T cosang = dot(nh, lh);
    const bool lit = cosang > (T)0;
    const V3<T> view = {cam.cc.x - p.x, cam.cc.y - p.y, cam.cc.z - p.z};
    const V3<T> vh = normalize_rs(view, (T)kNormalizeEps, &idv, &cv);
    const V3<T> refl = {-lh.x + (T)2 * (cosang * nh.x), -lh.y + (T)2 * (cosang * nh.y), -lh.z + (T)2 * (cosang * nh.z)};
    const T dvr = dot(vh, refl);
    const bool spec_on = lit && dvr > (T)0;
    const T alpha = spec_on ? dvr : (T)0;
    const T sc = cam.ph.ambient + cam.ph.diffuse * (lit ? cosang : (T)0);
    const V3<T> g_tex = {G.x * sc, G.y * sc, G.z * sc};
    const T g_sc = (G.x * tex.x + G.y * tex.y) + G.z * tex.z;
    T g_cos = lit ? cam.ph.diffuse * g_sc : (T)0;
    V3<T> g_nh = {(T)0, (T)0, (T)0}, g_lh = {(T)0, (T)0, (T)0}, g_p = {(T)0, (T)0, (T)0};
    if (spec_on) {
      const T g_spec = (G.x + G.y) + G.z;
      T dpow;
      if (cam.ph.shininess == (T)64) {
        const T a2_ = alpha * alpha, a4 = a2_ * a2_, a8 = a4 * a4, a16 = a8 * a8, a32 = a16 * a16;
        dpow = (T)64 * (((a32 * a16) * (a8 * a4)) * (a2_ * alpha));
      } else {
        dpow = cam.ph.shininess * pow_shininess(alpha, cam.ph.shininess - (T)1);
      }
      const T g_dvr = cam.ph.specular * dpow * g_spec;
      const V3<T> g_vh = {g_dvr * refl.x, g_dvr * refl.y, g_dvr * refl.z};
      const V3<T> g_r = {g_dvr * vh.x, g_dvr * vh.y, g_dvr * vh.z};
      g_lh = {-g_r.x, -g_r.y, -g_r.z};
      g_cos += (T)2 * dot(g_r, nh);
      g_nh = {(T)2 * cosang * g_r.x, (T)2 * cosang * g_r.y, (T)2 * cosang * g_r.z};
      const V3<T> g_view = normalize_rs_bwd(vh, idv, cv, g_vh);
      g_cc = g_view;
      g_p = {-g_view.x, -g_view.y, -g_view.z};
    }
    g_nh = {g_nh.x + g_cos * lh.x, g_nh.y + g_cos * lh.y, g_nh.z + g_cos * lh.z};
    g_lh = {g_lh.x + g_cos * nh.x, g_lh.y + g_cos * nh.y, g_lh.z + g_cos * nh.z};
    const V3<T> g_n = normalize_rs_bwd(nh, idn, cn, g_nh);
    g_l = cam.l_is_unit ? g_lh : normalize_rs_bwd(lh, idl, cl, g_lh);
    gb0 = (dot(g_p, X0) + dot(g_n, N0)) + dot(g_tex, c0);
    gb1 = (dot(g_p, X1) + dot(g_n, N1)) + dot(g_tex, c1);
    gb2 = (dot(g_p, X2) + dot(g_n, N2)) + dot(g_tex, c2);
  } else {   // texel only (SoftSilhouetteShader's RGB)
    gb0 = dot(G, c0); gb1 = dot(G, c1); gb2 = dot(G, c2);
  }
  // depth: z = sum b_i z_i
  gb0 += gzbuf * z0; gb1 += gzbuf * z1; gb2 += gzbuf * z2;
  T gz0 = gzbuf * b0, gz1 = gzbuf * b1, gz2 = gzbuf * b2;
  // clip backward: b_i = x_i / max(sum x, 1e-5), x_i = clamp(w_i, 0, 1)
  T gw0 = gb0, gw1 = gb1, gw2 = gb2;
  if (clip_bary) {
    const T sdot = s_clamped ? (T)0 : (gb0 * b0 + gb1 * b1) + gb2 * b2;
    const T gx0 = (gb0 - sdot) * rS, gx1 = (gb1 - sdot) * rS, gx2 = (gb2 - sdot) * rS;
    gw0 = (w0 > (T)0 && w0 < (T)1) ? gx0 : (T)0;
    gw1 = (w1 > (T)0 && w1 < (T)1) ? gx1 : (T)0;
    gw2 = (w2 > (T)0 && w2 < (T)1) ? gx2 : (T)0;
  }
  T ga0 = gw0, ga1 = gw1, ga2 = gw2;
  if (cam.persp) {
    T gt0, gt1, gt2;
    if (den_clamped) {
      gt0 = gw0 * rden; gt1 = gw1 * rden; gt2 = gw2 * rden;
    } else {
      const T s_ = (gw0 * w0 + gw1 * w1) + gw2 * w2;
      gt0 = (gw0 - s_) * rden; gt1 = (gw1 - s_) * rden; gt2 = (gw2 - s_) * rden;
    }
    ga0 = gt0 * (z1 * z2); gz1 += gt0 * (a0 * z2); gz2 += gt0 * (a0 * z1);
    ga1 = gt1 * (z0 * z2); gz0 += gt1 * (a1 * z2); gz2 += gt1 * (z0 * a1);
    ga2 = gt2 * (z0 * z1); gz0 += gt2 * (z1 * a2); gz1 += gt2 * (z0 * a2);
  }
  const T ge0 = ga0 * ra, ge1 = ga1 * ra, ge2 = ga2 * ra;
  const T g_area = -((ga0 * a0 + ga1 * a1) + ga2 * a2) * ra;
  T gu0 = (T)0, gv0 = (T)0, gu1 = (T)0, gv1 = (T)0, gu2 = (T)0, gv2 = (T)0;
  gu1 += ge0 * (py - v2); gv1 += ge0 * (u2 - px); gu2 += ge0 * (v1 - py); gv2 += ge0 * (px - u1);
  gu2 += ge1 * (py - v0); gv2 += ge1 * (u0 - px); gu0 += ge1 * (v2 - py); gv0 += ge1 * (px - u2);
  gu0 += ge2 * (py - v1); gv0 += ge2 * (u1 - px); gu1 += ge2 * (v0 - py); gv1 += ge2 * (px - u0);
  gu2 += g_area * (v1 - v0); gv2 += g_area * (-(u1 - u0));
  gu0 += g_area * (v2 - v1); gv0 += g_area * (u1 - u2);
  gu1 += g_area * (v0 - v2); gv1 += g_area * (u2 - u0);
  // signed squared distance to the nearest edge (a, b): q = a + t (b - a); d|p - q|^2 / da = -2 (1 - t) r, / db = -2 t r
  if (gdist != (T)0) {
    int edge;
    T t;
    const V3<T> P0 = {u0, v0, z0}, P1 = {u1, v1, z1}, P2 = {u2, v2, z2};
    (void)point_triangle_dist2(px, py, P0, P1, P2, &edge, &t);
    const T ax = edge == 2 ? u1 : u0, ay = edge == 2 ? v1 : v0, bx = edge == 0 ? u1 : u2, by = edge == 0 ? v1 : v2;
    T wa, wb, rx, ry;
    if (t < (T)0) { wa = (T)0; wb = (T)1; rx = px - bx; ry = py - by; }   // degenerate edge: distance to b
    else { wa = (T)1 - t; wb = t; rx = px - (ax + t * (bx - ax)); ry = py - (ay + t * (by - ay)); }
    const T sgn = inside ? (T)-1 : (T)1;
    const T k = (T)-2 * sgn * gdist;
    const T gax = k * wa * rx, gay = k * wa * ry, gbx = k * wb * rx, gby = k * wb * ry;
    if (edge == 2) { gu1 += gax; gv1 += gay; } else { gu0 += gax; gv0 += gay; }
    if (edge == 0) { gu1 += gbx; gv1 += gby; } else { gu2 += gbx; gv2 += gby; }
  }
  auto proj_bwd = [&](T rz, T u, T v, T gu, T gv, T gz, const V3<T> &Xw) {
    const T gx = gu * cam.s * rz, gy = gv * cam.s * rz;
    const T gzz = gz - (gu * u + gv * v) * rz;
    acc[0] += Xw.x * gx; acc[1] += Xw.x * gy; acc[2] += Xw.x * gzz;
    acc[3] += Xw.y * gx; acc[4] += Xw.y * gy; acc[5] += Xw.y * gzz;
    acc[6] += Xw.z * gx; acc[7] += Xw.z * gy; acc[8] += Xw.z * gzz;
    acc[9] += gx; acc[10] += gy; acc[11] += gzz;
  };
  proj_bwd(rz0, u0, v0, gu0, gv0, gz0, X0);
  proj_bwd(rz1, u1, v1, gu1, gv1, gz1, X1);
  proj_bwd(rz2, u2, v2, gu2, gv2, gz2, X2);
  acc[12] += g_cc.x; acc[13] += g_cc.y; acc[14] += g_cc.z;
  acc[15] += g_l.x; acc[16] += g_l.y; acc[17] += g_l.z;
}

// Backward of one soft fragment that lies on a PIECE of a face the z-clip plane cuts (round 4; rounds 2-3 differentiated the whole
// face, which is the wrong function next to the clip line: the piece's edge there is a silhouette edge of its own, and it moves
// with the camera).  The forward is PyTorch3D's: clip_faces() turns the face into one or two triangles whose new vertices lie on
// the edges leaving the lone vertex p1, at the view-depth parameter w = (z1 - zc) / (z1 - z_other); the rasteriser runs on the
// piece; its (clipped) barycentrics b_k are mapped back to the face's, B_j = sum_k b_k beta_kj, for the colour; depth and signed
// distance are the piece's.  In VIEW space a new vertex is P1 + w (P_other - P1): everything below is the chain rule through
// that — piece vertices Q_k (view space) -> NDC -> edge functions, perspective correction, clip -> b_k -> (B_j, z, dist), and
// back: dL/dQ_k, then to the face's own vertices (directly, or through (1 - w, w) and dw/dz), then R, T.  `sub`: which piece
// (clip_face's t0 / t1).  Same accumulators as soft_fragment_bwd.
// (used by k_mesh_soft_backward_cut, a launch of its own: inlined into — or called from — k_mesh_soft_backward's pixel loop this
// rare path made the register allocator spill in the common one: 0.93 -> 2.4-2.6 ms)
template <typename T> MVR_HD void soft_fragment_bwd_piece(const PixelGradIn<T> &cam, T px, T py, bool clip_bary, int shade,
                                                          const V3<T> &X0, const V3<T> &X1, const V3<T> &X2,
                                                          const V3<T> &N0, const V3<T> &N1, const V3<T> &N2,
                                                          const V3<T> &c0, const V3<T> &c1, const V3<T> &c2,
                                                          const V3<T> &G, T gzbuf, T gdist, T zc, int sub, T acc[18]) {
#if defined(__clang__)
#pragma clang fp contract(fast)
#endif
  const V3<T> Pv0 = world_to_view(cam.R, cam.Tr, X0), Pv1 = world_to_view(cam.R, cam.Tr, X1), Pv2 = world_to_view(cam.R, cam.Tr, X2);
  const bool h0 = Pv0.z < zc, h1 = Pv1.z < zc, h2 = Pv2.z < zc;
  const int nb = (h0 ? 1 : 0) + (h1 ? 1 : 0) + (h2 ? 1 : 0);
  if (nb == 0 || nb == 3) {   // (not cut in this view after all: the whole face)
    soft_fragment_bwd<T>(cam, px, py, clip_bary, shade, X0, X1, X2, N0, N1, N2, c0, c1, c2, G, gzbuf, gdist, acc);
    return;
  }
  auto pick = [](int i, const V3<T> &a, const V3<T> &b, const V3<T> &c) -> V3<T> { return i == 0 ? a : (i == 1 ? b : c); };
  const bool alone = nb == 1;
  const int i1 = (h2 == alone) ? 2 : ((h1 == alone) ? 1 : 0), i2 = i1 == 2 ? 0 : i1 + 1, i3 = i2 == 2 ? 0 : i2 + 1;
  const V3<T> P1 = pick(i1, Pv0, Pv1, Pv2), P2 = pick(i2, Pv0, Pv1, Pv2), P3 = pick(i3, Pv0, Pv1, Pv2);
  const T D2 = P1.z - P2.z, D3 = P1.z - P3.z;
  const T w2 = (P1.z - zc) / D2, w3 = (P1.z - zc) / D3;
  const V3<T> P4 = {P1.x + w2 * (P2.x - P1.x), P1.y + w2 * (P2.y - P1.y), P1.z + w2 * (P2.z - P1.z)};
  const V3<T> P5 = {P1.x + w3 * (P3.x - P1.x), P1.y + w3 * (P3.y - P1.y), P1.z + w3 * (P3.z - P1.z)};
  // piece vertices: kind 0 = p1, 1 = p2, 2 = p3, 3 = p4 (on p1 p2), 4 = p5 (on p1 p3)
  int kd0, kd1, kd2;
  if (nb == 1) { if (sub == 0) { kd0 = 3; kd1 = 1; kd2 = 4; } else { kd0 = 4; kd1 = 1; kd2 = 2; } }
  else { kd0 = 0; kd1 = 3; kd2 = 4; }
  auto vert_of = [&](int kd) -> V3<T> { return kd == 0 ? P1 : (kd == 1 ? P2 : (kd == 2 ? P3 : (kd == 3 ? P4 : P5))); };
  const V3<T> Q0 = vert_of(kd0), Q1 = vert_of(kd1), Q2 = vert_of(kd2);
  // beta_kj in the order (p1, p2, p3) of the face's vertices
  auto beta_of = [&](int kd, T *bp1, T *bp2, T *bp3) {
    *bp1 = kd == 0 ? (T)1 : (kd == 3 ? (T)1 - w2 : (kd == 4 ? (T)1 - w3 : (T)0));
    *bp2 = kd == 1 ? (T)1 : (kd == 3 ? w2 : (T)0);
    *bp3 = kd == 2 ? (T)1 : (kd == 4 ? w3 : (T)0);
  };
  T be0[3], be1[3], be2[3];
  beta_of(kd0, &be0[0], &be0[1], &be0[2]); beta_of(kd1, &be1[0], &be1[1], &be1[2]); beta_of(kd2, &be2[0], &be2[1], &be2[2]);
  // ---- forward on the piece ------------------------------------------------------------------------------------------
  const T rz0 = trcp(Q0.z), rz1 = trcp(Q1.z), rz2 = trcp(Q2.z);
  const T u0 = (cam.s * Q0.x) * rz0, v0 = (cam.s * Q0.y) * rz0, z0 = Q0.z;
  const T u1 = (cam.s * Q1.x) * rz1, v1 = (cam.s * Q1.y) * rz1, z1 = Q1.z;
  const T u2 = (cam.s * Q2.x) * rz2, v2 = (cam.s * Q2.y) * rz2, z2 = Q2.z;
  const T area = edge_fn(u2, v2, u0, v0, u1, v1) + (T)kEps;
  const T ra = trcp(area);
  const T e0 = edge_fn(px, py, u1, v1, u2, v2), e1 = edge_fn(px, py, u2, v2, u0, v0), e2 = edge_fn(px, py, u0, v0, u1, v1);
  const T a0 = e0 * ra, a1 = e1 * ra, a2 = e2 * ra;
  T w0_ = a0, w1_ = a1, w2_ = a2, rden = (T)1;
  bool den_clamped = false;
  if (cam.persp) {
    const T t0 = (a0 * z1) * z2, t1 = (z0 * a1) * z2, t2 = (z0 * z1) * a2;
    const T sum = (t0 + t1) + t2;
    rden = trcp(tmax(sum, (T)kEps));
    den_clamped = !(sum > (T)kEps);
    w0_ = t0 * rden; w1_ = t1 * rden; w2_ = t2 * rden;
  }
  const bool inside = w0_ > (T)0 && w1_ > (T)0 && w2_ > (T)0;
  T x0 = w0_, x1 = w1_, x2 = w2_, rS = (T)1;
  bool s_clamped = false;
  if (clip_bary) {
    x0 = tmax(tmin(w0_, (T)1), (T)0); x1 = tmax(tmin(w1_, (T)1), (T)0); x2 = tmax(tmin(w2_, (T)1), (T)0);
    const T sm = (x0 + x1) + x2;
    s_clamped = !(sm > (T)1e-5);
    rS = trcp(tmax(sm, (T)1e-5));
  }
  const T b0 = x0 * rS, b1 = x1 * rS, b2 = x2 * rS;
  // the face's barycentrics, in (p1, p2, p3) order, then in the face's own vertex order
  const T Bp1 = (b0 * be0[0] + b1 * be1[0]) + b2 * be2[0], Bp2 = (b0 * be0[1] + b1 * be1[1]) + b2 * be2[1], Bp3 = (b0 * be0[2] + b1 * be1[2]) + b2 * be2[2];
  auto unpick = [&](int j) -> T { return j == i1 ? Bp1 : (j == i2 ? Bp2 : Bp3); };
  const T B0 = unpick(0), B1 = unpick(1), B2 = unpick(2);
  // ---- colour backward: gradients of the FACE's barycentrics, of the camera centre and of the light (as soft_fragment_bwd) ----
  T gB0, gB1, gB2;
  V3<T> g_cc = {(T)0, (T)0, (T)0}, g_l = {(T)0, (T)0, (T)0};
  if (shade == 0) {
    const V3<T> p = {(B0 * X0.x + B1 * X1.x) + B2 * X2.x, (B0 * X0.y + B1 * X1.y) + B2 * X2.y, (B0 * X0.z + B1 * X1.z) + B2 * X2.z};
    const V3<T> n = {(B0 * N0.x + B1 * N1.x) + B2 * N2.x, (B0 * N0.y + B1 * N1.y) + B2 * N2.y, (B0 * N0.z + B1 * N1.z) + B2 * N2.z};
    const V3<T> tex = {(B0 * c0.x + B1 * c1.x) + B2 * c2.x, (B0 * c0.y + B1 * c1.y) + B2 * c2.y, (B0 * c0.z + B1 * c1.z) + B2 * c2.z};
    T idn, idl, idv;
    bool cn, cl, cv;
    const V3<T> nh = normalize_rs(n, (T)kNormalizeEps, &idn, &cn);
    V3<T> lh = cam.l;
    idl = (T)1; cl = true;
    if (!cam.l_is_unit) lh = normalize_rs(cam.l, (T)kNormalizeEps, &idl, &cl);
    const T cosang = dot(nh, lh);
    const bool lit = cosang > (T)0;
    const V3<T> view = {cam.cc.x - p.x, cam.cc.y - p.y, cam.cc.z - p.z};
    const V3<T> vh = normalize_rs(view, (T)kNormalizeEps, &idv, &cv);
    const V3<T> refl = {-lh.x + (T)2 * (cosang * nh.x), -lh.y + (T)2 * (cosang * nh.y), -lh.z + (T)2 * (cosang * nh.z)};
    const T dvr = dot(vh, refl);
    const bool spec_on = lit && dvr > (T)0;
    const T alpha = spec_on ? dvr : (T)0;
    const T sc = cam.ph.ambient + cam.ph.diffuse * (lit ? cosang : (T)0);
    const V3<T> g_tex = {G.x * sc, G.y * sc, G.z * sc};
    const T g_sc = (G.x * tex.x + G.y * tex.y) + G.z * tex.z;
    T g_cos = lit ? cam.ph.diffuse * g_sc : (T)0;
    V3<T> g_nh = {(T)0, (T)0, (T)0}, g_lh = {(T)0, (T)0, (T)0}, g_p = {(T)0, (T)0, (T)0};
    if (spec_on) {
      const T g_spec = (G.x + G.y) + G.z;
      const T dpow = cam.ph.shininess * pow_shininess(alpha, cam.ph.shininess - (T)1);
      const T g_dvr = cam.ph.specular * dpow * g_spec;
      const V3<T> g_vh = {g_dvr * refl.x, g_dvr * refl.y, g_dvr * refl.z};
      const V3<T> g_r = {g_dvr * vh.x, g_dvr * vh.y, g_dvr * vh.z};
      g_lh = {-g_r.x, -g_r.y, -g_r.z};
      g_cos += (T)2 * dot(g_r, nh);
      g_nh = {(T)2 * cosang * g_r.x, (T)2 * cosang * g_r.y, (T)2 * cosang * g_r.z};
      const V3<T> g_view = normalize_rs_bwd(vh, idv, cv, g_vh);
      g_cc = g_view;
      g_p = {-g_view.x, -g_view.y, -g_view.z};
    }
    g_nh = {g_nh.x + g_cos * lh.x, g_nh.y + g_cos * lh.y, g_nh.z + g_cos * lh.z};
    g_lh = {g_lh.x + g_cos * nh.x, g_lh.y + g_cos * nh.y, g_lh.z + g_cos * nh.z};
    const V3<T> g_n = normalize_rs_bwd(nh, idn, cn, g_nh);
    g_l = cam.l_is_unit ? g_lh : normalize_rs_bwd(lh, idl, cl, g_lh);
    gB0 = (dot(g_p, X0) + dot(g_n, N0)) + dot(g_tex, c0);
    gB1 = (dot(g_p, X1) + dot(g_n, N1)) + dot(g_tex, c1);
    gB2 = (dot(g_p, X2) + dot(g_n, N2)) + dot(g_tex, c2);
  } else {
    gB0 = dot(G, c0); gB1 = dot(G, c1); gB2 = dot(G, c2);
  }
  // face barycentrics in (p1, p2, p3) order -> piece barycentrics and the map's own parameters
  const T gBp1 = i1 == 0 ? gB0 : (i1 == 1 ? gB1 : gB2), gBp2 = i2 == 0 ? gB0 : (i2 == 1 ? gB1 : gB2), gBp3 = i3 == 0 ? gB0 : (i3 == 1 ? gB1 : gB2);
  T gb0 = (gBp1 * be0[0] + gBp2 * be0[1]) + gBp3 * be0[2], gb1 = (gBp1 * be1[0] + gBp2 * be1[1]) + gBp3 * be1[2],
    gb2 = (gBp1 * be2[0] + gBp2 * be2[1]) + gBp3 * be2[2];
  // d B / d w2, d B / d w3: a new vertex's row of beta is (1 - w, w) on (p1, other)
  auto gw_beta = [&](int kd, T bk, T *g2, T *g3) {
    if (kd == 3) *g2 += bk * (gBp2 - gBp1);
    if (kd == 4) *g3 += bk * (gBp3 - gBp1);
  };
  T gw2 = (T)0, gw3 = (T)0;
  gw_beta(kd0, b0, &gw2, &gw3); gw_beta(kd1, b1, &gw2, &gw3); gw_beta(kd2, b2, &gw2, &gw3);
  // depth: z = sum b_k z_k (the piece's)
  gb0 += gzbuf * z0; gb1 += gzbuf * z1; gb2 += gzbuf * z2;
  T gz0 = gzbuf * b0, gz1 = gzbuf * b1, gz2 = gzbuf * b2;
  T gwb0 = gb0, gwb1 = gb1, gwb2 = gb2;
  if (clip_bary) {
    const T sdot = s_clamped ? (T)0 : (gb0 * b0 + gb1 * b1) + gb2 * b2;
    const T gx0 = (gb0 - sdot) * rS, gx1 = (gb1 - sdot) * rS, gx2 = (gb2 - sdot) * rS;
    gwb0 = (w0_ > (T)0 && w0_ < (T)1) ? gx0 : (T)0;
    gwb1 = (w1_ > (T)0 && w1_ < (T)1) ? gx1 : (T)0;
    gwb2 = (w2_ > (T)0 && w2_ < (T)1) ? gx2 : (T)0;
  }
  T ga0 = gwb0, ga1 = gwb1, ga2 = gwb2;
  if (cam.persp) {
    T gt0, gt1, gt2;
    if (den_clamped) { gt0 = gwb0 * rden; gt1 = gwb1 * rden; gt2 = gwb2 * rden; }
    else {
      const T s_ = (gwb0 * w0_ + gwb1 * w1_) + gwb2 * w2_;
      gt0 = (gwb0 - s_) * rden; gt1 = (gwb1 - s_) * rden; gt2 = (gwb2 - s_) * rden;
    }
    ga0 = gt0 * (z1 * z2); gz1 += gt0 * (a0 * z2); gz2 += gt0 * (a0 * z1);
    ga1 = gt1 * (z0 * z2); gz0 += gt1 * (a1 * z2); gz2 += gt1 * (z0 * a1);
    ga2 = gt2 * (z0 * z1); gz0 += gt2 * (z1 * a2); gz1 += gt2 * (z0 * a2);
  }
  const T ge0 = ga0 * ra, ge1 = ga1 * ra, ge2 = ga2 * ra;
  const T g_area = -((ga0 * a0 + ga1 * a1) + ga2 * a2) * ra;
  T gu0 = (T)0, gv0 = (T)0, gu1 = (T)0, gv1 = (T)0, gu2 = (T)0, gv2 = (T)0;
  gu1 += ge0 * (py - v2); gv1 += ge0 * (u2 - px); gu2 += ge0 * (v1 - py); gv2 += ge0 * (px - u1);
  gu2 += ge1 * (py - v0); gv2 += ge1 * (u0 - px); gu0 += ge1 * (v2 - py); gv0 += ge1 * (px - u2);
  gu0 += ge2 * (py - v1); gv0 += ge2 * (u1 - px); gu1 += ge2 * (v0 - py); gv1 += ge2 * (px - u0);
  gu2 += g_area * (v1 - v0); gv2 += g_area * (-(u1 - u0));
  gu0 += g_area * (v2 - v1); gv0 += g_area * (u1 - u2);
  gu1 += g_area * (v0 - v2); gv1 += g_area * (u2 - u0);
  if (gdist != (T)0) {
    int edge;
    T t;
    const V3<T> E0 = {u0, v0, z0}, E1 = {u1, v1, z1}, E2 = {u2, v2, z2};
    (void)point_triangle_dist2(px, py, E0, E1, E2, &edge, &t);
    const T ax = edge == 2 ? u1 : u0, ay = edge == 2 ? v1 : v0, bx = edge == 0 ? u1 : u2, by = edge == 0 ? v1 : v2;
    T wa, wb, rx, ry;
    if (t < (T)0) { wa = (T)0; wb = (T)1; rx = px - bx; ry = py - by; }
    else { wa = (T)1 - t; wb = t; rx = px - (ax + t * (bx - ax)); ry = py - (ay + t * (by - ay)); }
    const T sgn = inside ? (T)-1 : (T)1;
    const T k = (T)-2 * sgn * gdist;
    const T gax = k * wa * rx, gay = k * wa * ry, gbx = k * wb * rx, gby = k * wb * ry;
    if (edge == 2) { gu1 += gax; gv1 += gay; } else { gu0 += gax; gv0 += gay; }
    if (edge == 0) { gu1 += gbx; gv1 += gby; } else { gu2 += gbx; gv2 += gby; }
  }
  // projection backward: gradient w.r.t. the piece's VIEW-space vertices
  auto proj_bwd_v = [&](T rz, T u, T v, T gu, T gv, T gz) -> V3<T> {
    return {gu * cam.s * rz, gv * cam.s * rz, gz - (gu * u + gv * v) * rz};
  };
  const V3<T> gQ0 = proj_bwd_v(rz0, u0, v0, gu0, gv0, gz0), gQ1 = proj_bwd_v(rz1, u1, v1, gu1, gv1, gz1), gQ2 = proj_bwd_v(rz2, u2, v2, gu2, gv2, gz2);
  // piece vertices -> (p1, p2, p3) of the face, and the two clip parameters
  V3<T> gP1 = {(T)0, (T)0, (T)0}, gP2 = gP1, gP3 = gP1;
  auto scatter = [&](int kd, const V3<T> &g) {
    if (kd == 0) { gP1 = {gP1.x + g.x, gP1.y + g.y, gP1.z + g.z}; }
    else if (kd == 1) { gP2 = {gP2.x + g.x, gP2.y + g.y, gP2.z + g.z}; }
    else if (kd == 2) { gP3 = {gP3.x + g.x, gP3.y + g.y, gP3.z + g.z}; }
    else if (kd == 3) {
      gP1 = {gP1.x + ((T)1 - w2) * g.x, gP1.y + ((T)1 - w2) * g.y, gP1.z + ((T)1 - w2) * g.z};
      gP2 = {gP2.x + w2 * g.x, gP2.y + w2 * g.y, gP2.z + w2 * g.z};
      gw2 += (g.x * (P2.x - P1.x) + g.y * (P2.y - P1.y)) + g.z * (P2.z - P1.z);
    } else {
      gP1 = {gP1.x + ((T)1 - w3) * g.x, gP1.y + ((T)1 - w3) * g.y, gP1.z + ((T)1 - w3) * g.z};
      gP3 = {gP3.x + w3 * g.x, gP3.y + w3 * g.y, gP3.z + w3 * g.z};
      gw3 += (g.x * (P3.x - P1.x) + g.y * (P3.y - P1.y)) + g.z * (P3.z - P1.z);
    }
  };
  scatter(kd0, gQ0); scatter(kd1, gQ1); scatter(kd2, gQ2);
  // w = (z1 - zc) / (z1 - zo): dw/dz1 = (zc - zo) / (z1 - zo)^2, dw/dzo = (z1 - zc) / (z1 - zo)^2
  {
    const T r2 = (T)1 / (D2 * D2), r3 = (T)1 / (D3 * D3);
    gP1.z += gw2 * (zc - P2.z) * r2 + gw3 * (zc - P3.z) * r3;
    gP2.z += gw2 * (P1.z - zc) * r2;
    gP3.z += gw3 * (P1.z - zc) * r3;
  }
  // (p1, p2, p3) -> the face's own vertices, then R, T through Xv = Xw R + T
  auto to_acc = [&](const V3<T> &Xw, const V3<T> &g) {
    acc[0] += Xw.x * g.x; acc[1] += Xw.x * g.y; acc[2] += Xw.x * g.z;
    acc[3] += Xw.y * g.x; acc[4] += Xw.y * g.y; acc[5] += Xw.y * g.z;
    acc[6] += Xw.z * g.x; acc[7] += Xw.z * g.y; acc[8] += Xw.z * g.z;
    acc[9] += g.x; acc[10] += g.y; acc[11] += g.z;
  };
  to_acc(pick(i1, X0, X1, X2), gP1); to_acc(pick(i2, X0, X1, X2), gP2); to_acc(pick(i3, X0, X1, X2), gP3);
  acc[12] += g_cc.x; acc[13] += g_cc.y; acc[14] += g_cc.z;
  acc[15] += g_l.x; acc[16] += g_l.y; acc[17] += g_l.z;
}

// softmax_rgb_blend / sigmoid_alpha_blend constants (renderer/blending.py): znear 1, zfar 100, eps 1e-10
MVR_HD float texp(float x) { return expf(x); }
MVR_HD double texp(double x) { return exp(x); }
template <typename T> MVR_HD T sigmoid_neg(T d, T sigma) { return (T)1 / ((T)1 + texp(d / sigma)); }   // sigmoid(-d / sigma)

// Fold the per-view camera-centre gradient into (gR, gT):  cc_i = -sum_j T_j R[i][j].
template <typename T> MVR_HD void fold_camera_center(const T R[9], const T Tr[3], const T gcc[3], T gR[9], T gT[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      gR[3 * i + j] += -gcc[i] * Tr[j];
      gT[j] += -gcc[i] * R[3 * i + j];
    }
}

}  // namespace mvr
