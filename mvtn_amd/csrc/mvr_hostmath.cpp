// mvr_hostmath.cpp — TEST-ONLY host (g++) instantiation of the templated arithmetic in
// mvr_math.h, in float and double.  It lets the CPU test-suite (no GPU in the build container)
// check the exact code the HIP kernels inline: forward parity against the oracle, and the
// analytic backward against finite differences in double.  Never loaded by the product path.
#include <cstdint>
#include <cstring>
#include <vector>

#include "mvr_math.h"

using namespace mvr;

template <typename T>
static void t_look_at(int n, const T *azim, const T *elev, const T *dist, T *R, T *Tr, int32_t *valid) {
  for (int i = 0; i < n; ++i) {
    look_at<T>(azim[i], elev[i], dist[i], R + 9 * i, Tr + 3 * i);
    if (valid) valid[i] = rotation_valid<T>(R + 9 * i, (T)1e-6) ? 1 : 0;
  }
}

template <typename T>
static void t_look_at_bwd(int n, const T *azim, const T *elev, const T *dist, const T *gR, const T *gT, T *ga, T *ge, T *gd) {
  for (int i = 0; i < n; ++i) look_at_bwd<T>(azim[i], elev[i], dist[i], gR + 9 * i, gT + 3 * i, ga + i, ge + i, gd + i);
}

template <typename T> static V3<T> ld(const T *p) { return {p[0], p[1], p[2]}; }

// rasterise (face-major, (z, face) key order) + shade, mirroring k_project_verts + k_mesh_raster_shade
template <typename T>
static void t_mesh_forward(int B, int V, int H, int W, const T *verts, const int32_t *faces, const int32_t *vert_off,
                           const int32_t *face_off, const T *normals, const T *vcol, const T *R, const T *Tr, const T *light,
                           int n_light, const T *obj, const T *bg, int flags, T fov, const T *phong4, const int32_t *fixed_p2f,
                           T *images, int32_t *p2f_out) {
  const bool persp = flags & 2, cull = flags & 1;
  const T s = fov_scale(fov);
  const size_t HW = (size_t)H * W;
  Phong<T> ph = {phong4[0], phong4[1], phong4[2], phong4[3]};
  for (int n = 0; n < B * V; ++n) {
    int b = n / V;
    int vo = vert_off[b], nv = vert_off[b + 1] - vo, fo = face_off[b], nf = face_off[b + 1] - fo;
    const T *Rn = R + 9 * n, *Tn = Tr + 3 * n;
    std::vector<V3<T>> proj(nv);
    for (int i = 0; i < nv; ++i) proj[i] = project_vertex<T>(Rn, Tn, s, ld(verts + 3 * (size_t)(vo + i)));
    std::vector<int32_t> p2f(HW, -1);
    std::vector<T> zb(HW, (T)0);
    const int32_t *fc = faces + 3 * (size_t)fo;
    if (fixed_p2f) {
      std::memcpy(p2f.data(), fixed_p2f + (size_t)n * HW, sizeof(int32_t) * HW);
    } else {
      for (int f = 0; f < nf; ++f) {
        FaceSetup<T> fs = face_setup<T>(proj[fc[3 * f]], proj[fc[3 * f + 1]], proj[fc[3 * f + 2]], cull);
        if (!fs.drawable) continue;
        for (int jy = 0; jy < H; ++jy) {
          T py = pix_to_ndc<T>(jy, H);
          if (py > fs.ymax || py < fs.ymin) continue;
          for (int ix = 0; ix < W; ++ix) {
            T px = pix_to_ndc<T>(ix, W), pz, w0, w1, w2;
            if (!face_test<T>(fs, px, py, persp, &pz, &w0, &w1, &w2)) continue;
            size_t pix = (size_t)(H - 1 - jy) * W + (W - 1 - ix);
            if (p2f[pix] < 0 || pz < zb[pix] || (pz == zb[pix] && f < p2f[pix])) { p2f[pix] = f; zb[pix] = pz; }
          }
        }
      }
    }
    V3<T> cc = camera_center<T>(Rn, Tn);
    V3<T> l = ld(light + 3 * (size_t)(n_light == 1 ? 0 : n));
    V3<T> oc = ld(obj);
    T *img = images + (size_t)n * 3 * HW;
    for (size_t pix = 0; pix < HW; ++pix) {
      int f = p2f[pix];
      if (f < 0) { img[pix] = bg[0]; img[HW + pix] = bg[1]; img[2 * HW + pix] = bg[2]; continue; }
      int row = (int)(pix / W), col = (int)(pix % W);
      int i0 = fc[3 * f], i1 = fc[3 * f + 1], i2 = fc[3 * f + 2];
      FaceSetup<T> fs = face_setup<T>(proj[i0], proj[i1], proj[i2], cull);
      T w0, w1, w2, pz;
      face_bary<T>(fs, pix_to_ndc<T>(W - 1 - col, W), pix_to_ndc<T>(H - 1 - row, H), persp, &w0, &w1, &w2, &pz);
      V3<T> c0 = oc, c1 = oc, c2 = oc;
      if (vcol) { c0 = ld(vcol + 3 * (size_t)(vo + i0)); c1 = ld(vcol + 3 * (size_t)(vo + i1)); c2 = ld(vcol + 3 * (size_t)(vo + i2)); }
      V3<T> rgb = phong_pixel<T>(w0, w1, w2, ld(verts + 3 * (size_t)(vo + i0)), ld(verts + 3 * (size_t)(vo + i1)),
                                 ld(verts + 3 * (size_t)(vo + i2)), ld(normals + 3 * (size_t)(vo + i0)),
                                 ld(normals + 3 * (size_t)(vo + i1)), ld(normals + 3 * (size_t)(vo + i2)), c0, c1, c2, l, cc, ph);
      img[pix] = rgb.x; img[HW + pix] = rgb.y; img[2 * HW + pix] = rgb.z;
    }
    if (p2f_out) std::memcpy(p2f_out + (size_t)n * HW, p2f.data(), sizeof(int32_t) * HW);
  }
}

// mirrors k_mesh_backward + k_mesh_backward_finalize
template <typename T>
static void t_mesh_backward(int B, int V, int H, int W, const T *verts, const int32_t *faces, const int32_t *vert_off,
                            const int32_t *face_off, const T *normals, const T *vcol, const T *R, const T *Tr, const T *light,
                            int n_light, const T *obj, int flags, T fov, const T *phong4, const int32_t *p2f, const T *G,
                            T *gR, T *gT, T *gL) {
  const size_t HW = (size_t)H * W;
  for (int n = 0; n < B * V; ++n) {
    int b = n / V;
    int vo = vert_off[b], fo = face_off[b];
    const int32_t *fc = faces + 3 * (size_t)fo;
    PixelGradIn<T> cam;
    for (int k = 0; k < 9; ++k) cam.R[k] = R[9 * n + k];
    for (int k = 0; k < 3; ++k) cam.Tr[k] = Tr[3 * n + k];
    cam.s = fov_scale(fov);
    cam.l = ld(light + 3 * (size_t)(n_light == 1 ? 0 : n));
    cam.cc = camera_center<T>(cam.R, cam.Tr);
    cam.ph = {phong4[0], phong4[1], phong4[2], phong4[3]};
    cam.persp = flags & 2;
    V3<T> oc = ld(obj);
    T acc[18];
    for (int k = 0; k < 18; ++k) acc[k] = 0;
    for (size_t pix = 0; pix < HW; ++pix) {
      int f = p2f[(size_t)n * HW + pix];
      if (f < 0) continue;
      int row = (int)(pix / W), col = (int)(pix % W);
      int i0 = fc[3 * f], i1 = fc[3 * f + 1], i2 = fc[3 * f + 2];
      V3<T> c0 = oc, c1 = oc, c2 = oc;
      if (vcol) { c0 = ld(vcol + 3 * (size_t)(vo + i0)); c1 = ld(vcol + 3 * (size_t)(vo + i1)); c2 = ld(vcol + 3 * (size_t)(vo + i2)); }
      const T *g = G + (size_t)n * 3 * HW;
      V3<T> Gp = {g[pix], g[HW + pix], g[2 * HW + pix]};
      phong_pixel_bwd<T>(cam, pix_to_ndc<T>(W - 1 - col, W), pix_to_ndc<T>(H - 1 - row, H),
                         ld(verts + 3 * (size_t)(vo + i0)), ld(verts + 3 * (size_t)(vo + i1)), ld(verts + 3 * (size_t)(vo + i2)),
                         ld(normals + 3 * (size_t)(vo + i0)), ld(normals + 3 * (size_t)(vo + i1)), ld(normals + 3 * (size_t)(vo + i2)),
                         c0, c1, c2, Gp, acc);
    }
    fold_camera_center<T>(cam.R, cam.Tr, acc + 12, acc, acc + 9);
    for (int k = 0; k < 9; ++k) gR[9 * n + k] = acc[k];
    for (int k = 0; k < 3; ++k) gT[3 * n + k] = acc[9 + k];
    if (gL) for (int k = 0; k < 3; ++k) gL[3 * n + k] = acc[15 + k];
  }
}

template <typename T>
static void t_camera_position(int n, const T *azim, const T *elev, const T *dist, T *C) {
  for (int i = 0; i < n; ++i) { V3<T> c = camera_position<T>(azim[i], elev[i], dist[i]); C[3 * i] = c.x; C[3 * i + 1] = c.y; C[3 * i + 2] = c.z; }
}
template <typename T>
static void t_camera_position_bwd(int n, const T *azim, const T *elev, const T *dist, const T *g, T *ga, T *ge, T *gd) {
  for (int i = 0; i < n; ++i) camera_position_bwd<T>(azim[i], elev[i], dist[i], ld(g + 3 * i), ga + i, ge + i, gd + i);
}

#define EXPORT(T, SFX)                                                                                                          \
  extern "C" void hm_look_at##SFX(int n, const T *a, const T *e, const T *d, T *R, T *Tr, int32_t *valid) { t_look_at<T>(n, a, e, d, R, Tr, valid); } \
  extern "C" void hm_look_at_bwd##SFX(int n, const T *a, const T *e, const T *d, const T *gR, const T *gT, T *ga, T *ge, T *gd) { \
    t_look_at_bwd<T>(n, a, e, d, gR, gT, ga, ge, gd); }                                                                         \
  extern "C" void hm_camera_position##SFX(int n, const T *a, const T *e, const T *d, T *C) { t_camera_position<T>(n, a, e, d, C); } \
  extern "C" void hm_camera_position_bwd##SFX(int n, const T *a, const T *e, const T *d, const T *g, T *ga, T *ge, T *gd) {     \
    t_camera_position_bwd<T>(n, a, e, d, g, ga, ge, gd); }                                                                      \
  extern "C" void hm_mesh_forward##SFX(int B, int V, int H, int W, const T *verts, const int32_t *faces, const int32_t *vo,     \
                                       const int32_t *fo, const T *normals, const T *vcol, const T *R, const T *Tr,             \
                                       const T *light, int n_light, const T *obj, const T *bg, int flags, T fov,                \
                                       const T *phong4, const int32_t *fixed_p2f, T *images, int32_t *p2f_out) {                \
    t_mesh_forward<T>(B, V, H, W, verts, faces, vo, fo, normals, vcol, R, Tr, light, n_light, obj, bg, flags, fov, phong4,      \
                      fixed_p2f, images, p2f_out); }                                                                            \
  extern "C" void hm_mesh_backward##SFX(int B, int V, int H, int W, const T *verts, const int32_t *faces, const int32_t *vo,    \
                                        const int32_t *fo, const T *normals, const T *vcol, const T *R, const T *Tr,            \
                                        const T *light, int n_light, const T *obj, int flags, T fov, const T *phong4,           \
                                        const int32_t *p2f, const T *G, T *gR, T *gT, T *gL) {                                  \
    t_mesh_backward<T>(B, V, H, W, verts, faces, vo, fo, normals, vcol, R, Tr, light, n_light, obj, flags, fov, phong4, p2f, G, \
                       gR, gT, gL); }

EXPORT(float, _f32)
EXPORT(double, _f64)
