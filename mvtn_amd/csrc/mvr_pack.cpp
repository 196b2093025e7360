// Host half of the mesh hand-over (models/renderer.py:66-77): B CPU meshes -> ONE staging block holding
// what `mvr_mesh_batch` points at, so that a forward needs one H2D copy instead of the reference's 2*B
// `.cuda()` calls.  Plain C++ (g++), linked into libmvr_hip.so; no GPU call in this file.
#include <cstdio>
#include <cstring>

#include "../../include/mvr.h"

extern "C" __attribute__((visibility("hidden"))) void mvr_set_error_(const char *msg);  // mvr_kernels.hip: the thread-local message mvr_last_error() returns

namespace {

constexpr uint64_t kAlign = 256;  // every section starts on a 256-byte boundary (float4 / int4 loads on the device)

inline uint64_t align_up(uint64_t x) { return (x + kAlign - 1) & ~(kAlign - 1); }

int fail(const char *fmt, int a, long long b = 0) {
  char buf[160];
  std::snprintf(buf, sizeof buf, fmt, a, b);
  mvr_set_error_(buf);
  return MVR_ERR_INVALID_ARG;
}

// int64 (the boundary's face dtype) -> int32 with the range check folded in: the OR of all words catches
// negatives and anything >= 2^31, the running maximum of the low words the rest.  Plain loops that
// gcc vectorises; the avx2 clone is picked at load time where the CPU has it.
__attribute__((target_clones("avx2", "default")))
int narrow_checked(const int64_t *__restrict src, int32_t *__restrict dst, int64_t n, int32_t n_verts) {
  int64_t any = 0;
  int32_t mx = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t v = src[i];
    const int32_t lo = (int32_t)v;
    any |= v;
    mx = lo > mx ? lo : mx;
    dst[i] = lo;
  }
  return ((any >> 31) == 0 && mx < n_verts) ? 0 : 1;
}

}  // namespace

extern "C" int mvr_pack_meshes_layout(int32_t n_meshes, const int32_t *n_verts, const int32_t *n_faces,
                                      mvr_pack_layout *layout) {
  if (n_meshes <= 0 || !n_verts || !n_faces || !layout) return fail("mvr_pack_meshes_layout: empty mesh batch (%d)", n_meshes);
  int64_t tv = 0, tf = 0;
  int32_t mv = 0, mf = 0;
  for (int32_t b = 0; b < n_meshes; ++b) {
    if (n_verts[b] <= 0 || n_faces[b] <= 0) return fail("mesh %d: every mesh needs at least one vertex and one face", b);
    tv += n_verts[b];
    tf += n_faces[b];
    mv = n_verts[b] > mv ? n_verts[b] : mv;
    mf = n_faces[b] > mf ? n_faces[b] : mf;
  }
  if (tv > INT32_MAX / 3 || tf > INT32_MAX / 3) return fail("mesh batch too large for 32-bit packed indices (%d meshes)", n_meshes);
  layout->n_meshes = n_meshes;
  layout->total_verts = (int32_t)tv;
  layout->total_faces = (int32_t)tf;
  layout->max_verts = mv;
  layout->max_faces = mf;
  layout->reserved_ = 0;
  layout->verts_at = 0;
  layout->faces_at = align_up((uint64_t)tv * 12);
  layout->vert_off_at = layout->faces_at + align_up((uint64_t)tf * 12);
  layout->face_off_at = layout->vert_off_at + align_up((uint64_t)(n_meshes + 1) * 4);
  layout->bytes = layout->face_off_at + align_up((uint64_t)(n_meshes + 1) * 4);
  return MVR_OK;
}

extern "C" int mvr_pack_meshes_host(const mvr_pack_layout *layout, const float *const *verts, const int32_t *n_verts,
                                    const int64_t *const *faces, const int32_t *n_faces, void *block) {
  if (!layout || !verts || !faces || !n_verts || !n_faces || !block) return fail("mvr_pack_meshes_host: null argument (%d)", 0);
  char *base = static_cast<char *>(block);
  float *pv = reinterpret_cast<float *>(base + layout->verts_at);
  int32_t *pf = reinterpret_cast<int32_t *>(base + layout->faces_at);
  int32_t *vo = reinterpret_cast<int32_t *>(base + layout->vert_off_at);
  int32_t *fo = reinterpret_cast<int32_t *>(base + layout->face_off_at);
  int64_t tv = 0, tf = 0;
  for (int32_t b = 0; b < layout->n_meshes; ++b) {
    if (!verts[b] || !faces[b]) return fail("mesh %d: null vertex or face pointer", b);
    vo[b] = (int32_t)tv;
    fo[b] = (int32_t)tf;
    if (tv + n_verts[b] > layout->total_verts || tf + n_faces[b] > layout->total_faces)
      return fail("mesh %d: sizes do not match the layout", b);
    std::memcpy(pv + tv * 3, verts[b], (size_t)n_verts[b] * 12);
    if (narrow_checked(faces[b], pf + tf * 3, (int64_t)n_faces[b] * 3, n_verts[b]))
      return fail("mesh %d: face index out of range [0, %lld)", b, (long long)n_verts[b]);
    tv += n_verts[b];
    tf += n_faces[b];
  }
  if (tv != layout->total_verts || tf != layout->total_faces) return fail("mvr_pack_meshes_host: sizes do not match the layout (%d)", 0);
  vo[layout->n_meshes] = (int32_t)tv;
  fo[layout->n_meshes] = (int32_t)tf;
  return MVR_OK;
}
