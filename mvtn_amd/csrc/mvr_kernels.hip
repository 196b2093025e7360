// mvr_kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4) + the C ABI of include/mvr.h.
//
// Path: models/renderer.py MVRenderer.forward (reference :173-198) -> render_meshes (:65-114) /
// render_points (:116-151) and their autograd backward.  Design:
//
//   mesh forward = 3 launches on the caller's stream
//     k_look_at            : R, T of all views (+ validity, device-side remedy, relative-light camera positions)
//     k_mesh_prepass       : workgroups [0, B): world-space vertex normals once per MESH (64-bit fixed-point LDS sums);
//                            workgroups [B, B+N): one VIEW each: project the vertices (16 B records, L2 resident), 2-byte
//                            row-bin range per face, per-strip face lists (near faces first), per-view shading constants
//     k_mesh_raster_shade  : one workgroup per (view, strip of 8 rows), heaviest strips first:
//         list -> setup (one lane per face: exact candidate range, signed coefficient record in LDS) -> balanced
//         split of the batch's pixel PAIRS over the lanes -> packed-f32 exact inside test / barycentrics / depth ->
//         64-bit LDS atomicMin on (float_bits(z) << 32 | face) — exactly PyTorch3D's (z, face index) order;
//         far faces are rejected against per-tile depth bounds left by the near ones (hierarchical z);
//         covered pixels are compacted and shaded (Phong + hard blend), planar (B,V,3,H,W) stores.
//     No per-pixel fragment tensors, no bin tables, no V-fold mesh copies ever touch HBM.
//   mesh backward = k_mesh_backward (per covered pixel, 18 partial sums per workgroup) + one finalize launch
//   points        = k_points_project + k_points_raster_shade (hits compacted, 16-byte stores) [+ k_points_raster_topk]
//
// wave = 64 lanes; block = 256 threads (4 waves); strips are full image rows so every store
// instruction writes whole contiguous rows.
//
// Compiled with -ffp-contract=off: coverage/depth arithmetic is the exact IEEE sequence of
// mvr_math.h (shared with nothing outside this library; the CPU oracle restates it independently).

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/mvr.h"
#include "mvr_math.h"

using mvr::V3;
typedef V3<float> f3;

#define MVR_BLOCK 256
#define MVR_WAVES (MVR_BLOCK / 64)
#define MVR_LIST_CAP 256       // LDS face list (4 B entries); a strip touched by more faces is processed in windows of this size
#define MVR_BATCH 256           // faces set up / rasterised per batch (one per lane in the setup phase)
#define MVR_V3_ROUNDS 4          // scan: 8 faces per lane per round -> 256 * 8 * 4 = 8192 faces per pass
#ifndef MVR_RB
#define MVR_RB 256               // threads per rasteriser workgroup
#endif
#define MVR_RW (MVR_RB / 64)
#define MVR_RBATCH MVR_RB        // faces set up per batch: one per lane
#ifndef MVR_PTS_LIST_INTS
#define MVR_PTS_LIST_INTS 2048   // point-kernel hit list (one chunk of 2048 points can hit entirely)
#endif
#ifndef MVR_TILE_SHIFT
#define MVR_TILE_SHIFT 2          // hierarchical-z tiles: 4 columns x the strip's rows (2 / 4 / 8 / 16 columns measured; LDS is at the 4-workgroup limit)
#endif
#define MVR_TILE_SPAN (32 >> MVR_TILE_SHIFT)   // faces wider than 32 columns are not tested
#ifndef MVR_RB_WAVES_PER_SIMD
#define MVR_RB_WAVES_PER_SIMD 4
#endif
#define MVR_KEY_EMPTY 0xFFFFFFFFFFFFFFFFull
#define MVR_NACC 18

static thread_local char g_err[512] = "";

static int set_err(int code, const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
// mvr_pack.cpp (host-only object of the same library) reports through the same message buffer
extern "C" __attribute__((visibility("hidden"))) void mvr_set_error_(const char *msg) { set_err(0, msg); }
static int hip_check(hipError_t e, const char *what) {
  if (e == hipSuccess) return MVR_OK;
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return MVR_ERR_HIP;
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    int _rc = hip_check((expr), #expr);                 \
    if (_rc != MVR_OK) return _rc;                      \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- optional in-library kernel timing (bench.py roofline leg) -------------------------------------
// When enabled, the dominant kernel of each path is bracketed by HIP events recorded on the SAME
// stream it is launched on.  Disabled (the default) nothing below runs in the launch path.
#include <vector>
#include <utility>
enum { PROF_MESH_FWD = 0, PROF_POINTS_FWD = 1, PROF_MESH_BWD = 2, PROF_KINDS = 3 };
static bool g_prof_on = false;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[PROF_KINDS];

struct ProfScope {
  hipStream_t st; int kind; hipEvent_t stop; bool on;
  ProfScope(int kind_, hipStream_t st_) : st(st_), kind(kind_), stop(nullptr), on(g_prof_on) {
    if (!on) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(a, st);
    stop = b;
    g_prof_ev[kind].push_back(std::make_pair(a, b));
  }
  ~ProfScope() { if (on) (void)hipEventRecord(stop, st); }
};

// ======================================================================================
// cameras
// ======================================================================================

// counter-based uniform [0,1) (splitmix64 finaliser): no state, reproducible from (seed, trial, view, k)
__device__ __forceinline__ float rng_uniform(unsigned long long seed, unsigned trial, unsigned i, unsigned k) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)trial << 40) ^ ((unsigned long long)i << 2) ^ (unsigned long long)k);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// Single workgroup (n = B*V is hundreds to thousands).  Fuses look_at_view_transform with
// check_valid_rotation_matrix (util.py:403-420) and, when max_trials >= 0, with the remedy loop of
// check_and_correct_rotation_matrix (ops.py:156-165): while ANY view is invalid, ALL views are re-drawn
// as look_at(azim + 180 U, elev + 90 U') from the ORIGINAL angles; the reference gives up
// (sys.exit) once a re-draw numbered > max_trials is still invalid.  Running this on the device removes
// the host synchronisation the reference pays on every forward.
//   status[0] = re-draws performed, status[1] = views still invalid at the end.
#define MVR_LA_BLOCK 1024
__global__ void __launch_bounds__(MVR_LA_BLOCK) k_look_at(const float *__restrict__ azim, const float *__restrict__ elev,
                                                          const float *__restrict__ dist, int n, int max_trials,
                                                          unsigned long long seed, float *__restrict__ R,
                                                          float *__restrict__ T, float *__restrict__ azim_used,
                                                          float *__restrict__ elev_used, int32_t *__restrict__ view_valid,
                                                          int32_t *__restrict__ status, float *__restrict__ cam_pos = nullptr) {
  __shared__ int s_bad;
  for (int trial = 0;; ++trial) {
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    int bad = 0;
    // TWO lanes per view: the f64 sin/cos of the elevation (even lane) and of the azimuth (odd lane) are the long
    // dependency chains of a look-at; side by side they take half the time.  The even lane finishes the view.
    for (int base = 0; base < n; base += MVR_LA_BLOCK / 2) {
      const int i = base + ((int)threadIdx.x >> 1);
      const bool live = i < n, role_az = (threadIdx.x & 1) != 0;
      float a = live ? azim[i] : 0.f, e = live ? elev[i] : 0.f;
      if (trial > 0) {
        a = a + 180.0f * rng_uniform(seed, trial, i, 0);
        e = e + 90.0f * rng_uniform(seed, trial, i, 1);
      }
      float sn, cs;
      mvr::sincos_deg<float>(role_az ? a : e, &sn, &cs);
      const float sa = __shfl_xor(sn, 1), ca = __shfl_xor(cs, 1);   // even lanes receive the azimuth's pair
      if (!live || role_az) continue;
      const f3 C = mvr::camera_position_sc<float>(sn, cs, sa, ca, dist[i]);
      if (trial == 0 && cam_pos) {   // camera position of the REQUESTED angles: the reference's "relative" light (:168-169)
        cam_pos[3 * i] = C.x; cam_pos[3 * i + 1] = C.y; cam_pos[3 * i + 2] = C.z;
      }
      float Rm[9], Tv[3];
      mvr::look_at_from_position<float>(C, Rm, Tv);
#pragma unroll
      for (int k = 0; k < 9; ++k) R[9 * i + k] = Rm[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) T[3 * i + k] = Tv[k];
      if (azim_used) azim_used[i] = a;
      if (elev_used) elev_used[i] = e;
      const bool ok = mvr::rotation_valid<float>(Rm, 1e-6f);
      if (view_valid) view_valid[i] = ok ? 1 : 0;
      bad += ok ? 0 : 1;
    }
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    const int nb = s_bad;
    __syncthreads();
    if (nb == 0 || max_trials < 0 || trial > max_trials) {
      if (threadIdx.x == 0 && status) { status[0] = trial; status[1] = nb; }
      break;
    }
  }
}

__global__ void __launch_bounds__(MVR_BLOCK) k_look_at_bwd(const float *__restrict__ azim, const float *__restrict__ elev,
                                                           const float *__restrict__ dist, int n,
                                                           const float *__restrict__ gR, const float *__restrict__ gT,
                                                           float *__restrict__ g_azim, float *__restrict__ g_elev,
                                                           float *__restrict__ g_dist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // tiny (n values): evaluate the chain in double, round once
  double gr[9], gt[3];
  for (int k = 0; k < 9; ++k) gr[k] = gR ? (double)gR[9 * i + k] : 0.0;
  for (int k = 0; k < 3; ++k) gt[k] = gT ? (double)gT[3 * i + k] : 0.0;
  double ga, ge, gd;
  mvr::look_at_bwd<double>((double)azim[i], (double)elev[i], (double)dist[i], gr, gt, &ga, &ge, &gd);
  g_azim[i] = (float)ga;
  g_elev[i] = (float)ge;
  g_dist[i] = (float)gd;
}

__global__ void __launch_bounds__(MVR_BLOCK) k_camera_position(const float *__restrict__ azim, const float *__restrict__ elev,
                                                               const float *__restrict__ dist, int n, float *__restrict__ pos) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f3 c = mvr::camera_position<float>(azim[i], elev[i], dist[i]);
  pos[3 * i] = c.x; pos[3 * i + 1] = c.y; pos[3 * i + 2] = c.z;
}

__global__ void __launch_bounds__(MVR_BLOCK) k_camera_position_bwd(const float *__restrict__ azim, const float *__restrict__ elev,
                                                                   const float *__restrict__ dist, int n,
                                                                   const float *__restrict__ gpos, float *__restrict__ g_azim,
                                                                   float *__restrict__ g_elev, float *__restrict__ g_dist) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  V3<double> g = {(double)gpos[3 * i], (double)gpos[3 * i + 1], (double)gpos[3 * i + 2]};
  double ga, ge, gd;
  mvr::camera_position_bwd<double>((double)azim[i], (double)elev[i], (double)dist[i], g, &ga, &ge, &gd);
  g_azim[i] = (float)ga; g_elev[i] = (float)ge; g_dist[i] = (float)gd;
}

// ======================================================================================
// mesh: vertex attributes (position + normal, 2 x 16 B per vertex)
// ======================================================================================

struct MeshPtrs {
  const float *verts;
  const int32_t *faces;
  const int32_t *vert_off;
  const int32_t *face_off;
  const float *vert_colors;
  int B;
};

// grid (ceil(max_faces / BLOCK), B): one lane per face; area-weighted face normal
// n_f = cross(v2 - v1, v0 - v1) added to its three vertices (Meshes.verts_normals_packed).
// The sum is accumulated in 64-bit fixed point (2^-32 units): integer adds commute, so the
// result is bit-reproducible run to run (float atomics would depend on arrival order).
#define MVR_FIX_SCALE 4294967296.0f
__global__ void __launch_bounds__(MVR_BLOCK) k_normals_accum(MeshPtrs m, long long *__restrict__ nacc) {
  int b = blockIdx.y;
  int f0 = m.face_off[b], nf = m.face_off[b + 1] - f0, v0 = m.vert_off[b];
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const int32_t *fc = m.faces + 3 * (size_t)(f0 + f);
  int i0 = fc[0], i1 = fc[1], i2 = fc[2];
  const float *pa = m.verts + 3 * (size_t)(v0 + i0), *pb = m.verts + 3 * (size_t)(v0 + i1), *pc = m.verts + 3 * (size_t)(v0 + i2);
  f3 e0 = {pc[0] - pb[0], pc[1] - pb[1], pc[2] - pb[2]};
  f3 e1 = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
  f3 n = mvr::cross(e0, e1);
  // clamp keeps the conversion defined for absurdly large meshes (|n| < 2^30 after scaling by 2^32 needs |n| < 2^-2 * 2^32)
  const float lim = 1.0e9f;
  long long qx = __float2ll_rn(fminf(fmaxf(n.x, -lim), lim) * MVR_FIX_SCALE);
  long long qy = __float2ll_rn(fminf(fmaxf(n.y, -lim), lim) * MVR_FIX_SCALE);
  long long qz = __float2ll_rn(fminf(fmaxf(n.z, -lim), lim) * MVR_FIX_SCALE);
  unsigned long long *d0 = (unsigned long long *)(nacc + 3 * (size_t)(v0 + i0));
  unsigned long long *d1 = (unsigned long long *)(nacc + 3 * (size_t)(v0 + i1));
  unsigned long long *d2 = (unsigned long long *)(nacc + 3 * (size_t)(v0 + i2));
  atomicAdd(d0, (unsigned long long)qx); atomicAdd(d0 + 1, (unsigned long long)qy); atomicAdd(d0 + 2, (unsigned long long)qz);
  atomicAdd(d1, (unsigned long long)qx); atomicAdd(d1 + 1, (unsigned long long)qy); atomicAdd(d1 + 2, (unsigned long long)qz);
  atomicAdd(d2, (unsigned long long)qx); atomicAdd(d2 + 1, (unsigned long long)qy); atomicAdd(d2 + 2, (unsigned long long)qz);
}

// Whole mesh in ONE workgroup when its 3 x int64 per-vertex accumulators fit in LDS (<= MVR_NRM_LDS_VERTS
// vertices): faces -> 64-bit LDS atomics -> barrier -> normalise -> vattr.  One launch, no memset, no global
// atomics; same fixed-point sum as the 3-step global path, hence the same bits.
#define MVR_NRM_BLOCK 1024
#define MVR_NRM_LDS_VERTS 6400   // 6400 * 24 B = 150 KiB
__device__ __forceinline__ void vattr_mesh_lds_body(const MeshPtrs &m, int b, int chunk, unsigned char *smem,
                                                    float4 *__restrict__ vattr, float *__restrict__ normals_out) {
  // the workgroup owns the vertices [vlo, vhi) of mesh b (a mesh with more than MVR_NRM_LDS_VERTS vertices is
  // covered by several workgroups, each scanning all faces and keeping only the corners that fall in its range)
  unsigned long long *acc = (unsigned long long *)smem;
  const int f0 = m.face_off[b], nf = m.face_off[b + 1] - f0, v0 = m.vert_off[b], nv = m.vert_off[b + 1] - v0;
  const int vlo = chunk * MVR_NRM_LDS_VERTS, vhi = min(nv, vlo + MVR_NRM_LDS_VERTS);
  if (vlo >= nv) return;   // workgroup-uniform
  const int nloc = vhi - vlo;
  for (int i = threadIdx.x; i < 3 * nloc; i += MVR_NRM_BLOCK) acc[i] = 0ull;
  __syncthreads();
  const float lim = 1.0e9f;
  for (int f = threadIdx.x; f < nf; f += MVR_NRM_BLOCK) {
    const int32_t *fc = m.faces + 3 * (size_t)(f0 + f);
    int i0 = fc[0], i1 = fc[1], i2 = fc[2];
    const bool in0 = i0 >= vlo && i0 < vhi, in1 = i1 >= vlo && i1 < vhi, in2 = i2 >= vlo && i2 < vhi;
    if (!(in0 || in1 || in2)) continue;
    const float *pa = m.verts + 3 * (size_t)(v0 + i0), *pb = m.verts + 3 * (size_t)(v0 + i1), *pc = m.verts + 3 * (size_t)(v0 + i2);
    f3 e0 = {pc[0] - pb[0], pc[1] - pb[1], pc[2] - pb[2]};
    f3 e1 = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    f3 n = mvr::cross(e0, e1);
    unsigned long long qx = (unsigned long long)__float2ll_rn(fminf(fmaxf(n.x, -lim), lim) * MVR_FIX_SCALE);
    unsigned long long qy = (unsigned long long)__float2ll_rn(fminf(fmaxf(n.y, -lim), lim) * MVR_FIX_SCALE);
    unsigned long long qz = (unsigned long long)__float2ll_rn(fminf(fmaxf(n.z, -lim), lim) * MVR_FIX_SCALE);
    if (in0) { atomicAdd(acc + 3 * (i0 - vlo), qx); atomicAdd(acc + 3 * (i0 - vlo) + 1, qy); atomicAdd(acc + 3 * (i0 - vlo) + 2, qz); }
    if (in1) { atomicAdd(acc + 3 * (i1 - vlo), qx); atomicAdd(acc + 3 * (i1 - vlo) + 1, qy); atomicAdd(acc + 3 * (i1 - vlo) + 2, qz); }
    if (in2) { atomicAdd(acc + 3 * (i2 - vlo), qx); atomicAdd(acc + 3 * (i2 - vlo) + 1, qy); atomicAdd(acc + 3 * (i2 - vlo) + 2, qz); }
  }
  __syncthreads();
  const double inv = 1.0 / 4294967296.0;
  for (int i = threadIdx.x; i < nloc; i += MVR_NRM_BLOCK) {
    f3 a = {(float)((double)(long long)acc[3 * i] * inv), (float)((double)(long long)acc[3 * i + 1] * inv),
            (float)((double)(long long)acc[3 * i + 2] * inv)};
    f3 n = mvr::normalize<float>(a, mvr::kNormalizeEps);
    size_t v = (size_t)v0 + vlo + i;
    vattr[2 * v] = make_float4(m.verts[3 * v], m.verts[3 * v + 1], m.verts[3 * v + 2], 0.f);
    vattr[2 * v + 1] = make_float4(n.x, n.y, n.z, 0.f);
    if (normals_out) { normals_out[3 * v] = n.x; normals_out[3 * v + 1] = n.y; normals_out[3 * v + 2] = n.z; }
  }
}

__global__ void __launch_bounds__(MVR_NRM_BLOCK) k_vattr_mesh_lds(MeshPtrs m, int chunks, float4 *__restrict__ vattr,
                                                                  float *__restrict__ normals_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  vattr_mesh_lds_body(m, blockIdx.x / chunks, blockIdx.x % chunks, smem, vattr, normals_out);
}

// one lane per vertex: normalise the accumulated normal (eps 1e-6) and store the position beside it.
__global__ void __launch_bounds__(MVR_BLOCK) k_vattr_finalize(const float *__restrict__ verts, int total_verts,
                                                              const long long *__restrict__ nacc,
                                                              float4 *__restrict__ vattr, float *__restrict__ normals_out) {
  int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= total_verts) return;
  const double inv = 1.0 / 4294967296.0;
  f3 a = {(float)((double)nacc[3 * (size_t)v] * inv), (float)((double)nacc[3 * (size_t)v + 1] * inv),
          (float)((double)nacc[3 * (size_t)v + 2] * inv)};
  f3 n = mvr::normalize<float>(a, mvr::kNormalizeEps);
  vattr[2 * (size_t)v] = make_float4(verts[3 * (size_t)v], verts[3 * (size_t)v + 1], verts[3 * (size_t)v + 2], 0.f);
  vattr[2 * (size_t)v + 1] = make_float4(n.x, n.y, n.z, 0.f);
  if (normals_out) {
    normals_out[3 * (size_t)v] = n.x; normals_out[3 * (size_t)v + 1] = n.y; normals_out[3 * (size_t)v + 2] = n.z;
  }
}

// ======================================================================================
// mesh: projection of every (view, vertex): proj[V * vert_off[b] + v * Vn_b + i]
// ======================================================================================

// grid (ceil(max_verts / BLOCK), N)
__global__ void __launch_bounds__(MVR_BLOCK) k_project_verts(MeshPtrs m, int V, const float *__restrict__ R,
                                                             const float *__restrict__ T, float s, float4 *__restrict__ proj) {
  int n = blockIdx.y;
  int b = n / V, v = n - b * V;
  int vo = m.vert_off[b], nv = m.vert_off[b + 1] - vo;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  float Rm[9], Tv[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rm[k] = R[9 * (size_t)n + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) Tv[k] = T[3 * (size_t)n + k];
  const float *p = m.verts + 3 * (size_t)(vo + i);
  f3 q = mvr::project_vertex<float>(Rm, Tv, s, {p[0], p[1], p[2]});
  proj[(size_t)V * vo + (size_t)v * nv + i] = make_float4(q.x, q.y, q.z, 0.f);
}

// ======================================================================================
// mesh: rasterise + shade, one workgroup per (view, strip)
// ======================================================================================

#ifdef MVR_DIAG
// diagnostic build only (tools/diag_phases.py): per-workgroup phase timestamps, written to a buffer of
// their own; never compiled into libmvr_hip.so.
__device__ unsigned long long *g_diag = nullptr;
#ifdef MVR_DIAG_WALL   // chip-wide 100 MHz clock: comparable across CUs (tools/diag_timeline.py)
#define DIAG_STAMP(slot) do { if (threadIdx.x == 0 && g_diag) g_diag[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define DIAG_STAMP(slot) do { if (threadIdx.x == 0 && g_diag) g_diag[(size_t)blockIdx.x * 8 + (slot)] = clock64(); } while (0)
#endif
#define DIAG_PP(view, slot) do { if (threadIdx.x == 0 && g_diag) g_diag[(size_t)(1 << 20) + (size_t)(view) * 8 + (slot)] = clock64(); } while (0)
#define DIAG_ADD(slot, v) do { if (threadIdx.x == 0 && g_diag) g_diag[(size_t)blockIdx.x * 8 + (slot)] += (v); } while (0)
#else
#define DIAG_STAMP(slot)
#define DIAG_PP(view, slot)
#define DIAG_ADD(slot, v)
#endif

__device__ __forceinline__ f3 ld3(const float4 *p) { float4 t = *p; return {t.x, t.y, t.z}; }

// Pixel-index range (NDC index space, i = S-1-col) whose centres c_i = -1 + (2i+1)/S can satisfy
// lo <= c_i <= hi.  c_i >= lo  <=>  i >= (lo+1) S/2 - 1/2 in exact arithmetic; the float evaluation of
// that bound and of c_i itself are each off by < S * 2.4e-7 pixel, so a margin of 1e-3 + 2e-6 S pixel
// keeps the range conservative (never drops a pixel the exact bbox test would accept) while adding
// at most one extra column/row.  The exact IEEE bbox test still runs per pixel.
__device__ __forceinline__ void ndc_range(float lo, float hi, int S, int *i0, int *i1) {
  const float m = 1e-3f + 2e-6f * (float)S;
  float a = (lo + 1.0f) * (0.5f * (float)S) - 0.5f;
  float b = (hi + 1.0f) * (0.5f * (float)S) - 0.5f;
  a = fminf(fmaxf(a, -2.0f), (float)S + 2.0f);
  b = fminf(fmaxf(b, -2.0f), (float)S + 2.0f);
  int x0 = (int)ceilf(a - m), x1 = (int)floorf(b + m);
  *i0 = x0 < 0 ? 0 : x0;
  *i1 = x1 > S - 1 ? S - 1 : x1;
}

// ======================================================================================
// mesh: per (view, face) strip range — the rasteriser's scan reads 2 bytes per face instead of
// gathering 3 indices + 3 projected vertices per face per strip
// ======================================================================================

#define MVR_BIN_NONE 0xFFFFu   // face can touch no pixel of this view (culled / off-screen)

// Row-bin range (g0 | g1 << 8) of one projected face, MVR_BIN_NONE if it can touch no pixel: the per-face culls of
// mvr::face_setup (without its reciprocal) + the conservative pixel range of its bounding box.
__device__ __forceinline__ unsigned short face_bin_range(const f3 &v0, const f3 &v1, const f3 &v2, int H, int W, int G,
                                                         int cull_backfaces, float px_lo, float px_hi, float py_lo, float py_hi) {
  const float xmin = fminf(fminf(v0.x, v1.x), v2.x), xmax = fmaxf(fmaxf(v0.x, v1.x), v2.x);
  const float ymin = fminf(fminf(v0.y, v1.y), v2.y), ymax = fmaxf(fmaxf(v0.y, v1.y), v2.y);
  const float zmin = fminf(fminf(v0.z, v1.z), v2.z), zmax = fmaxf(fmaxf(v0.z, v1.z), v2.z);
  const float face_area = mvr::edge_fn(v0.x, v0.y, v1.x, v1.y, v2.x, v2.y);
  const bool zero_area = (face_area <= mvr::kEps && face_area >= -mvr::kEps);
  bool ok = !(zmax < 0.f || (cull_backfaces && face_area < 0.f) || zmin < mvr::kEps || zero_area);
  // NaN coordinates fail every comparison below -> not drawable (the exact test would reject every pixel too)
  ok = ok && xmax >= px_lo && xmin <= px_hi && ymax >= py_lo && ymin <= py_hi;
  unsigned short out = MVR_BIN_NONE;
  if (ok) {
    // Only the rows matter for the bins.  (A face that slips between two pixel-centre columns is still listed; the
    // rasteriser's setup finds no candidate pixel for it.  Not computing the column range saves ~20 instructions per
    // face of every view in a pre-pass that is bound by VALU issue on its one CU per view.)
    int y0, y1;
    ndc_range(ymin, ymax, H, &y0, &y1);
    if (y0 <= y1) {
      // NDC row index jy <-> image row H-1-jy; rows < 2^15, so the float quotient settles with one correction
      // (a 32-bit integer division is ~25 VALU instructions, twice per face of every view)
      const float rG = __builtin_amdgcn_rcpf((float)G);
      int g0 = (int)((float)(H - 1 - y1) * rG), g1 = (int)((float)(H - 1 - y0) * rG);
      g0 += ((H - 1 - y1) - g0 * G >= G ? 1 : 0) - ((H - 1 - y1) - g0 * G < 0 ? 1 : 0);
      g1 += ((H - 1 - y0) - g1 * G >= G ? 1 : 0) - ((H - 1 - y0) - g1 * G < 0 ? 1 : 0);
      out = (unsigned short)(g0 | (g1 << 8));
    }
  }
  return out;
}

// grid (ceil(max_faces / BLOCK), N).  bins[n * stride + f] = g0 | (g1 << 8): first / last row-bin (G image rows
// each, G a multiple of the strip height chosen so that there are <= 254 bins) that the face's bounding box
// can reach; conservative (ndc_range).  Rows of the table are padded to `stride` (multiple of 8) entries so
// that the rasteriser reads them as aligned 16-byte vectors.
__global__ void __launch_bounds__(MVR_BLOCK) k_face_bins(MeshPtrs m, int V, int H, int W, int G, int stride, int cull_backfaces,
                                                         const float4 *__restrict__ proj_all, unsigned short *__restrict__ bins) {
  const int n = blockIdx.y;
  const int b = n / V, v = n - b * V;
  const int vo = m.vert_off[b], nv = m.vert_off[b + 1] - vo;
  const int fo = m.face_off[b], nf = m.face_off[b + 1] - fo;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  (void)fo;
  const float4 *__restrict__ proj = proj_all + (size_t)V * vo + (size_t)v * nv;
  const int32_t *__restrict__ fc = m.faces + 3 * (size_t)(fo + f);
  const f3 v0 = ld3(proj + fc[0]), v1 = ld3(proj + fc[1]), v2 = ld3(proj + fc[2]);
  const unsigned short out = face_bin_range(v0, v1, v2, H, W, G, cull_backfaces, mvr::pix_to_ndc<float>(0, W),
                                            mvr::pix_to_ndc<float>(W - 1, W), mvr::pix_to_ndc<float>(0, H),
                                            mvr::pix_to_ndc<float>(H - 1, H));
  bins[(size_t)n * stride + f] = out;
}

// One 1024-thread workgroup per VIEW when the mesh's projected vertices fit in LDS (12 B each): project every
// vertex (-> LDS and the global `proj` table the rasteriser gathers from), barrier, then the per-face row-bin
// ranges straight from LDS.  Same arithmetic as k_project_verts + k_face_bins, one launch instead of two and no
// global gathers for the bins.
#define MVR_PB_BLOCK 1024
#ifndef MVR_PB_LDS_VERTS
#define MVR_PB_LDS_VERTS 12288   // 12288 * 12 B = 144 KiB
#endif
// Per-(view, row-bin) face lists (optional, `sl.off != nullptr`): off[n][g] .. off[n][g + 1] delimit, inside
// list[n * cap ...], the faces whose bounding box can reach row-bin g — the rasteriser then reads its strip's faces
// directly instead of scanning every face's 2-byte range.  The order inside a list depends on LDS atomics and may
// vary between runs; the depth resolve is an order-independent min, so the images do not.  A view whose lists would
// exceed `cap` entries gets off[n][n_bins] = -1 and the rasteriser falls back to the scan.
// Inside a list the faces NEARER than the object's centre (min z < z of the world origin in view space) come first and
// front[n][g] counts them: the rasteriser draws them first and then rejects whole far faces against the depth they left
// (hierarchical z; only an ordering heuristic — any order gives the same image).
struct StripLists { int *off; int4 *list; int *front; int cap, n_bins; };   // list entries: (face, its 3 vertex indices)
// Per-view shading constants written by the pre-pass (8 floats per view: normalised light direction, camera centre,
// 1 / projection scale): uniform data the rasteriser would otherwise recompute in every lane of every workgroup.
struct ViewConsts { float *out; const float *light; int n_light; };
// Optional fusion of k_look_at into the pre-pass (mvr_render_mesh_views_forward): every view workgroup evaluates the
// look-at of ALL views (one per thread) so that it knows, like the single-workgroup kernel, whether the remedy has to
// re-draw — the counter-based RNG makes every workgroup take the same decisions — and then keeps its own view's R, T.
// Same device functions, same results as k_look_at; one launch (and its ~5 us) less per forward.
struct LookAtFuse {
  const float *azim, *elev, *dist;   // nullptr: R, T are inputs
  int max_trials; unsigned long long seed;
  float *R, *T, *azim_used, *elev_used, *cam_pos;   // outputs (this workgroup writes row n)
  int32_t *view_valid, *status;
  int N;
};

// The look-at of a whole batch of views evaluated by ONE workgroup of MVR_PB_BLOCK threads (every view workgroup of a
// fused pre-pass runs it: the remedy of ops.py:156-165 re-draws ALL views when any is invalid, so each workgroup needs
// the global verdict, and 512 views per pass cost no more than one).  Publishes view n's R, T in s_RT (and in la.R /
// la.T, with the angles used, the validity flag and, on the first trial, the camera position); a workgroup that owns
// n_own consecutive views gets them at s_RT + 12 k.
__device__ __forceinline__ void look_at_views_block(const LookAtFuse &la, int n, float *s_RT, int *s_bad_p, int n_own = 1) {
  int &s_bad = *s_bad_p;
  for (int trial = 0;; ++trial) {
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    int bad = 0;
    // TWO lanes per view: the f64 sin/cos of the elevation (even lane) and of the azimuth (odd lane) are the long
    // dependency chains of a look-at; side by side they take half the time.  The even lane finishes the view.
    for (int base = 0; base < la.N; base += MVR_PB_BLOCK / 2) {
      const int i = base + ((int)threadIdx.x >> 1);
      const bool live = i < la.N, role_az = (threadIdx.x & 1) != 0;
      float a = live ? la.azim[i] : 0.f, e = live ? la.elev[i] : 0.f;
      if (trial > 0) {
        a = a + 180.0f * rng_uniform(la.seed, trial, i, 0);
        e = e + 90.0f * rng_uniform(la.seed, trial, i, 1);
      }
      float sn, cs;
      mvr::sincos_deg<float>(role_az ? a : e, &sn, &cs);
      const float sa = __shfl_xor(sn, 1), ca = __shfl_xor(cs, 1);   // even lanes receive the azimuth's pair
      if (!live || role_az) continue;
      float Ri[9], Ti[3];
      mvr::look_at_from_position<float>(mvr::camera_position_sc<float>(sn, cs, sa, ca, la.dist[i]), Ri, Ti);
      const bool ok = mvr::rotation_valid<float>(Ri, 1e-6f);
      bad += ok ? 0 : 1;
      if (i >= n && i < n + n_own) {   // this workgroup's view(s): publish (the last trial's values stay)
        float *rt = s_RT + 12 * (i - n);
#pragma unroll
        for (int k = 0; k < 9; ++k) { rt[k] = Ri[k]; la.R[9 * (size_t)i + k] = Ri[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k) { rt[9 + k] = Ti[k]; la.T[3 * (size_t)i + k] = Ti[k]; }
        if (la.azim_used) la.azim_used[i] = a;
        if (la.elev_used) la.elev_used[i] = e;
        if (la.view_valid) la.view_valid[i] = ok ? 1 : 0;
        if (trial == 0 && la.cam_pos) {   // relative light: camera position of the REQUESTED angles (= this trial's)
          const f3 c = mvr::camera_position_sc<float>(sn, cs, sa, ca, la.dist[i]);
          la.cam_pos[3 * i] = c.x; la.cam_pos[3 * i + 1] = c.y; la.cam_pos[3 * i + 2] = c.z;
        }
      }
    }
    if (bad) atomicAdd(&s_bad, bad);
    __syncthreads();
    const int nb = s_bad;
    __syncthreads();
    if (nb == 0 || la.max_trials < 0 || trial > la.max_trials) {
      if (n == 0 && threadIdx.x == 0 && la.status) { la.status[0] = trial; la.status[1] = nb; }
      break;
    }
  }
}

__device__ __forceinline__ void project_bins_view_body(const MeshPtrs &m, int n, unsigned char *smem, int V, int H, int W, int G,
                                                       int stride, int cull_backfaces, const float *__restrict__ R,
                                                       const float *__restrict__ T, float s, float4 *__restrict__ proj_all,
                                                       unsigned short *__restrict__ bins, const StripLists sl = StripLists{nullptr, nullptr, nullptr, 0, 0},
                                                       const ViewConsts vc = ViewConsts{nullptr, nullptr, 0},
                                                       const LookAtFuse la = LookAtFuse{nullptr, nullptr, nullptr, 0, 0ull, nullptr, nullptr,
                                                                                        nullptr, nullptr, nullptr, nullptr, nullptr, 0}) {
  __shared__ int s_hist[512], s_cur[512];   // [2 g] near faces of row-bin g, [2 g + 1] far faces: ONE atomic per (face, bin)
  __shared__ float s_RT[12];
  __shared__ int s_bad;
  float *pl = (float *)smem;   // 3 * nv
  DIAG_PP(n, 0);
  if (sl.off && threadIdx.x < 512) s_hist[threadIdx.x] = 0;
  const int b = n / V, v = n - b * V;
  const int vo = m.vert_off[b], nv = m.vert_off[b + 1] - vo;
  const int fo = m.face_off[b], nf = m.face_off[b + 1] - fo;
  float Rm[9], Tv[3];
  if (la.azim) {
    look_at_views_block(la, n, s_RT, &s_bad);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rm[k] = s_RT[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) Tv[k] = s_RT[9 + k];
    DIAG_PP(n, 5);
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) Rm[k] = R[9 * (size_t)n + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) Tv[k] = T[3 * (size_t)n + k];
  }
  float4 *__restrict__ proj = proj_all + (size_t)V * vo + (size_t)v * nv;
  if (vc.out && threadIdx.x == 0) {
    f3 lraw;
    if (la.azim && la.cam_pos) lraw = mvr::camera_position<float>(la.azim[n], la.elev[n], la.dist[n]);   // = cam_pos[n], written above by another thread
    else { const float *lp = vc.light + 3 * (size_t)(vc.n_light == 1 ? 0 : n); lraw = {lp[0], lp[1], lp[2]}; }
    const f3 lh = mvr::normalize<float>(lraw, mvr::kNormalizeEps);
    const f3 cc = mvr::camera_center<float>(Rm, Tv);
    float *o = vc.out + 8 * (size_t)n;
    o[0] = lh.x; o[1] = lh.y; o[2] = lh.z; o[3] = cc.x; o[4] = cc.y; o[5] = cc.z; o[6] = 1.0f / s; o[7] = 0.f;
  }
  for (int i = threadIdx.x; i < nv; i += MVR_PB_BLOCK) {
    const float *p = m.verts + 3 * (size_t)(vo + i);
    const f3 q = mvr::project_vertex<float>(Rm, Tv, s, {p[0], p[1], p[2]});
    pl[3 * i] = q.x; pl[3 * i + 1] = q.y; pl[3 * i + 2] = q.z;
    proj[i] = make_float4(q.x, q.y, q.z, 0.f);
  }
  __syncthreads();
  DIAG_PP(n, 1);
  const float px_lo = mvr::pix_to_ndc<float>(0, W), px_hi = mvr::pix_to_ndc<float>(W - 1, W);
  const float py_lo = mvr::pix_to_ndc<float>(0, H), py_hi = mvr::pix_to_ndc<float>(H - 1, H);
  // the first MVR_PB_KEEP faces of every thread (all of them up to 8192 faces per mesh) keep their bin range, near
  // flag and vertex indices in registers for the fill pass below; later ones are re-derived there
  constexpr int MVR_PB_KEEP = 8;
  unsigned keep_e[MVR_PB_KEEP];   // bin range | near << 16
  int keep_i[MVR_PB_KEEP][3];
#pragma unroll
  for (int k = 0; k < MVR_PB_KEEP; ++k) keep_e[k] = MVR_BIN_NONE;
  auto bin_face = [&](int f, int &i0, int &i1, int &i2) -> unsigned {
    const int32_t *__restrict__ fc = m.faces + 3 * (size_t)(fo + f);
    i0 = fc[0]; i1 = fc[1]; i2 = fc[2];
    const f3 v0 = {pl[3 * i0], pl[3 * i0 + 1], pl[3 * i0 + 2]};
    const f3 v1 = {pl[3 * i1], pl[3 * i1 + 1], pl[3 * i1 + 2]};
    const f3 v2 = {pl[3 * i2], pl[3 * i2 + 1], pl[3 * i2 + 2]};
    const unsigned e = face_bin_range(v0, v1, v2, H, W, G, cull_backfaces, px_lo, px_hi, py_lo, py_hi);
    const bool near_face = fminf(fminf(v0.z, v1.z), v2.z) < Tv[2];
    return e | (near_face ? 0x10000u : 0u);
  };
#pragma unroll
  for (int k = 0; k < MVR_PB_KEEP; ++k) {
    const int f = (int)threadIdx.x + k * MVR_PB_BLOCK;
    if (f < nf) {
      const unsigned en = bin_face(f, keep_i[k][0], keep_i[k][1], keep_i[k][2]);
      keep_e[k] = en;
      const unsigned e = en & 0xffffu;
      bins[(size_t)n * stride + f] = (unsigned short)e;
#ifndef MVR_ABL_PP_NO_HIST
      if (sl.off && e != MVR_BIN_NONE)
        for (int g = e & 0xff; g <= (int)(e >> 8); ++g) atomicAdd(&s_hist[2 * g + ((en >> 16) ? 0 : 1)], 1);
#endif
    }
  }
  for (int f = (int)threadIdx.x + MVR_PB_KEEP * MVR_PB_BLOCK; f < nf; f += MVR_PB_BLOCK) {
    int i0, i1, i2;
    const unsigned en = bin_face(f, i0, i1, i2);
    const unsigned e = en & 0xffffu;
    bins[(size_t)n * stride + f] = (unsigned short)e;
    if (sl.off && e != MVR_BIN_NONE)
      for (int g = e & 0xff; g <= (int)(e >> 8); ++g) atomicAdd(&s_hist[2 * g + ((en >> 16) ? 0 : 1)], 1);
  }
  if (!sl.off) return;
  __syncthreads();
  DIAG_PP(n, 2);
  int *off = sl.off + (size_t)n * (sl.n_bins + 1);
  if (threadIdx.x < 64) {   // exclusive scan of the <= 254 bin counts by one wave (4 per lane)
    const int l = threadIdx.x;
    int c[4], tot = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = (4 * l + k < sl.n_bins) ? s_hist[8 * l + 2 * k] + s_hist[8 * l + 2 * k + 1] : 0; tot += c[k]; }
    int incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (l >= o) incl += t;
    }
    const int total = __shfl(incl, 63);
    const bool fits = total <= sl.cap;
    int run = incl - tot;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (4 * l + k < sl.n_bins) {
        const int nn = s_hist[8 * l + 2 * k];
        s_cur[8 * l + 2 * k] = fits ? run : -1; s_cur[8 * l + 2 * k + 1] = run + nn; off[4 * l + k] = run;
        sl.front[(size_t)n * sl.n_bins + 4 * l + k] = nn;
      }
      run += c[k];
    }
    if (l == 0) off[sl.n_bins] = fits ? total : -1;
  }
  __syncthreads();
  DIAG_PP(n, 3);
  if (s_cur[0] < 0) return;   // overflow: the rasteriser scans
  int4 *lst = sl.list + (size_t)n * sl.cap;
  // list entries carry the vertex indices: the rasteriser's setup saves a dependent load
#pragma unroll
  for (int k = 0; k < MVR_PB_KEEP; ++k) {
    const unsigned e = keep_e[k] & 0xffffu;
    if (e == MVR_BIN_NONE) continue;
    const int4 ent = make_int4((int)threadIdx.x + k * MVR_PB_BLOCK, keep_i[k][0], keep_i[k][1], keep_i[k][2]);
    for (int g = e & 0xff; g <= (int)(e >> 8); ++g) lst[atomicAdd(&s_cur[2 * g + ((keep_e[k] >> 16) ? 0 : 1)], 1)] = ent;
  }
  for (int f = (int)threadIdx.x + MVR_PB_KEEP * MVR_PB_BLOCK; f < nf; f += MVR_PB_BLOCK) {
    // faces beyond the register-resident ones (meshes with more than 8192 faces): the bin range is re-read (this
    // thread's own store), only the near flag is re-derived — not the whole cull / range arithmetic
    const unsigned e = bins[(size_t)n * stride + f];
    if (e == MVR_BIN_NONE) continue;
    const int32_t *__restrict__ fc = m.faces + 3 * (size_t)(fo + f);
    const int i0 = fc[0], i1 = fc[1], i2 = fc[2];
    const bool near_face = fminf(fminf(pl[3 * i0 + 2], pl[3 * i1 + 2]), pl[3 * i2 + 2]) < Tv[2];
    const int4 ent = make_int4(f, i0, i1, i2);
    for (int g = e & 0xff; g <= (int)(e >> 8); ++g) lst[atomicAdd(&s_cur[2 * g + (near_face ? 0 : 1)], 1)] = ent;
  }
  __syncthreads();
  DIAG_PP(n, 4);
}

__global__ void __launch_bounds__(MVR_PB_BLOCK) k_project_bins_view(MeshPtrs m, int V, int H, int W, int G, int stride,
                                                                    int cull_backfaces, const float *__restrict__ R,
                                                                    const float *__restrict__ T, float s,
                                                                    float4 *__restrict__ proj_all, unsigned short *__restrict__ bins) {
  extern __shared__ __align__(16) unsigned char smem[];
  project_bins_view_body(m, blockIdx.x, smem, V, H, W, G, stride, cull_backfaces, R, T, s, proj_all, bins);
}

// Pre-pass in ONE launch: workgroups [0, B) build the per-mesh vertex attributes (position + normal), workgroups
// [B, B + N) project one view each and bin its faces.  The two halves are independent, so they simply run
// concurrently on different CUs (no second stream, no event round trips between queues).
__global__ void __launch_bounds__(1024) k_mesh_prepass(MeshPtrs m, int chunks, int V, int H, int W, int G, int stride,
                                                      int cull_backfaces, const float *__restrict__ R,
                                                      const float *__restrict__ T, float s, float4 *__restrict__ proj_all,
                                                      unsigned short *__restrict__ bins, float4 *__restrict__ vattr, StripLists sl,
                                                      ViewConsts vc, LookAtFuse la) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int nb_attr = m.B * chunks;
#ifdef MVR_ABL_PREPASS_NO_NORMALS
  if ((int)blockIdx.x < nb_attr) return;
#endif
#ifdef MVR_ABL_PREPASS_NO_VIEWS
  if ((int)blockIdx.x >= nb_attr) return;
#endif
  if ((int)blockIdx.x < nb_attr) vattr_mesh_lds_body(m, blockIdx.x / chunks, blockIdx.x % chunks, smem, vattr, nullptr);
  else project_bins_view_body(m, (int)blockIdx.x - nb_attr, smem, V, H, W, G, stride, cull_backfaces, R, T, s, proj_all, bins, sl, vc, la);
}

struct RasterArgs {
  MeshPtrs m;
  StripLists sl;               // per-(view, strip) face lists of the pre-pass, or off == nullptr
  const float *vconst;         // per-view shading constants of the pre-pass (8 floats), or nullptr
  const unsigned short *bins;
  int bin_rows, bin_stride;
  const float4 *proj;
  const float4 *vattr;
  const float *R, *T, *light;
  float *images;
  int32_t *p2f;
  int V, N, H, W, TH, n_strips, views_per_xcd;
  int n_light, flags;
  float s;
  float ambient, diffuse, specular, shininess;
  float obj[3], bg[3];
};

__device__ __forceinline__ void decode_block(int bid, int n_strips, int views_per_xcd, int *n, int *strip) {
  // XCD-aware mapping: workgroups bid and bid+8 share an XCD (round-robin dispatch); give each
  // XCD a contiguous range of views so a mesh's faces / projected vertices stay in ONE L2.
  // Within an XCD the order is heaviest-first: the centre strips of all its views, then outwards (objects are
  // centred, so the outermost strips are the cheap / empty ones) — the tail of the launch is then filled with
  // short workgroups instead of waiting for a late 100k-cycle one.
  int xcd = bid & 7, j = bid >> 3;
  const int rank = j / views_per_xcd, mid = n_strips >> 1;
  *n = xcd * views_per_xcd + (j - rank * views_per_xcd);
  *strip = (rank & 1) ? mid - ((rank + 1) >> 1) : mid + (rank >> 1);
}

// The point kernels are store-bound with near-uniform workgroups: plain view-major order (consecutive workgroups of
// an XCD write consecutive strips of one image) measured 9 % faster than the heaviest-first order above.
__device__ __forceinline__ void decode_block_view_major(int bid, int n_strips, int views_per_xcd, int *n, int *strip) {
  int xcd = bid & 7, j = bid >> 3;
  *n = xcd * views_per_xcd + j / n_strips;
  *strip = j % n_strips;
}

// ---- shading of a resolved pixel (device fast path) -------------------------------------------------
// Coverage and depth order are already decided (bit-exactly) when this runs; the colour itself only has to
// meet the 1e-4 pixel tolerance, so it uses the hardware reciprocal / reciprocal-square-root (1 ulp) and lets
// the compiler contract multiply-adds.  Same formulas as mvr::phong_pixel (the exact restatement used by the
// backward kernel and the host tests); observed difference ~1e-6.
struct ShadeConst {
  f3 lh;            // normalised light direction (per view)
  float Rm[9];      // camera rotation (view -> world for directions: d_world = d_view R^T)
  float inv_s;      // 1 / projection scale
  f3 cc;            // camera centre (only for perspective_correct = false)
  f3 objc;
  float ambient, diffuse, specular, shininess;
  const float4 *vat;
  const float *vcol;
};

__device__ __forceinline__ f3 shade_fast(const f3 &P0, const f3 &P1, const f3 &P2, float px, float py, bool persp,
                                         const f3 &N0, const f3 &N1, const f3 &N2,
                                         const f3 &c0, const f3 &c1, const f3 &c2, const ShadeConst &sh, int i0, int i1, int i2) {
#pragma clang fp contract(fast)
  const float e0 = (px - P1.x) * (P2.y - P1.y) - (py - P1.y) * (P2.x - P1.x);
  const float e1 = (px - P2.x) * (P0.y - P2.y) - (py - P2.y) * (P0.x - P2.x);
  const float e2 = (px - P0.x) * (P1.y - P0.y) - (py - P0.y) * (P1.x - P0.x);
  const float area = ((P2.x - P0.x) * (P1.y - P0.y) - (P2.y - P0.y) * (P1.x - P0.x)) + mvr::kEps;
  const float ra = __builtin_amdgcn_rcpf(area);
  float w0 = e0 * ra, w1 = e1 * ra, w2 = e2 * ra;
  if (persp) {
    const float t0 = w0 * P1.z * P2.z, t1 = P0.z * w1 * P2.z, t2 = P0.z * P1.z * w2;
    const float rd = __builtin_amdgcn_rcpf(fmaxf(t0 + t1 + t2, mvr::kEps));
    w0 = t0 * rd; w1 = t1 * rd; w2 = t2 * rd;
  }
  // The interpolated world position p = sum w_i X_i only feeds the view direction C - p.  With perspective-correct
  // barycentrics p is exactly where the pixel ray meets the triangle, so in VIEW space (camera at the origin) it is
  // (px z / s, py z / s, z) with z = sum w_i z_i, and (C - p) = -(x_v, y_v, z_v) R^T: no vertex-position gathers.
  // With perspective_correct = false PyTorch3D's p is the screen-space interpolation (not on the ray): gather X.
  f3 view;
  if (persp) {
    const float zv = w0 * P0.z + w1 * P1.z + w2 * P2.z;
    const float xv = px * zv * sh.inv_s, yv = py * zv * sh.inv_s;
    view = {-(xv * sh.Rm[0] + yv * sh.Rm[1] + zv * sh.Rm[2]), -(xv * sh.Rm[3] + yv * sh.Rm[4] + zv * sh.Rm[5]),
            -(xv * sh.Rm[6] + yv * sh.Rm[7] + zv * sh.Rm[8])};
  } else {
    const f3 X0 = ld3(sh.vat + 2 * i0), X1 = ld3(sh.vat + 2 * i1), X2 = ld3(sh.vat + 2 * i2);
    view = {sh.cc.x - (w0 * X0.x + w1 * X1.x + w2 * X2.x), sh.cc.y - (w0 * X0.y + w1 * X1.y + w2 * X2.y),
            sh.cc.z - (w0 * X0.z + w1 * X1.z + w2 * X2.z)};
  }
  const f3 nn = {w0 * N0.x + w1 * N1.x + w2 * N2.x, w0 * N0.y + w1 * N1.y + w2 * N2.y, w0 * N0.z + w1 * N1.z + w2 * N2.z};
  f3 tex;
  if (sh.vcol) {
    tex = {w0 * c0.x + w1 * c1.x + w2 * c2.x, w0 * c0.y + w1 * c1.y + w2 * c2.y, w0 * c0.z + w1 * c1.z + w2 * c2.z};
  } else {   // one colour for the whole object: sum_i w_i c = (w0 + w1 + w2) c
    const float ws = w0 + w1 + w2;
    tex = {ws * sh.objc.x, ws * sh.objc.y, ws * sh.objc.z};
  }
  const float eps2 = mvr::kNormalizeEps * mvr::kNormalizeEps;
  const float rn = __builtin_amdgcn_rsqf(fmaxf(nn.x * nn.x + nn.y * nn.y + nn.z * nn.z, eps2));  // v / max(|v|, eps)
  const f3 nh = {nn.x * rn, nn.y * rn, nn.z * rn};
  const float cosang = nh.x * sh.lh.x + nh.y * sh.lh.y + nh.z * sh.lh.z;
  const float rv = __builtin_amdgcn_rsqf(fmaxf(view.x * view.x + view.y * view.y + view.z * view.z, eps2));
  const float two_c = 2.0f * cosang;
  const f3 refl = {two_c * nh.x - sh.lh.x, two_c * nh.y - sh.lh.y, two_c * nh.z - sh.lh.z};
  const float dvr = (view.x * refl.x + view.y * refl.y + view.z * refl.z) * rv;
  const float alpha = cosang > 0.0f ? fmaxf(dvr, 0.0f) : 0.0f;
  const float spec = sh.specular * mvr::pow_shininess(alpha, sh.shininess);
  const float sc = sh.ambient + sh.diffuse * fmaxf(cosang, 0.0f);
  return {sc * tex.x + spec, sc * tex.y + spec, sc * tex.z + spec};
}

// Exact-conservative candidate pixel range of a face inside the strip; returns the number of candidates
// (0: the face cannot touch a pixel centre of this strip).
__device__ __forceinline__ int face_range(const mvr::FaceSetup<float> &fs, int W, int H, int jy_lo, int jy_hi,
                                          int *x0, int *x1, int *y0, int *y1) {
  if (!fs.drawable) return 0;
  ndc_range(fs.xmin, fs.xmax, W, x0, x1);
  ndc_range(fs.ymin, fs.ymax, H, y0, y1);
  *y0 = max(*y0, jy_lo); *y1 = min(*y1, jy_hi);
  if (!(*x0 <= *x1 && *y0 <= *y1)) return 0;   // also rejects NaN bounds (ranges come out inverted)
  return (*x1 - *x0 + 1) * (*y1 - *y0 + 1);
}

__device__ __forceinline__ mvr::FaceSetup<float> load_face(const float4 *__restrict__ proj, const int32_t *__restrict__ faces,
                                                          int f, bool cull) {
  int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
  return mvr::face_setup<float>(ld3(proj + i0), ld3(proj + i1), ld3(proj + i2), cull);
}

// The same for a face that is already known to be drawable (the pre-pass only bins / lists faces that pass the culls of
// mvr::face_setup): bounding box, barycentric denominator and its reciprocal — identical values, no cull arithmetic.
__device__ __forceinline__ mvr::FaceSetup<float> load_face_drawable(const float4 *__restrict__ proj, int i0, int i1, int i2) {
  mvr::FaceSetup<float> fs;
  fs.v0 = ld3(proj + i0); fs.v1 = ld3(proj + i1); fs.v2 = ld3(proj + i2);
  fs.xmin = fminf(fminf(fs.v0.x, fs.v1.x), fs.v2.x); fs.xmax = fmaxf(fmaxf(fs.v0.x, fs.v1.x), fs.v2.x);
  fs.ymin = fminf(fminf(fs.v0.y, fs.v1.y), fs.v2.y); fs.ymax = fmaxf(fmaxf(fs.v0.y, fs.v1.y), fs.v2.y);
  fs.area = mvr::edge_fn(fs.v2.x, fs.v2.y, fs.v0.x, fs.v0.y, fs.v1.x, fs.v1.y) + mvr::kEps;
  fs.inv_area = 1.0f / fs.area;
  fs.drawable = true;
  return fs;
}

// LDS record of one face of the current batch: everything the per-pixel test needs.
struct FaceCoef { float4 a, b, c; float2 d, e, g; };  // a = v0.xyz v1.x | b = v1.yz v2.xy | c = v2.z |area| 1/|area| face(bits)
                                                      // d = A0 A1 | e = A2 B0 | g = B1 B2: edge coefficients times sign(area)

__device__ __forceinline__ void unpack_face(const FaceCoef &q, mvr::FaceSetup<float> &fs, int &f) {
  fs.v0 = {q.a.x, q.a.y, q.a.z}; fs.v1 = {q.a.w, q.b.x, q.b.y}; fs.v2 = {q.b.z, q.b.w, q.c.x};
  fs.area = q.c.y; fs.inv_area = q.c.z; f = __float_as_int(q.c.w);
  fs.xmin = mvr::tmin(mvr::tmin(fs.v0.x, fs.v1.x), fs.v2.x); fs.xmax = mvr::tmax(mvr::tmax(fs.v0.x, fs.v1.x), fs.v2.x);
  fs.ymin = mvr::tmin(mvr::tmin(fs.v0.y, fs.v1.y), fs.v2.y); fs.ymax = mvr::tmax(mvr::tmax(fs.v0.y, fs.v1.y), fs.v2.y);
  fs.drawable = true;
}


// ---- packed (2 x f32) arithmetic for a PAIR of horizontally adjacent pixels --------------------------------------
// gfx950 issues v_pk_{add,mul,fma}_f32 at the rate of the scalar forms, and the rasteriser is bound by VALU issue
// (SQ_ACTIVE_INST_VALU = 97 % of the kernel's cycles), so evaluating the exact test of two pixels with packed
// instructions halves its cost.  Each lane-wise operation is the same IEEE operation as the scalar code's.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// mvr::div_by on two numerators (shared divisor b, y = RN(1 / b)) and on two (numerator, divisor) pairs
__device__ __forceinline__ v2f div_by2(v2f a, float b, float y) { const v2f q = a * y; const v2f r = fma2((v2f)(-b), q, a); return fma2(r, (v2f)(y), q); }
__device__ __forceinline__ v2f div_by2v(v2f a, v2f b, v2f y) { const v2f q = a * y; const v2f r = fma2(-b, q, a); return fma2(r, y, q); }
// RN(1 / d) for normal d > 0 whose reciprocal is normal: hardware reciprocal (1 ulp) + two Newton steps with exact
// residuals.  Equality with the IEEE division 1.0f / d over every mantissa and the exponents the rasteriser can
// produce is checked exhaustively on the GPU by mvr_selftest_rcp (tests/test_gpu_parity.py).
__device__ __forceinline__ v2f rcp_rn2(v2f d) {
  v2f y = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  v2f e = fma2(-d, y, (v2f)(1.0f));
  y = fma2(y, e, y);
  e = fma2(-d, y, (v2f)(1.0f));
  return fma2(y, e, y);
}

__global__ void k_selftest_rcp(unsigned lo_bits, unsigned hi_bits, unsigned *__restrict__ n_bad, unsigned *__restrict__ first_bad) {
  const unsigned long long i0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  for (unsigned long long b = (unsigned long long)lo_bits + 2 * i0; b <= hi_bits; b += 2 * stride) {
    const unsigned b1 = (b + 1 <= hi_bits) ? (unsigned)(b + 1) : (unsigned)b;
    const v2f d = {__uint_as_float((unsigned)b), __uint_as_float(b1)};
    const v2f y = rcp_rn2(d);
    const float r0 = 1.0f / d.x, r1 = 1.0f / d.y;
    if (__float_as_uint(y.x) != __float_as_uint(r0)) { if (atomicAdd(n_bad, 1u) == 0) *first_bad = (unsigned)b; }
    if (__float_as_uint(y.y) != __float_as_uint(r1)) { if (atomicAdd(n_bad, 1u) == 0) *first_bad = b1; }
  }
}


// floor(x / d) for 0 <= x < 2^22 and d >= 1, given rd ~ 1 / d (relative error <= 2^-22): the float product is within
// one of the quotient, one multiply-subtract settles it (a 32-bit integer division costs ~25 VALU instructions and
// the rasteriser is bound by VALU issue).
__device__ __forceinline__ int div_small(int x, int d, float rd) {
  int q = (int)((float)x * rd);
  const int r = x - q * d;
  q += (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}

// One workgroup per (view, strip of TH image rows).
//   scan     : 2-byte row-bin ranges of the view's faces (k_face_bins) -> LDS list of faces touching the strip
//   setup    : one lane per listed face: gather its 3 projected vertices, exact candidate range, coefficient
//              record -> LDS; workgroup prefix sum of the candidate counts
//   raster   : the candidate (face, pixel) pairs of the batch are split EVENLY over the 256 lanes (each lane
//              takes a contiguous run of the concatenated per-face pixel lists, located by a binary search of
//              the prefix sums) -> no lane waits for another lane's big triangle; any face size is handled by
//              the same loop.  Depth resolve: 64-bit LDS atomicMin on (float_bits(z) << 32 | face).
//   shade    : every pixel of the strip: winner -> barycentrics -> Phong + hard blend -> planar store.
template <bool PERSP>
__global__ void __launch_bounds__(MVR_RB, MVR_RB_WAVES_PER_SIMD) k_mesh_raster_shade(RasterArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int W = a.W, H = a.H, TH = a.TH;
  unsigned long long *keys = (unsigned long long *)smem;                       // TH * W
  FaceCoef *coef = (FaceCoef *)(keys + (size_t)TH * W);                        // MVR_RBATCH
  int2 *rng = (int2 *)(coef + MVR_RBATCH);                                      // MVR_RBATCH: x0|x1<<16 , y0|y1<<16
  int *cum = (int *)(rng + MVR_RBATCH);                                         // MVR_RBATCH + 1 exclusive prefix sums
  int *list = cum + (MVR_RBATCH + 1);                                           // MVR_LIST_CAP face indices
  int *cnt = list + MVR_LIST_CAP;                                              // [0] list entries, [1..] wave totals (32 ints)
  float *px_tab = (float *)(cnt + 32);                                         // W + 1 pixel-centre x coordinates
  float *py_tab = px_tab + (W + 1);                                            // TH pixel-centre y coordinates (NDC row jy_lo + i)
  unsigned *tz = (unsigned *)(py_tab + TH);                                    // depth bound per tile of 2^MVR_TILE_SHIFT columns (hierarchical z)
  __shared__ unsigned short start[MVR_RB];                                  // first record of every lane's run

  int n, strip;
  decode_block(blockIdx.x, a.n_strips, a.views_per_xcd, &n, &strip);
  if (n >= a.N) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = n / a.V, v = n - b * a.V;
  const int vo = a.m.vert_off[b], nv = a.m.vert_off[b + 1] - vo;
  const int fo = a.m.face_off[b], nf = a.m.face_off[b + 1] - fo;
  const float4 *__restrict__ proj = a.proj + (size_t)a.V * vo + (size_t)v * nv;
  const int32_t *__restrict__ faces = a.m.faces + 3 * (size_t)fo;
  const unsigned short *__restrict__ bins = a.bins + (size_t)n * a.bin_stride;
  const int r0 = strip * TH;
  const int rows = min(TH, H - r0);
  const int jy_hi = H - 1 - r0, jy_lo = jy_hi - (rows - 1);  // NDC row indices covered by the strip
  const unsigned g_lo = (unsigned)(r0 / a.bin_rows), g_hi = (unsigned)((r0 + rows - 1) / a.bin_rows);
  constexpr bool persp = PERSP;   // compile-time: the two variants share nothing in the hot loop
  const bool cull = (a.flags & MVR_CULL_BACKFACES) != 0;
  // pixel centre of NDC index i: -1 + (2 i + 1)/S, the division done as div_by with RN(1/S): bit-identical
  const float fW = (float)W, fH = (float)H, invW = 1.0f / fW, invH = 1.0f / fH;

  // face list of this strip from the pre-pass (else: scan the view's row-bin ranges below)
  const int4 *gl = nullptr;
  int gl_n = 0, gl_near = 0;
  if (a.sl.off) {
    const int *off = a.sl.off + (size_t)n * (a.sl.n_bins + 1);
    if (off[a.sl.n_bins] >= 0) {
      const int o0 = off[strip];
      gl = a.sl.list + (size_t)n * a.sl.cap + o0;
      gl_n = (strip + 1 < a.sl.n_bins ? off[strip + 1] : off[a.sl.n_bins]) - o0;
      gl_near = a.sl.front[(size_t)n * a.sl.n_bins + strip];
      if (gl_n <= MVR_RBATCH) gl_near = gl_n;   // one batch holds the whole strip: no second setup / tile pass
    }
  }
  const bool use_list = gl != nullptr;
  if (use_list && gl_n == 0) {   // nothing can touch this strip: background only
    const size_t HW0 = (size_t)H * W;
    float *__restrict__ img0 = a.images + (size_t)n * 3 * HW0 + (size_t)r0 * W;
    int32_t *__restrict__ p2f0 = a.p2f ? a.p2f + (size_t)n * HW0 + (size_t)r0 * W : nullptr;
    if ((W & 3) == 0) {   // 16-byte stores (rows start 16-byte aligned when W is a multiple of 4)
      const float4 b0 = make_float4(a.bg[0], a.bg[0], a.bg[0], a.bg[0]), b1 = make_float4(a.bg[1], a.bg[1], a.bg[1], a.bg[1]),
                   b2 = make_float4(a.bg[2], a.bg[2], a.bg[2], a.bg[2]);
      for (int g = tid; g < (rows * W) >> 2; g += MVR_RB) {
        *reinterpret_cast<float4 *>(img0 + 4 * g) = b0;
        *reinterpret_cast<float4 *>(img0 + HW0 + 4 * g) = b1;
        *reinterpret_cast<float4 *>(img0 + 2 * HW0 + 4 * g) = b2;
        if (p2f0) *reinterpret_cast<int4 *>(p2f0 + 4 * g) = make_int4(-1, -1, -1, -1);
      }
      return;
    }
    for (int pix = tid; pix < rows * W; pix += MVR_RB) {
      img0[pix] = a.bg[0]; img0[HW0 + pix] = a.bg[1]; img0[2 * HW0 + pix] = a.bg[2];
      if (p2f0) p2f0[pix] = -1;
    }
    return;
  }
  // (face, indices) entry of this lane's face in the NEXT batch, requested one batch ahead (list mode): the setup then
  // has one dependent gather (the projected vertices) instead of three
  int4 ent_pf = make_int4(0, 0, 0, 0);
  if (use_list) {
    const int first = gl_near > 0 ? 0 : gl_near, first_hi = gl_near > 0 ? gl_near : gl_n;
    if (first + tid < first_hi) ent_pf = gl[first + tid];
  }
  for (int i = tid; i <= W; i += MVR_RB) px_tab[i] = -1.0f + mvr::div_by(2.0f * (float)i + 1.0f, fW, invW);
  if (tid < rows) py_tab[tid] = -1.0f + mvr::div_by(2.0f * (float)(jy_lo + tid) + 1.0f, fH, invH);
  if (tid == 0) cnt[0] = 0;
  if (use_list) for (int i = tid; i < rows * W; i += MVR_RB) keys[i] = MVR_KEY_EMPTY;
  __syncthreads();

  DIAG_STAMP(0);
#ifdef MVR_DIAG
  unsigned long long t_scan = 0, t_small = 0, t_large = 0, n_small_tot = 0, n_large_tot = 0;
  unsigned long long tc0 = clock64();
#endif
  const int faces_per_pass = MVR_RB * 8 * MVR_V3_ROUNDS;
  for (int pbase = 0; pbase < nf; pbase += faces_per_pass) {
    int total_hits = gl_n, my_pos = 0;
    unsigned mask = 0;   // bit (8 r + k): face pbase + (r * BLOCK + tid) * 8 + k touches this strip
    if (!use_list) {
    // ---- scan: the row-bin vectors of up to 8192 faces are requested up front (4 x 16 B per lane); hit masks
    //      stay in registers; one wave prefix sum + ONE LDS atomic per wave orders the hits ---------------------
    uint4 q[MVR_V3_ROUNDS];
#pragma unroll
    for (int r = 0; r < MVR_V3_ROUNDS; ++r) {
      const int f0 = pbase + (r * MVR_RB + tid) * 8;
      q[r] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      if (f0 < nf) q[r] = *reinterpret_cast<const uint4 *>(bins + f0);
    }
    if (pbase == 0) for (int i = tid; i < rows * W; i += MVR_RB) keys[i] = MVR_KEY_EMPTY;   // under the loads' latency
#pragma unroll
    for (int r = 0; r < MVR_V3_ROUNDS; ++r) {
      const int f0 = pbase + (r * MVR_RB + tid) * 8;
      const unsigned w[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const unsigned e = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
        const bool h = (e != MVR_BIN_NONE) && (e & 0xffu) <= g_hi && (e >> 8) >= g_lo && (f0 + k) < nf;
        mask |= (h ? 1u : 0u) << (8 * r + k);
      }
    }
    const int my_hits = __popc(mask);
    int hincl = my_hits;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(hincl, off);
      if (lane >= off) hincl += t;
    }
    int wave_base = 0;
    if (lane == 63 && hincl) wave_base = atomicAdd(&cnt[0], hincl);
    wave_base = __shfl(wave_base, 63);
    my_pos = wave_base + hincl - my_hits;   // rank of this lane's first hit in the pass
    __syncthreads();
    total_hits = cnt[0];
    }   // !use_list
    // List mode draws the strip's faces in two segments: the near faces, then the far ones — and before the far ones it
    // reduces the depth keys to one bound per 8-column tile, so that a far face whose nearest vertex lies behind
    // everything already drawn over its bounding box is dropped in the setup (hierarchical z).  Sound for any order:
    // such a face would lose the min at every one of its pixels.
    for (int seg = 0; seg < (use_list ? 2 : 1); ++seg) {
    const int seg_lo = (use_list && seg == 1) ? gl_near : 0;
    const int seg_hi = use_list ? (seg == 0 ? gl_near : gl_n) : total_hits;
    if (use_list && seg_hi <= seg_lo) continue;
    const bool hiz = use_list && seg == 1 && gl_near > 0 && seg_hi > seg_lo;
    if (hiz) {
      const int n_tiles = (W + (1 << MVR_TILE_SHIFT) - 1) >> MVR_TILE_SHIFT;
      for (int i = tid; i < n_tiles; i += MVR_RB) tz[i] = 0u;
      __syncthreads();
      int col = tid % W;
      const int dcol = MVR_RB % W;
      for (int pix = tid; pix < rows * W; pix += MVR_RB) {
        atomicMax(&tz[col >> MVR_TILE_SHIFT], (unsigned)(keys[pix] >> 32));   // empty pixel: 0xFFFFFFFF, nothing is behind it
        col += dcol; if (col >= W) col -= W;
      }
      __syncthreads();
    }
    for (int win = seg_lo; win < seg_hi; win += MVR_LIST_CAP) {
    if (!use_list) {  // materialise hits [win, win + CAP) of the pass in the LDS list
      int pos = my_pos - win;
      unsigned m = mask;
      while (m) {
        const int bit = __ffs(m) - 1;
        m &= m - 1;
        if (pos >= 0 && pos < MVR_LIST_CAP) list[pos] = pbase + ((bit >> 3) * MVR_RB + tid) * 8 + (bit & 7);
        ++pos;
      }
      __syncthreads();
    }
    const int *lst = list;
    const int n_list = min(MVR_LIST_CAP, seg_hi - win);
#ifdef MVR_DIAG
    unsigned long long tc1 = clock64();
    t_scan += tc1 - tc0; n_small_tot += n_list;
#endif
    for (int b0 = 0; b0 < n_list; b0 += MVR_RBATCH) {
      const int nb = min(MVR_RBATCH, n_list - b0);
#ifdef MVR_DIAG
      unsigned long long ts0 = clock64();
#endif
      // ---- setup: one lane per face of the batch.  Faces without a candidate pixel in this strip are dropped: the
      //      records are written at the face's RANK among the non-empty ones, so "next face" in the loop is ++e ----
      int my_cnt = 0;
      FaceCoef q;
      int2 rr = make_int2(0, 0);
      if (tid < nb) {
        int4 ent = ent_pf;
        if (!use_list) { const int ff = lst[b0 + tid]; ent = make_int4(ff, faces[3 * ff], faces[3 * ff + 1], faces[3 * ff + 2]); }
        const int f = ent.x;
        const mvr::FaceSetup<float> fs = load_face_drawable(proj, ent.y, ent.z, ent.w);
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        if (face_range(fs, W, H, jy_lo, jy_hi, &x0, &x1, &y0, &y1) > 0) {
          // trim the conservative range to EXACTLY the pixel centres inside the face's bounding box (the rule of the
          // reference's per-pixel bbox test, evaluated on the same float values), so the loop needs no bbox test
          while (x0 <= x1 && px_tab[x0] < fs.xmin) ++x0;
          while (x1 >= x0 && px_tab[x1] > fs.xmax) --x1;
          while (y0 <= y1 && py_tab[y0 - jy_lo] < fs.ymin) ++y0;
          while (y1 >= y0 && py_tab[y1 - jy_lo] > fs.ymax) --y1;
          if (x0 <= x1 && y0 <= y1)
            my_cnt = ((x1 - x0 + 2) >> 1) * (y1 - y0 + 1);   // work items = PAIRS of horizontally adjacent candidate pixels
          if (hiz && my_cnt > 0) {
            // interpolated depth >= min vertex depth up to rounding (the barycentrics are positive and sum to 1 within
            // a few ulp; without perspective correction the sum is 1 - kEps/area, hence the area guard there)
            const float zmin = fminf(fminf(fs.v0.z, fs.v1.z), fs.v2.z);
            const int t0 = (W - 1 - x1) >> MVR_TILE_SHIFT, t1 = (W - 1 - x0) >> MVR_TILE_SHIFT;
            if (t1 - t0 <= MVR_TILE_SPAN && zmin > 1e-3f && (PERSP || fabsf(fs.area) > 1e-3f)) {
              unsigned zm = tz[t0];
              for (int t = t0 + 1; t <= t1; ++t) zm = max(zm, tz[t]);
              if (zmin * (1.0f - 1e-5f) > __uint_as_float(zm)) my_cnt = 0;   // 0xFFFFFFFF (an empty pixel) is a NaN: never true
            }
          }
        }
        q.a = make_float4(fs.v0.x, fs.v0.y, fs.v0.z, fs.v1.x);
        q.b = make_float4(fs.v1.y, fs.v1.z, fs.v2.x, fs.v2.y);
        // For a back face (area < 0) the edge coefficients, the area and its reciprocal are all negated: every edge
        // value and every quotient e_i / area is then bit-identical up to that common sign (IEEE negation is exact
        // and commutes with rounding), so the barycentrics come out the same and "inside" is "all three > 0".
        const float sg = fs.area > 0.0f ? 1.0f : -1.0f;
        q.c = make_float4(fs.v2.z, sg * fs.area, sg * fs.inv_area, __int_as_float(f));
        q.d = make_float2(sg * (fs.v2.y - fs.v1.y), sg * (fs.v0.y - fs.v2.y));
        q.e = make_float2(sg * (fs.v1.y - fs.v0.y), sg * (fs.v2.x - fs.v1.x));
        q.g = make_float2(sg * (fs.v0.x - fs.v2.x), sg * (fs.v1.x - fs.v0.x));
        rr = make_int2(x0 | (x1 << 16), y0 | (y1 << 16));
      }
      const bool has = my_cnt > 0;
      const unsigned long long hb = __ballot(has);
      // workgroup exclusive prefix sums of the candidate counts and of the non-empty flags
      int incl = my_cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      if (lane == 63) { cnt[1 + wave] = incl; cnt[1 + MVR_RW + wave] = __popcll(hb); }
      __syncthreads();
      int wave_off = 0, total = 0, rec_off = 0;
#pragma unroll
      for (int w = 0; w < MVR_RW; ++w) {
        const int t = cnt[1 + w], r = cnt[1 + MVR_RW + w];
        if (w < wave) { wave_off += t; rec_off += r; }
        total += t;
      }
#ifdef MVR_DIAG
      unsigned long long ts1 = clock64();
      t_small += ts1 - ts0; n_large_tot += total;
#endif
      // ---- raster: balanced split of the batch's `total` (face, pixel) candidates over the lanes ---------------
      const int quota = (total + MVR_RB - 1) / MVR_RB;
      const float rquota = 1.0f / (float)(quota > 0 ? quota : 1);
      if (has) {
        const int c = rec_off + __popcll(hb & ((1ull << lane) - 1ull));
        const int c0 = wave_off + incl - my_cnt;
        coef[c] = q; rng[c] = rr; cum[c] = c0;
        // the face owns the lanes whose run starts inside [c0, c0 + my_cnt): it writes its record index into start[]
        // for them (no dependent LDS chain, unlike a binary search per lane)
        int l0, l1;
        if (total < (1 << 22)) { l0 = div_small(c0 + quota - 1, quota, rquota); l1 = div_small(c0 + my_cnt - 1, quota, rquota); }
        else { l0 = (c0 + quota - 1) / quota; l1 = (c0 + my_cnt - 1) / quota; }
        if (l1 > MVR_RB - 1) l1 = MVR_RB - 1;
        for (int l = l0; l <= l1; ++l) start[l] = (unsigned short)c;
      }
      __syncthreads();
      if (use_list) {   // request the next batch's entries now; they land while this batch is rasterised
        int nxt = win + MVR_LIST_CAP, nxt_hi = seg_hi;
        if (nxt >= seg_hi) { if (seg == 0) { nxt = gl_near; nxt_hi = gl_n; } else { nxt = nxt_hi = 0; } }
        if (nxt + tid < nxt_hi) ent_pf = gl[nxt + tid];
      }
      const int p0 = tid * quota;
#ifdef MVR_ABL_NO_LOOP
      int rem = 0;
#else
      int rem = min(quota, total - p0);
#endif
      if (rem > 0) {
        int e = start[tid];   // the record whose candidates contain item p0
        // per-face state.  Edge i of the exact test (mvr::face_test / edge_fn, same operations in the same order):
        //   e0 = (px - X1) * (Y2 - Y1) - (py - Y1) * (X2 - X1),  e1 = (px - X2) * (Y0 - Y2) - (py - Y2) * (X0 - X2),
        //   e2 = (px - X0) * (Y1 - Y0) - (py - Y0) * (X1 - X0); the second products only change with the row.
        // The coefficients A_i = +-(Y_j - Y_k), B_i = +-(X_j - X_k), |area| and its reciprocal come signed from the setup.
        float X0, Y0, Z0, X1, Y1, Z1, X2, Y2, Z2, sarea, sinv, A0, A1, A2, B0, B1, B2;
        int f, x0, x1, ix, jy, left;
        float R0, R1, R2;
        unsigned long long *kp;
#define MVR_LOAD_FACE()                                                                                      \
        {                                                                                                    \
          const FaceCoef &qq = coef[e];                                                                      \
          const float4 qa = qq.a, qb = qq.b, qc = qq.c;                                                      \
          const float2 qd = qq.d, qe = qq.e, qg = qq.g;                                                      \
          X0 = qa.x; Y0 = qa.y; Z0 = qa.z; X1 = qa.w; Y1 = qb.x; Z1 = qb.y; X2 = qb.z; Y2 = qb.w;            \
          Z2 = qc.x; sarea = qc.y; sinv = qc.z; f = __float_as_int(qc.w);                                    \
          A0 = qd.x; A1 = qd.y; A2 = qe.x; B0 = qe.y; B1 = qg.x; B2 = qg.y;                                  \
        }
#define MVR_SET_ROW()                                                                                        \
        {                                                                                                    \
          const float py = py_tab[jy - jy_lo];                                                               \
          R0 = (py - Y1) * B0; R1 = (py - Y2) * B1; R2 = (py - Y0) * B2;                                     \
        }
        MVR_LOAD_FACE();
        {
          const int2 r = rng[e];
          x0 = r.x & 0xffff; x1 = r.x >> 16;
          const int y0 = r.y & 0xffff, y1 = r.y >> 16;
          const int npx = (x1 - x0 + 2) >> 1;
          const int o = p0 - cum[e];
          const int dy = div_small(o, npx, __builtin_amdgcn_rcpf((float)npx));   // o < 2^18: at most 8 rows of 2^14 pairs
          jy = y0 + dy; ix = x0 + 2 * (o - dy * npx);
          left = npx * (y1 - y0 + 1) - o;
        }
        MVR_SET_ROW();
        kp = keys + (__mul24(jy_hi - jy, W) + (W - 1 - ix));
        v2f px = {px_tab[ix], px_tab[ix + 1]};   // requested one iteration ahead: its LDS latency hides behind the pair math
        while (true) {
          // two horizontally adjacent pixels (ix, ix + 1) per iteration, evaluated with packed f32 instructions;
          // straight-line: the barycentrics of both are always computed, the final "> 0" tests decide (the sign
          // pre-filter of mvr::face_test only skips work, it rejects nothing the final tests would accept)
          const v2f t0 = (px - X1) * A0 - R0, t1 = (px - X2) * A1 - R1, t2 = (px - X0) * A2 - R2;
          v2f a0 = div_by2(t0, sarea, sinv), a1 = div_by2(t1, sarea, sinv), a2 = div_by2(t2, sarea, sinv);
          if (persp) {
            const v2f u0 = (a0 * Z1) * Z2, u1 = (Z0 * a1) * Z2, u2 = (Z0 * Z1) * a2;
            v2f den = (u0 + u1) + u2;
            den.x = mvr::tmax(den.x, mvr::kEps); den.y = mvr::tmax(den.y, mvr::kEps);
            const v2f yd = rcp_rn2(den);
            a0 = div_by2v(u0, den, yd); a1 = div_by2v(u1, den, yd); a2 = div_by2v(u2, den, yd);
          }
          const v2f pz = (a0 * Z0 + a1 * Z1) + a2 * Z2;
          if ((a0.x > 0.0f) & (a1.x > 0.0f) & (a2.x > 0.0f) & !(pz.x < 0.0f))
            atomicMin(kp, ((unsigned long long)__float_as_uint(pz.x + 0.0f) << 32) | (unsigned)f);
          if ((ix < x1) & (a0.y > 0.0f) & (a1.y > 0.0f) & (a2.y > 0.0f) & !(pz.y < 0.0f))
            atomicMin(kp - 1, ((unsigned long long)__float_as_uint(pz.y + 0.0f) << 32) | (unsigned)f);
          if (--rem == 0) break;
          if (--left == 0) {
            ++e;
            MVR_LOAD_FACE();
            const int2 r = rng[e];
            x0 = r.x & 0xffff; x1 = r.x >> 16;
            const int y0 = r.y & 0xffff, y1 = r.y >> 16;
            left = ((x1 - x0 + 2) >> 1) * (y1 - y0 + 1);
            ix = x0; jy = y0;
            MVR_SET_ROW();
            kp = keys + (__mul24(jy_hi - jy, W) + (W - 1 - ix));
          } else if (ix + 2 > x1) {
            kp += (ix - x0) - W;   // next NDC row = previous image row, back to column x0
            ix = x0; ++jy;
            MVR_SET_ROW();
          } else {
            ix += 2;
            kp -= 2;
          }
          px = (v2f){px_tab[ix], px_tab[ix + 1]};
        }
#undef MVR_LOAD_FACE
#undef MVR_SET_ROW
      }
      __syncthreads();   // coef / rng / cum are rewritten by the next batch; keys are complete for this one
#ifdef MVR_DIAG
      t_large += clock64() - ts1;
#endif
    }
#ifdef MVR_DIAG
    tc0 = clock64();
#endif
    }   // window
    }   // segment
    if (use_list) break;
    if (pbase + faces_per_pass < nf) {
      if (tid == 0) cnt[0] = 0;
      __syncthreads();
    }
  }
  DIAG_STAMP(1);
#ifdef MVR_DIAG
  if (threadIdx.x == 0 && g_diag) {
#ifndef MVR_DIAG_SHADE
    g_diag[(size_t)blockIdx.x * 8 + 3] = t_scan; g_diag[(size_t)blockIdx.x * 8 + 4] = t_small;
#endif
    g_diag[(size_t)blockIdx.x * 8 + 5] = t_large; g_diag[(size_t)blockIdx.x * 8 + 6] = n_small_tot;
    g_diag[(size_t)blockIdx.x * 8 + 7] = n_large_tot;
  }
#endif

  // ---- resolve: background pixels are stored at once; covered pixels are COMPACTED into an LDS list so
  //      the shading arithmetic below runs with every lane busy (coverage is ~40 % of the frame) ----------
  const size_t HW = (size_t)H * W;
  float *__restrict__ img = a.images + (size_t)n * 3 * HW + (size_t)r0 * W;
  int32_t *__restrict__ p2f = a.p2f ? a.p2f + (size_t)n * HW + (size_t)r0 * W : nullptr;
  ShadeConst sh;
  {
    float Rm[9], Tv[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) Rm[k] = a.R[9 * (size_t)n + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) Tv[k] = a.T[3 * (size_t)n + k];
#pragma unroll
    for (int k = 0; k < 9; ++k) sh.Rm[k] = Rm[k];
    if (a.vconst) {
      const float *c = a.vconst + 8 * (size_t)n;
      sh.lh = {c[0], c[1], c[2]}; sh.cc = {c[3], c[4], c[5]}; sh.inv_s = c[6];
    } else {
      sh.inv_s = 1.0f / a.s;
      sh.cc = mvr::camera_center<float>(Rm, Tv);
      const float *lp = a.light + 3 * (size_t)(a.n_light == 1 ? 0 : n);
      sh.lh = mvr::normalize<float>({lp[0], lp[1], lp[2]}, mvr::kNormalizeEps);
    }
    sh.objc = {a.obj[0], a.obj[1], a.obj[2]};
    sh.ambient = a.ambient; sh.diffuse = a.diffuse; sh.specular = a.specular; sh.shininess = a.shininess;
    sh.vat = a.vattr + 2 * (size_t)vo;
    sh.vcol = a.m.vert_colors ? a.m.vert_colors + 3 * (size_t)vo : nullptr;
  }

  int *plist = (int *)coef;                       // the batch records are dead now: reuse their LDS
  const int plist_cap = (int)((sizeof(FaceCoef) + sizeof(int2)) * MVR_RBATCH / sizeof(int));
  const int n_pix = rows * W;
  if (tid == 0) cnt[0] = 0;
  __syncthreads();
  const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];
  for (int pbase = 0; pbase < n_pix; pbase += plist_cap) {
    const int pend = min(n_pix, pbase + plist_cap);
    // groups of 8 x 256 pixels: the 8 key reads are issued together and ONE LDS atomic per wave reserves the list slots
    // of the whole group (one per 256 pixels used to serialise 7 dependent atomic round trips per strip)
    for (int g0 = pbase; g0 < pend; g0 += 8 * MVR_RB) {
      unsigned long long kk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int pix = g0 + k * MVR_RB + tid;
        kk[k] = pix < pend ? keys[pix] : 0ull;   // 0: "not covered and nothing to store" marker for out-of-range lanes
      }
      int cntk[8], total = 0;
      unsigned covbits = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int pix = g0 + k * MVR_RB + tid;
        const bool in = pix < pend;
        const bool covered = in && kk[k] != MVR_KEY_EMPTY;
        if (in && !covered) {
          img[pix] = bg0; img[HW + pix] = bg1; img[2 * HW + pix] = bg2;
          if (p2f) p2f[pix] = -1;
        }
        const unsigned long long mc = __ballot(covered);
        cntk[k] = __popcll(mc);
        total += cntk[k];
        // rank of this lane among the covered lanes of round k, kept in 6 bits per round
        covbits |= covered ? (1u << k) : 0u;
        kk[k] = (unsigned long long)__popcll(mc & ((1ull << lane) - 1ull));
      }
      int bs = 0;
      if (lane == 0 && total) bs = atomicAdd(&cnt[0], total);
      bs = __shfl(bs, 0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (covbits & (1u << k)) plist[bs + (int)kk[k]] = g0 + k * MVR_RB + tid;
        bs += cntk[k];
      }
    }
    __syncthreads();
#ifdef MVR_DIAG_SHADE
    if (threadIdx.x == 0 && g_diag) g_diag[(size_t)blockIdx.x * 8 + 3] = clock64() - g_diag[(size_t)blockIdx.x * 8 + 1];
    unsigned long long tsh0 = clock64();
#endif
    const int n_cov = cnt[0];
    // ---- shade the covered pixels (dense).  Software-pipelined: the index chain of the NEXT pixel
    //      (LDS list -> LDS key -> face's vertex indices from L2) is issued before the current pixel's
    //      9 attribute gathers are consumed, so the two dependent latencies overlap. ------------------------
    int e = tid;
    int pix = 0, face = 0, i0 = 0, i1 = 0, i2 = 0;
    if (e < n_cov) {
      pix = plist[e];
      face = (int)(unsigned)(keys[pix] & 0xffffffffull);
      i0 = faces[3 * face]; i1 = faces[3 * face + 1]; i2 = faces[3 * face + 2];
    }
    while (e < n_cov) {
      const f3 P0 = ld3(proj + i0), P1 = ld3(proj + i1), P2 = ld3(proj + i2);
      const f3 N0 = ld3(sh.vat + 2 * i0 + 1), N1 = ld3(sh.vat + 2 * i1 + 1), N2 = ld3(sh.vat + 2 * i2 + 1);
      f3 c0 = sh.objc, c1 = sh.objc, c2 = sh.objc;
      if (sh.vcol) {
        c0 = {sh.vcol[3 * i0], sh.vcol[3 * i0 + 1], sh.vcol[3 * i0 + 2]};
        c1 = {sh.vcol[3 * i1], sh.vcol[3 * i1 + 1], sh.vcol[3 * i1 + 2]};
        c2 = {sh.vcol[3 * i2], sh.vcol[3 * i2 + 1], sh.vcol[3 * i2 + 2]};
      }
      const int cur_pix = pix, cur_face = face, ci0 = i0, ci1 = i1, ci2 = i2;
      e += MVR_RB;
      if (e < n_cov) {  // prefetch the next pixel's indices
        pix = plist[e];
        face = (int)(unsigned)(keys[pix] & 0xffffffffull);
        i0 = faces[3 * face]; i1 = faces[3 * face + 1]; i2 = faces[3 * face + 2];
      }
      const int row = cur_pix / W, col = cur_pix - row * W;
      const float px = -1.0f + mvr::div_by(2.0f * (float)(W - 1 - col) + 1.0f, fW, invW);
      const float py = -1.0f + mvr::div_by(2.0f * (float)(jy_hi - row) + 1.0f, fH, invH);
#ifdef MVR_ABL_NO_SHADE
      const f3 rgb = {P0.x + N0.x + c0.x + px, P1.x + N1.y + c1.x + py, P2.x + N2.z + c2.x};
#else
      const f3 rgb = shade_fast(P0, P1, P2, px, py, persp, N0, N1, N2, c0, c1, c2, sh, ci0, ci1, ci2);
#endif
      img[cur_pix] = rgb.x; img[HW + cur_pix] = rgb.y; img[2 * HW + cur_pix] = rgb.z;
      if (p2f) p2f[cur_pix] = cur_face;
    }
#ifdef MVR_DIAG_SHADE
    __syncthreads();
    if (threadIdx.x == 0 && g_diag) g_diag[(size_t)blockIdx.x * 8 + 4] = clock64() - tsh0;
#endif
    if (pend < n_pix) {
      __syncthreads();
      if (tid == 0) cnt[0] = 0;
      __syncthreads();
    }
  }
  DIAG_STAMP(2);
}

// ======================================================================================
// mesh backward: one workgroup per (view, strip); 18 partial sums per workgroup
// ======================================================================================

struct MeshBwdArgs {
  MeshPtrs m;
  const float4 *vattr;
  const float *R, *T, *light;
  const int32_t *p2f;
  const float *gimg;
  float *partials;  // (N, n_strips, 18)
  int V, N, H, W, TH, n_strips, views_per_xcd;
  int n_light, flags;
  float s;
  float ambient, diffuse, specular, shininess;
  float obj[3];
};

#ifndef MVR_BWD_STRIP_MULT
#define MVR_BWD_STRIP_MULT 4
#endif
#define MVR_BWD_SCAN_GROUPS (64 / MVR_WAVES)   // (group, wave) counts one wave can scan at a time
#ifndef MVR_BWD_WAVES_PER_SIMD
#define MVR_BWD_WAVES_PER_SIMD 3   // 168 VGPRs without spills; 4 (128) spills 42 registers and doubles the time
#endif
__global__ void __launch_bounds__(MVR_BLOCK, MVR_BWD_WAVES_PER_SIMD) k_mesh_backward(MeshBwdArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  int *plist = (int *)smem;                               // rows * W covered-pixel indices
  __shared__ float s_part[MVR_WAVES][MVR_NACC];
  __shared__ int s_goff[MVR_BWD_SCAN_GROUPS * MVR_WAVES + 1];
  int n, strip;
  decode_block(blockIdx.x, a.n_strips, a.views_per_xcd, &n, &strip);
  if (n >= a.N) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.W, H = a.H;
  const int b = n / a.V;
  const int vo = a.m.vert_off[b];
  const int fo = a.m.face_off[b];
  const int32_t *__restrict__ faces = a.m.faces + 3 * (size_t)fo;
  const float4 *__restrict__ vat = a.vattr + 2 * (size_t)vo;
  const float *__restrict__ vcol = a.m.vert_colors ? a.m.vert_colors + 3 * (size_t)vo : nullptr;
  const int r0 = strip * a.TH;
  const int rows = min(a.TH, H - r0);
  const int n_pix = rows * W;
  const size_t HW = (size_t)H * W;
  const int32_t *__restrict__ p2f = a.p2f + (size_t)n * HW + (size_t)r0 * W;
  const float *__restrict__ g = a.gimg + (size_t)n * 3 * HW + (size_t)r0 * W;

  // ---- compact the covered pixels of the strip (coverage ~40 %): the gradient arithmetic below then runs dense.
  // Groups of 8 x 256 pixels (the 8 pix_to_face loads of a group are in flight together).  The list order is FIXED
  // (group, wave, round, lane) so that the float sums below are bit-reproducible from run to run: a first sweep
  // counts the covered pixels per (group, wave), one wave turns the counts into offsets, a second sweep (its loads
  // hit the cache) writes the list.  Chunks of MVR_BWD_SCAN_GROUPS groups bound the offset table.
  const int n_groups = (n_pix + 8 * MVR_BLOCK - 1) / (8 * MVR_BLOCK);
  int chunk_base = 0;
  for (int c0 = 0; c0 < n_groups; c0 += MVR_BWD_SCAN_GROUPS) {
    const int c1 = min(n_groups, c0 + MVR_BWD_SCAN_GROUPS);
    for (int gi = c0; gi < c1; ++gi) {
      const int g0 = gi * 8 * MVR_BLOCK;
      int total = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int pix = g0 + k * MVR_BLOCK + tid;
        total += __popcll(__ballot(pix < n_pix && p2f[pix] >= 0));
      }
      if (lane == 0) s_goff[(gi - c0) * MVR_WAVES + wave] = total;
    }
    __syncthreads();
    if (wave == 0) {   // exclusive prefix over the (group, wave) counts of the chunk; the last slot receives the chunk total
      const int n_ent = (c1 - c0) * MVR_WAVES;
      const int v = lane < n_ent ? s_goff[lane] : 0;
      int incl = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
      }
      if (lane < n_ent) s_goff[lane] = incl - v;
      if (lane == 63) s_goff[MVR_BWD_SCAN_GROUPS * MVR_WAVES] = incl;
    }
    __syncthreads();
    for (int gi = c0; gi < c1; ++gi) {
      const int g0 = gi * 8 * MVR_BLOCK;
      int bs = chunk_base + s_goff[(gi - c0) * MVR_WAVES + wave];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int pix = g0 + k * MVR_BLOCK + tid;
        const bool covered = pix < n_pix && p2f[pix] >= 0;
        const unsigned long long mc = __ballot(covered);
        if (covered) plist[bs + __popcll(mc & ((1ull << lane) - 1ull))] = pix;
        bs += __popcll(mc);
      }
    }
    chunk_base += s_goff[MVR_BWD_SCAN_GROUPS * MVR_WAVES];
    __syncthreads();
  }
  const int n_cov = chunk_base;

  mvr::PixelGradIn<float> cam;
#pragma unroll
  for (int k = 0; k < 9; ++k) cam.R[k] = a.R[9 * (size_t)n + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) cam.Tr[k] = a.T[3 * (size_t)n + k];
  cam.s = a.s;
  const float *lp = a.light + 3 * (size_t)(a.n_light == 1 ? 0 : n);
  {   // the light direction is per view: normalise it once here; its normalisation backward runs once in the finalize
    float idl; bool cl;
    cam.l = mvr::normalize_rs<float>({lp[0], lp[1], lp[2]}, mvr::kNormalizeEps, &idl, &cl);
    cam.l_is_unit = true;
  }
  cam.cc = mvr::camera_center<float>(cam.R, cam.Tr);
  cam.ph = {a.ambient, a.diffuse, a.specular, a.shininess};
  cam.persp = (a.flags & MVR_PERSPECTIVE_CORRECT) != 0;
  const f3 objc = {a.obj[0], a.obj[1], a.obj[2]};
  const float fW = (float)W, fH = (float)H, invW = 1.0f / fW, invH = 1.0f / fH;

  float acc[MVR_NACC];
#pragma unroll
  for (int k = 0; k < MVR_NACC; ++k) acc[k] = 0.f;
  // software-pipelined like the forward shade loop: next pixel's index chain overlaps this pixel's gathers
  int e = tid, pix = 0, i0 = 0, i1 = 0, i2 = 0;
  if (e < n_cov) {
    pix = plist[e];
    const int face = p2f[pix];
    i0 = faces[3 * face]; i1 = faces[3 * face + 1]; i2 = faces[3 * face + 2];
  }
  while (e < n_cov) {
    const f3 X0 = ld3(vat + 2 * i0), X1 = ld3(vat + 2 * i1), X2 = ld3(vat + 2 * i2);
    const f3 N0 = ld3(vat + 2 * i0 + 1), N1 = ld3(vat + 2 * i1 + 1), N2 = ld3(vat + 2 * i2 + 1);
    f3 c0 = objc, c1 = objc, c2 = objc;
    if (vcol) {
      c0 = {vcol[3 * i0], vcol[3 * i0 + 1], vcol[3 * i0 + 2]};
      c1 = {vcol[3 * i1], vcol[3 * i1 + 1], vcol[3 * i1 + 2]};
      c2 = {vcol[3 * i2], vcol[3 * i2 + 1], vcol[3 * i2 + 2]};
    }
    const int cur = pix;
    const f3 G = {g[cur], g[HW + cur], g[2 * HW + cur]};
    e += MVR_BLOCK;
    if (e < n_cov) {
      pix = plist[e];
      const int face = p2f[pix];
      i0 = faces[3 * face]; i1 = faces[3 * face + 1]; i2 = faces[3 * face + 2];
    }
    const int row = cur / W, col = cur - row * W;
    const float px = -1.0f + mvr::div_by(2.0f * (float)(W - 1 - col) + 1.0f, fW, invW);
    const float py = -1.0f + mvr::div_by(2.0f * (float)(H - 1 - (r0 + row)) + 1.0f, fH, invH);
    mvr::phong_pixel_bwd<float>(cam, px, py, X0, X1, X2, N0, N1, N2, c0, c1, c2, G, acc);
  }
  // wave reduce (64 lanes), then 4 waves through LDS in a fixed order: with the fixed list order above the
  // partial sums — and therefore the gradients — are bit-reproducible
#pragma unroll
  for (int k = 0; k < MVR_NACC; ++k) {
    float x = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    if (lane == 0) s_part[wave][k] = x;
  }
  __syncthreads();
  if (tid < MVR_NACC) {
    float x = 0.f;
#pragma unroll
    for (int w = 0; w < MVR_WAVES; ++w) x += s_part[w][tid];
    a.partials[((size_t)n * a.n_strips + strip) * MVR_NACC + tid] = x;
  }
}

// one 64-lane wave per view: lanes stride over the strips' partials (18 sums), shuffle-reduce, then fold the
// camera-centre gradient into (gR, gT).
// acc[15..17] holds the summed gradient w.r.t. the view's UNIT light vector: apply the normalisation backward once
__device__ __forceinline__ void light_unit_to_raw_grad(const float *__restrict__ light, int n_light, int n, float *g) {
  const float *lp = light + 3 * (size_t)(n_light == 1 ? 0 : n);
  float idl; bool cl;
  const f3 lh = mvr::normalize_rs<float>({lp[0], lp[1], lp[2]}, mvr::kNormalizeEps, &idl, &cl);
  const f3 gl = mvr::normalize_rs_bwd<float>(lh, idl, cl, {g[0], g[1], g[2]});
  g[0] = gl.x; g[1] = gl.y; g[2] = gl.z;
}

__global__ void __launch_bounds__(MVR_BLOCK) k_mesh_backward_finalize(const float *__restrict__ partials, int N, int n_strips,
                                                                      const float *__restrict__ R, const float *__restrict__ T,
                                                                      float *__restrict__ gR, float *__restrict__ gT,
                                                                      float *__restrict__ gL, const float *__restrict__ light,
                                                                      int n_light) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (MVR_BLOCK / 64) + (threadIdx.x >> 6);
  if (n >= N) return;
  float acc[MVR_NACC];
#pragma unroll
  for (int k = 0; k < MVR_NACC; ++k) acc[k] = 0.f;
  for (int s = lane; s < n_strips; s += 64) {
    const float *p = partials + ((size_t)n * n_strips + s) * MVR_NACC;
#pragma unroll
    for (int k = 0; k < MVR_NACC; ++k) acc[k] += p[k];
  }
#pragma unroll
  for (int k = 0; k < MVR_NACC; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
  }
  if (lane == 0) {
    float Rm[9], Tv[3];
    for (int k = 0; k < 9; ++k) Rm[k] = R[9 * (size_t)n + k];
    for (int k = 0; k < 3; ++k) Tv[k] = T[3 * (size_t)n + k];
    mvr::fold_camera_center<float>(Rm, Tv, acc + 12, acc, acc + 9);
    for (int k = 0; k < 9; ++k) gR[9 * (size_t)n + k] = acc[k];
    for (int k = 0; k < 3; ++k) gT[3 * (size_t)n + k] = acc[9 + k];
    if (gL) {
      light_unit_to_raw_grad(light, n_light, n, acc + 15);
      for (int k = 0; k < 3; ++k) gL[3 * (size_t)n + k] = acc[15 + k];
    }
  }
}

// The tail of a training step in ONE launch: sum the strips' partials of a view (as k_mesh_backward_finalize), then the
// look-at backward at the angles the forward used and, for the relative light, the camera-position backward at the
// requested angles (both in f64, rounded once, as k_look_at_bwd / k_camera_position_bwd), added together.
__global__ void __launch_bounds__(MVR_BLOCK) k_mesh_views_backward_finalize(const float *__restrict__ partials, int N, int n_strips,
                                                                            const float *__restrict__ R, const float *__restrict__ T,
                                                                            const float *__restrict__ azim, const float *__restrict__ elev,
                                                                            const float *__restrict__ dist,
                                                                            const float *__restrict__ azim_used,
                                                                            const float *__restrict__ elev_used, int relative_light,
                                                                            float *__restrict__ g_azim, float *__restrict__ g_elev,
                                                                            float *__restrict__ g_dist, const float *__restrict__ light,
                                                                            int n_light) {
  // TWO waves per view: both sum the view's partials; the even wave then runs the look-at chain and the odd wave the
  // camera-position chain of the relative light — two long f64 dependency chains side by side (in one wave the two
  // branches would run one after the other).
  __shared__ float s_light[MVR_BLOCK / 128][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = wave >> 1, role = wave & 1;
  const int n = blockIdx.x * (MVR_BLOCK / 128) + slot;
  const bool live = n < N;
  float acc[MVR_NACC];
#pragma unroll
  for (int k = 0; k < MVR_NACC; ++k) acc[k] = 0.f;
  if (live) {
    for (int s = lane; s < n_strips; s += 64) {
      const float *p = partials + ((size_t)n * n_strips + s) * MVR_NACC;
#pragma unroll
      for (int k = 0; k < MVR_NACC; ++k) acc[k] += p[k];
    }
  }
#pragma unroll
  for (int k = 0; k < MVR_NACC; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off);
  }
  float fa = 0.f, fe = 0.f, fd = 0.f;
  if (live && lane == 0) {
    if (role == 0) {
      float Rm[9], Tv[3];
      for (int k = 0; k < 9; ++k) Rm[k] = R[9 * (size_t)n + k];
      for (int k = 0; k < 3; ++k) Tv[k] = T[3 * (size_t)n + k];
      mvr::fold_camera_center<float>(Rm, Tv, acc + 12, acc, acc + 9);
      double gr[9], gt[3], ga, ge, gd;
      for (int k = 0; k < 9; ++k) gr[k] = (double)acc[k];
      for (int k = 0; k < 3; ++k) gt[k] = (double)acc[9 + k];
      mvr::look_at_bwd<double>((double)azim_used[n], (double)elev_used[n], (double)dist[n], gr, gt, &ga, &ge, &gd);
      fa = (float)ga; fe = (float)ge; fd = (float)gd;
    } else if (relative_light) {
      light_unit_to_raw_grad(light, n_light, n, acc + 15);
      const V3<double> gl = {(double)acc[15], (double)acc[16], (double)acc[17]};
      double la, le, ld;
      mvr::camera_position_bwd<double>((double)azim[n], (double)elev[n], (double)dist[n], gl, &la, &le, &ld);
      s_light[slot][0] = (float)la; s_light[slot][1] = (float)le; s_light[slot][2] = (float)ld;
    }
  }
  __syncthreads();
  if (live && lane == 0 && role == 0) {   // two float gradients added, like autograd's accumulation
    g_azim[n] = relative_light ? fa + s_light[slot][0] : fa;
    g_elev[n] = relative_light ? fe + s_light[slot][1] : fe;
    g_dist[n] = relative_light ? fd + s_light[slot][2] : fd;
  }
}

// ======================================================================================
// point clouds: one workgroup per (view, strip)
// ======================================================================================

struct PointsArgs {
  const float *points, *colors, *R, *T, *dist;
  const float4 *pv;            // (N, pstride) view-space points (k_points_project)
  const unsigned short *pbin;  // (N, pstride) row-bin range of each point's disc, 0xFFFF = cannot hit
  float *images;
  int32_t *p2p;
  int B, P, V, N, H, W, TH, n_strips, views_per_xcd, pstride, bin_rows;
  float radius, r2, r2w;
  float obj[3], bg[3];
};

__device__ __forceinline__ f3 point_view(const float *Rm, const float *Tv, float sc, const float *p) {
  f3 xs = {p[0] * sc, p[1] * sc, p[2] * sc};  // Pointclouds.scale_(1/dist), models/renderer.py:142-143
  return mvr::world_to_view<float>(Rm, Tv, xs);
}

// grid (ceil(P / BLOCK), N): every point is transformed ONCE per view (the rasteriser's strips used to redo it
// 28 times) and gets the conservative range of row-bins its radius-r disc can reach.
__global__ void __launch_bounds__(MVR_BLOCK) k_points_project(PointsArgs a, float4 *__restrict__ pv, unsigned short *__restrict__ pbin) {
  const int n = blockIdx.y;
  const int b = n / a.V;
  float Rm[9], Tv[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rm[k] = a.R[9 * (size_t)n + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) Tv[k] = a.T[3 * (size_t)n + k];
  const float sc = 1.0f / a.dist[n];
  const float rG = __builtin_amdgcn_rcpf((float)a.bin_rows);
  // 4 points per lane (the view's camera is loaded once per 1024 points); rows < 2^15, so the row-bin quotient is a
  // float product settled by one correction instead of a 32-bit integer division
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = (blockIdx.x * 4 + k) * MVR_BLOCK + threadIdx.x;
    if (p >= a.P) break;
    const f3 q = point_view(Rm, Tv, sc, a.points + 3 * ((size_t)b * a.P + p));
    unsigned short out = MVR_BIN_NONE;
    if (q.z >= 0.0f && q.x - a.radius <= 1.0f && q.x + a.radius >= -1.0f && q.y - a.radius <= 1.0f && q.y + a.radius >= -1.0f) {
      int y0, y1;
      ndc_range(q.y - a.radius, q.y + a.radius, a.H, &y0, &y1);
      if (y0 <= y1) {
        const int r0 = a.H - 1 - y1, r1 = a.H - 1 - y0;
        int g0 = (int)((float)r0 * rG), g1 = (int)((float)r1 * rG);
        g0 += (r0 - g0 * a.bin_rows >= a.bin_rows ? 1 : 0) - (r0 - g0 * a.bin_rows < 0 ? 1 : 0);
        g1 += (r1 - g1 * a.bin_rows >= a.bin_rows ? 1 : 0) - (r1 - g1 * a.bin_rows < 0 ? 1 : 0);
        out = (unsigned short)(g0 | (g1 << 8));
      }
    }
    pv[(size_t)n * a.pstride + p] = make_float4(q.x, q.y, q.z, 0.f);
    pbin[(size_t)n * a.pstride + p] = out;
  }
}

// Look-at + projection in one launch (MVR_PB_BLOCK threads per workgroup): the forward of the point path then is two
// launches (this + the rasteriser) instead of three.  A 1024-thread workgroup with the look-at's f64 chains fills a
// CU, so every workgroup takes `vpw` consecutive views such that all workgroups are resident at once (256 CUs).
#define MVR_POINTS_PREPASS_MAX_VPW 8
__global__ void __launch_bounds__(MVR_PB_BLOCK) k_points_prepass(PointsArgs a, LookAtFuse la, int vpw, float4 *__restrict__ pv,
                                                                unsigned short *__restrict__ pbin) {
  __shared__ float s_RT[12 * MVR_POINTS_PREPASS_MAX_VPW];
  __shared__ int s_bad;
  const int n0 = blockIdx.x * vpw, n_own = min(vpw, a.N - n0);
  look_at_views_block(la, n0, s_RT, &s_bad, n_own);
  const float rG = __builtin_amdgcn_rcpf((float)a.bin_rows);
  for (int k = 0; k < n_own; ++k) {
    const int n = n0 + k, b = n / a.V;
    float Rm[9], Tv[3];
#pragma unroll
    for (int j = 0; j < 9; ++j) Rm[j] = s_RT[12 * k + j];
#pragma unroll
    for (int j = 0; j < 3; ++j) Tv[j] = s_RT[12 * k + 9 + j];
    const float sc = 1.0f / a.dist[n];
    for (int p = threadIdx.x; p < a.P; p += MVR_PB_BLOCK) {   // same arithmetic as k_points_project
      const f3 q = point_view(Rm, Tv, sc, a.points + 3 * ((size_t)b * a.P + p));
      unsigned short out = MVR_BIN_NONE;
      if (q.z >= 0.0f && q.x - a.radius <= 1.0f && q.x + a.radius >= -1.0f && q.y - a.radius <= 1.0f && q.y + a.radius >= -1.0f) {
        int y0, y1;
        ndc_range(q.y - a.radius, q.y + a.radius, a.H, &y0, &y1);
        if (y0 <= y1) {
          const int r0 = a.H - 1 - y1, r1 = a.H - 1 - y0;
          int g0 = (int)((float)r0 * rG), g1 = (int)((float)r1 * rG);
          g0 += (r0 - g0 * a.bin_rows >= a.bin_rows ? 1 : 0) - (r0 - g0 * a.bin_rows < 0 ? 1 : 0);
          g1 += (r1 - g1 * a.bin_rows >= a.bin_rows ? 1 : 0) - (r1 - g1 * a.bin_rows < 0 ? 1 : 0);
          out = (unsigned short)(g0 | (g1 << 8));
        }
      }
      pv[(size_t)n * a.pstride + p] = make_float4(q.x, q.y, q.z, 0.f);
      pbin[(size_t)n * a.pstride + p] = out;
    }
  }
}

__global__ void __launch_bounds__(MVR_BLOCK) k_points_raster_shade(PointsArgs a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int W = a.W, H = a.H, TH = a.TH;
  unsigned long long *keys = (unsigned long long *)smem;
  float *px_tab = (float *)(keys + (size_t)TH * W);
  float *py_tab = px_tab + W;
  int n, strip;
  decode_block_view_major(blockIdx.x, a.n_strips, a.views_per_xcd, &n, &strip);
  if (n >= a.N) return;
  const int tid = threadIdx.x;
  const int b = n / a.V;
  const int r0 = strip * TH;
  const int rows = min(TH, H - r0);
  const int jy_hi = H - 1 - r0, jy_lo = jy_hi - (rows - 1);
  const unsigned g_lo = (unsigned)(r0 / a.bin_rows), g_hi = (unsigned)((r0 + rows - 1) / a.bin_rows);
  for (int i = tid; i < rows * W; i += MVR_BLOCK) keys[i] = MVR_KEY_EMPTY;
  for (int i = tid; i < W; i += MVR_BLOCK) px_tab[i] = mvr::pix_to_ndc<float>(i, W);
  for (int i = tid; i < rows; i += MVR_BLOCK) py_tab[i] = mvr::pix_to_ndc<float>(jy_lo + i, H);
  __syncthreads();
  const float4 *__restrict__ pv = a.pv + (size_t)n * a.pstride;
  const unsigned short *__restrict__ pbin = a.pbin + (size_t)n * a.pstride;
#ifdef MVR_ABL_PTS_NO_SCAN
  const int P_eff = 0;
#else
  const int P_eff = a.P;
#endif
  // scan: 4 points (one 8-byte vector of row-bin ranges) per lane per chunk of 1024.  Only ~7 % of a cloud reaches a
  // strip, so the hits are first COMPACTED into an LDS list and then splatted with every lane busy (testing them in
  // place made every wave run the splat code for every point slot: 40 M VALU instructions, 94 % VALU-bound).
  int *hlist = (int *)(py_tab + TH);                  // shared with the resolve's group list below
  int *hcnt = hlist + ((size_t)TH * (W >> 2) > MVR_PTS_LIST_INTS ? (size_t)TH * (W >> 2) : MVR_PTS_LIST_INTS);
  for (int c0 = 0; c0 < P_eff; c0 += MVR_BLOCK * 8) {   // 2048 points per chunk: both 8-byte vectors are requested together
    if (tid == 0) *hcnt = 0;
    __syncthreads();
    uint2 qb[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int p0 = c0 + (v * MVR_BLOCK + tid) * 4;
      qb[v] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
      if (p0 < P_eff) qb[v] = *reinterpret_cast<const uint2 *>(pbin + p0);
    }
    int hp[8], nh = 0;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      const int p0 = c0 + (v * MVR_BLOCK + tid) * 4;
      const unsigned w[2] = {qb[v].x, qb[v].y};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned e = (w[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
        const bool h = !(e == MVR_BIN_NONE || (e & 0xffu) > g_hi || (e >> 8) < g_lo || p0 + k >= a.P);
        hp[nh] = p0 + k;      // branch-free append
        nh += h ? 1 : 0;
      }
    }
    if (nh) {
      const int base = atomicAdd(hcnt, nh);
#pragma unroll
      for (int k = 0; k < 8; ++k) if (k < nh) hlist[base + k] = hp[k];
    }
    __syncthreads();
    const int n_hit = *hcnt;
    for (int e = tid; e < n_hit; e += MVR_BLOCK) {
      const int p = hlist[e];
      const float4 q = pv[p];
      int x0, x1, y0, y1;
      ndc_range(q.x - a.radius, q.x + a.radius, W, &x0, &x1);
      ndc_range(q.y - a.radius, q.y + a.radius, H, &y0, &y1);
      y0 = max(y0, jy_lo); y1 = min(y1, jy_hi);
      const unsigned long long key = ((unsigned long long)__float_as_uint(q.z + 0.0f) << 32) | (unsigned)p;
      for (int jy = y0; jy <= y1; ++jy) {
        const float dy = q.y - py_tab[jy - jy_lo];
        unsigned long long *krow = keys + (size_t)(jy_hi - jy) * W + (W - 1);
        for (int ix = x0; ix <= x1; ++ix) {
          const float dx = q.x - px_tab[ix];
          const float dd = dx * dx + dy * dy;
          if (dd < a.r2) atomicMin(krow - ix, key);
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  const size_t HW = (size_t)H * W;
  float *__restrict__ img = a.images + (size_t)n * 3 * HW + (size_t)r0 * W;
  int32_t *__restrict__ p2p = a.p2p ? a.p2p + (size_t)n * HW + (size_t)r0 * W : nullptr;
  const float *__restrict__ cols = a.colors ? a.colors + 3 * (size_t)b * a.P : nullptr;
  // colour of one covered pixel: w = 1 - d2 / r2, out = (f w) / max(w, 1e-4) (NormWeightedCompositor, K = 1).  The
  // quotients go through mvr::div_by with the correctly rounded reciprocals of their divisors (== IEEE division,
  // 3 instructions instead of ~11; r2 is uniform, max(w, 1e-4) lies in [2^-14, 1] where rcp_rn2 is verified).
  const float inv_r2w = 1.0f / a.r2w;
  auto shade_point = [&](unsigned long long key, int pix, float &cr, float &cg, float &cb, int &pi) {
    pi = (int)(unsigned)(key & 0xffffffffull);
    const int row = pix / W, col = pix - row * W;
    const float4 q = pv[pi];
    const float dx = q.x - px_tab[W - 1 - col], dy = q.y - py_tab[jy_hi - row - jy_lo];
    const float dd = dx * dx + dy * dy;
    const float w = 1.0f - mvr::div_by(dd, a.r2w, inv_r2w);
    const float cum = w < 1e-4f ? 1e-4f : w;  // NormWeightedCompositor kEpsilon clamp
    const float yc = rcp_rn2((v2f){cum, cum}).x;
    float f0 = a.obj[0], f1 = a.obj[1], f2 = a.obj[2];
    if (cols) { f0 = cols[3 * (size_t)pi]; f1 = cols[3 * (size_t)pi + 1]; f2 = cols[3 * (size_t)pi + 2]; }
    cr = mvr::div_by(f0 * w, cum, yc); cg = mvr::div_by(f1 * w, cum, yc); cb = mvr::div_by(f2 * w, cum, yc);
  };
  if ((W & 3) == 0) {
    // The kernel is bound by its 3 x 4-byte stores per pixel and only ~5 % of the pixels are covered: groups of 4
    // consecutive pixels without a point are written as three 16-byte stores straight away; the few groups with
    // a covered pixel are listed in LDS and finished afterwards (dense), also with 16-byte stores.
    int *glist = hlist;
    int *gcnt = hcnt;
    if (tid == 0) *gcnt = 0;
    __syncthreads();
    const int n_groups = rows * (W >> 2);
    const float4 bg4[3] = {make_float4(a.bg[0], a.bg[0], a.bg[0], a.bg[0]), make_float4(a.bg[1], a.bg[1], a.bg[1], a.bg[1]),
                           make_float4(a.bg[2], a.bg[2], a.bg[2], a.bg[2])};
    for (int g = tid; g < n_groups; g += MVR_BLOCK) {
      const ulonglong2 k01 = *reinterpret_cast<const ulonglong2 *>(keys + 4 * g);
      const ulonglong2 k23 = *reinterpret_cast<const ulonglong2 *>(keys + 4 * g + 2);
      if ((k01.x & k01.y & k23.x & k23.y) == MVR_KEY_EMPTY) {
#ifndef MVR_ABL_PTS_NO_STORE
        *reinterpret_cast<float4 *>(img + 4 * g) = bg4[0];
        *reinterpret_cast<float4 *>(img + HW + 4 * g) = bg4[1];
        *reinterpret_cast<float4 *>(img + 2 * HW + 4 * g) = bg4[2];
        if (p2p) *reinterpret_cast<int4 *>(p2p + 4 * g) = make_int4(-1, -1, -1, -1);
#endif
      } else {
        glist[atomicAdd(gcnt, 1)] = g;
      }
    }
    __syncthreads();
    const int n_slow = *gcnt;
    for (int e = tid; e < n_slow; e += MVR_BLOCK) {
      const int g = glist[e];
      float cr[4], cg[4], cb[4];
      int pi[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned long long key = keys[4 * g + k];
        cr[k] = a.bg[0]; cg[k] = a.bg[1]; cb[k] = a.bg[2]; pi[k] = -1;
        if (key != MVR_KEY_EMPTY) shade_point(key, 4 * g + k, cr[k], cg[k], cb[k], pi[k]);
      }
      *reinterpret_cast<float4 *>(img + 4 * g) = make_float4(cr[0], cr[1], cr[2], cr[3]);
      *reinterpret_cast<float4 *>(img + HW + 4 * g) = make_float4(cg[0], cg[1], cg[2], cg[3]);
      *reinterpret_cast<float4 *>(img + 2 * HW + 4 * g) = make_float4(cb[0], cb[1], cb[2], cb[3]);
      if (p2p) *reinterpret_cast<int4 *>(p2p + 4 * g) = make_int4(pi[0], pi[1], pi[2], pi[3]);
    }
    return;
  }
  for (int pix = tid; pix < rows * W; pix += MVR_BLOCK) {   // image width not a multiple of 4: pixel by pixel
    const unsigned long long key = keys[pix];
    float cr = a.bg[0], cg = a.bg[1], cb = a.bg[2];
    int pi = -1;
    if (key != MVR_KEY_EMPTY) shade_point(key, pix, cr, cg, cb, pi);
    img[pix] = cr; img[HW + pix] = cg; img[2 * HW + pix] = cb;
    if (p2p) p2p[pix] = pi;
  }
}

// points_per_pixel = K > 1 (PointsRasterizationSettings.points_per_pixel, models/renderer.py:129-133; the reference's
// default is 1 and takes k_points_raster_shade above).  The K nearest points of every pixel in ascending (z, index)
// are found by K min-passes over the strip: pass k keeps, per pixel, the smallest key that is >= lo[pix], where lo
// is one past the key chosen by pass k-1 (keys are unique per point, so ties in z are enumerated by index exactly
// as a sort by (z, index) would).  The slots are written to idx_out (N,H,W,K) as they are found, then composed
// NormWeightedCompositor-style in slot order: out_c = sum_k (f_ck w_k) / max(sum_k w_k, 1e-4).
__global__ void __launch_bounds__(MVR_BLOCK) k_points_raster_topk(PointsArgs a, int K, int32_t *__restrict__ idx_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int W = a.W, H = a.H, TH = a.TH;
  unsigned long long *keys = (unsigned long long *)smem;
  unsigned long long *lo = keys + (size_t)TH * W;
  float *px_tab = (float *)(lo + (size_t)TH * W);
  float *py_tab = px_tab + W;
  int n, strip;
  decode_block_view_major(blockIdx.x, a.n_strips, a.views_per_xcd, &n, &strip);
  if (n >= a.N) return;
  const int tid = threadIdx.x;
  const int b = n / a.V;
  const int r0 = strip * TH;
  const int rows = min(TH, H - r0);
  const int jy_hi = H - 1 - r0, jy_lo = jy_hi - (rows - 1);
  const unsigned g_lo = (unsigned)(r0 / a.bin_rows), g_hi = (unsigned)((r0 + rows - 1) / a.bin_rows);
  for (int i = tid; i < rows * W; i += MVR_BLOCK) lo[i] = 0ull;
  for (int i = tid; i < W; i += MVR_BLOCK) px_tab[i] = mvr::pix_to_ndc<float>(i, W);
  for (int i = tid; i < rows; i += MVR_BLOCK) py_tab[i] = mvr::pix_to_ndc<float>(jy_lo + i, H);
  const float4 *__restrict__ pv = a.pv + (size_t)n * a.pstride;
  const unsigned short *__restrict__ pbin = a.pbin + (size_t)n * a.pstride;
  const size_t HW = (size_t)H * W;
  int32_t *slots = idx_out + ((size_t)n * HW + (size_t)r0 * W) * K;
  int k_done = 0;
  for (int k = 0; k < K; ++k) {
    for (int i = tid; i < rows * W; i += MVR_BLOCK) keys[i] = MVR_KEY_EMPTY;
    __syncthreads();
    for (int p0 = tid * 4; p0 < a.P; p0 += MVR_BLOCK * 4) {
      const uint2 qb = *reinterpret_cast<const uint2 *>(pbin + p0);
      const unsigned w[2] = {qb.x, qb.y};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned e = (w[j >> 1] >> ((j & 1) * 16)) & 0xffffu;
        const int p = p0 + j;
        if (e == MVR_BIN_NONE || (e & 0xffu) > g_hi || (e >> 8) < g_lo || p >= a.P) continue;
        const float4 q = pv[p];
        const unsigned long long key = ((unsigned long long)__float_as_uint(q.z + 0.0f) << 32) | (unsigned)p;
        int x0, x1, y0, y1;
        ndc_range(q.x - a.radius, q.x + a.radius, W, &x0, &x1);
        ndc_range(q.y - a.radius, q.y + a.radius, H, &y0, &y1);
        y0 = max(y0, jy_lo); y1 = min(y1, jy_hi);
        for (int jy = y0; jy <= y1; ++jy) {
          const float dy = q.y - py_tab[jy - jy_lo];
          const size_t rbase = (size_t)(jy_hi - jy) * W + (W - 1);
          for (int ix = x0; ix <= x1; ++ix) {
            const float dx = q.x - px_tab[ix];
            const float dd = dx * dx + dy * dy;
            if (dd < a.r2 && key >= lo[rbase - ix]) atomicMin(keys + rbase - ix, key);
          }
        }
      }
    }
    __syncthreads();
    int any = 0;
    for (int pix = tid; pix < rows * W; pix += MVR_BLOCK) {
      const unsigned long long key = keys[pix];
      const bool hit = key != MVR_KEY_EMPTY;
      slots[(size_t)pix * K + k] = hit ? (int)(unsigned)(key & 0xffffffffull) : -1;
      lo[pix] = hit ? key + 1ull : MVR_KEY_EMPTY;
      any |= hit ? 1 : 0;
    }
    k_done = k + 1;
    if (!__syncthreads_or(any)) break;  // no pixel of the strip found a k-th point: the remaining slots are empty
  }
  for (int pix = tid; pix < rows * W; pix += MVR_BLOCK)
    for (int k = k_done; k < K; ++k) slots[(size_t)pix * K + k] = -1;
  float *__restrict__ img = a.images + (size_t)n * 3 * HW + (size_t)r0 * W;
  const float *__restrict__ cols = a.colors ? a.colors + 3 * (size_t)b * a.P : nullptr;
  for (int pix = tid; pix < rows * W; pix += MVR_BLOCK) {
    const int row = pix / W, col = pix - row * W;
    const float px = px_tab[W - 1 - col], py = py_tab[jy_hi - row - jy_lo];
    const int32_t *sl = slots + (size_t)pix * K;   // written by this same thread above
    float cr = a.bg[0], cg = a.bg[1], cb = a.bg[2];
    if (sl[0] >= 0) {
      float cum = 0.f;
      for (int k = 0; k < k_done; ++k) {
        const int pi = sl[k];
        if (pi < 0) break;
        const float4 q = pv[pi];
        const float dx = q.x - px, dy = q.y - py;
        cum += 1.0f - (dx * dx + dy * dy) / a.r2w;
      }
      if (cum < 1e-4f) cum = 1e-4f;  // NormWeightedCompositor kEpsilon clamp
      cr = cg = cb = 0.f;
      for (int k = 0; k < k_done; ++k) {
        const int pi = sl[k];
        if (pi < 0) break;
        const float4 q = pv[pi];
        const float dx = q.x - px, dy = q.y - py;
        const float w = 1.0f - (dx * dx + dy * dy) / a.r2w;
        float f0 = a.obj[0], f1 = a.obj[1], f2 = a.obj[2];
        if (cols) { f0 = cols[3 * (size_t)pi]; f1 = cols[3 * (size_t)pi + 1]; f2 = cols[3 * (size_t)pi + 2]; }
        cr += (f0 * w) / cum; cg += (f1 * w) / cum; cb += (f2 * w) / cum;
      }
    }
    img[pix] = cr; img[HW + pix] = cg; img[2 * HW + pix] = cb;
  }
}

#define MVR_PACC 13  // gR[9], gT[3], g(1/dist)

struct PointsBwdArgs {
  const float *points, *colors, *R, *T, *dist, *gimg;
  const int32_t *p2p;
  float *partials;  // (N, n_strips, 13)
  int B, P, V, N, H, W, TH, n_strips, views_per_xcd, K;
  float r2w;
  float obj[3];
};

__global__ void __launch_bounds__(MVR_BLOCK) k_points_backward(PointsBwdArgs a) {
  __shared__ float s_part[MVR_WAVES][MVR_PACC];
  int n, strip;
  decode_block_view_major(blockIdx.x, a.n_strips, a.views_per_xcd, &n, &strip);
  if (n >= a.N) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = a.W, H = a.H;
  const int b = n / a.V;
  const int r0 = strip * a.TH;
  const int rows = min(a.TH, H - r0);
  const size_t HW = (size_t)H * W;
  float Rm[9], Tv[3];
#pragma unroll
  for (int k = 0; k < 9; ++k) Rm[k] = a.R[9 * (size_t)n + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) Tv[k] = a.T[3 * (size_t)n + k];
  const float sc = 1.0f / a.dist[n];
  const float *pts = a.points + 3 * (size_t)b * a.P;
  const float *cols = a.colors ? a.colors + 3 * (size_t)b * a.P : nullptr;
  const int32_t *p2p = a.p2p + ((size_t)n * HW + (size_t)r0 * W) * a.K;
  const float *g = a.gimg + (size_t)n * 3 * HW + (size_t)r0 * W;
  float acc[MVR_PACC];
#pragma unroll
  for (int k = 0; k < MVR_PACC; ++k) acc[k] = 0.f;
  // PyTorch3D's norm_weighted_sum backward: d out_c / d w_k = (f_ck S - sum_j w_j f_cj) / S^2 with the CLAMPED
  // S = max(sum_j w_j, 1e-4).  For K = 1 away from the clamp this is f w - w f = 0 exactly, so those pixels are skipped.
  const int K = a.K;
  for (int pix = tid; pix < rows * W; pix += MVR_BLOCK) {
    const int32_t *sl = p2p + (size_t)pix * K;
    if (sl[0] < 0) continue;
    int row = pix / W, col = pix - row * W;
    float px = mvr::pix_to_ndc<float>(W - 1 - col, W), py = mvr::pix_to_ndc<float>(H - 1 - (r0 + row), H);
    float S = 0.f, A0 = 0.f, A1 = 0.f, A2 = 0.f;
    for (int k = 0; k < K; ++k) {
      const int pi = sl[k];
      if (pi < 0) break;
      f3 q = point_view(Rm, Tv, sc, pts + 3 * (size_t)pi);
      float dx = q.x - px, dy = q.y - py;
      float w = 1.0f - (dx * dx + dy * dy) / a.r2w;
      float f0 = a.obj[0], f1 = a.obj[1], f2 = a.obj[2];
      if (cols) { f0 = cols[3 * (size_t)pi]; f1 = cols[3 * (size_t)pi + 1]; f2 = cols[3 * (size_t)pi + 2]; }
      S += w; A0 += w * f0; A1 += w * f1; A2 += w * f2;
    }
    if (K == 1 && !(S < 1e-4f)) continue;
    if (S < 1e-4f) S = 1e-4f;
    const float S2 = S * S;
    const float g0 = g[pix], g1 = g[HW + pix], g2 = g[2 * HW + pix];
    for (int k = 0; k < K; ++k) {
      const int pi = sl[k];
      if (pi < 0) break;
      const float *pw = pts + 3 * (size_t)pi;
      f3 q = point_view(Rm, Tv, sc, pw);
      float dx = q.x - px, dy = q.y - py;
      float f0 = a.obj[0], f1 = a.obj[1], f2 = a.obj[2];
      if (cols) { f0 = cols[3 * (size_t)pi]; f1 = cols[3 * (size_t)pi + 1]; f2 = cols[3 * (size_t)pi + 2]; }
      float gw = (((f0 * S - A0) / S2) * g0 + ((f1 * S - A1) / S2) * g1) + ((f2 * S - A2) / S2) * g2;
      float gdd = -gw / a.r2w;
      float gx = 2.0f * dx * gdd, gy = 2.0f * dy * gdd;
      // q = (sc * Xw) R + T
      float xs = pw[0] * sc, ys = pw[1] * sc, zs = pw[2] * sc;
      acc[0] += xs * gx; acc[1] += xs * gy;
      acc[3] += ys * gx; acc[4] += ys * gy;
      acc[6] += zs * gx; acc[7] += zs * gy;
      acc[9] += gx; acc[10] += gy;
      acc[12] += (pw[0] * Rm[0] + pw[1] * Rm[3] + pw[2] * Rm[6]) * gx + (pw[0] * Rm[1] + pw[1] * Rm[4] + pw[2] * Rm[7]) * gy;
    }
  }
#pragma unroll
  for (int k = 0; k < MVR_PACC; ++k) {
    float x = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    if (lane == 0) s_part[wave][k] = x;
  }
  __syncthreads();
  if (tid < MVR_PACC) {
    float x = 0.f;
#pragma unroll
    for (int w = 0; w < MVR_WAVES; ++w) x += s_part[w][tid];
    a.partials[((size_t)n * a.n_strips + strip) * MVR_PACC + tid] = x;
  }
}

__global__ void __launch_bounds__(MVR_BLOCK) k_points_backward_finalize(const float *__restrict__ partials, int N, int n_strips,
                                                                        const float *__restrict__ dist, float *__restrict__ gR,
                                                                        float *__restrict__ gT, float *__restrict__ gD) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float acc[MVR_PACC];
  for (int k = 0; k < MVR_PACC; ++k) acc[k] = 0.f;
  for (int s = 0; s < n_strips; ++s)
    for (int k = 0; k < MVR_PACC; ++k) acc[k] += partials[((size_t)n * n_strips + s) * MVR_PACC + k];
  for (int k = 0; k < 9; ++k) gR[9 * (size_t)n + k] = acc[k];
  for (int k = 0; k < 3; ++k) gT[3 * (size_t)n + k] = acc[9 + k];
  float d = dist[n];
  gD[n] = -acc[12] / (d * d);  // sc = 1/dist
}

// ======================================================================================
// host side: C ABI
// ======================================================================================

static int default_strip_rows(int H, int W) {
  // 8 B of depth key per pixel: <= 16 KiB of keys per workgroup keeps ~5+ workgroups resident per CU
  // (160 KiB LDS) — the rasteriser hides its gather latency with occupancy
  int th = 8;
  while (th > 1 && (size_t)th * W * 8 > 16 * 1024) th >>= 1;
  if (th > H) th = H;
  return th;
}

// rows per row-bin of the (view, face) table: a multiple of the strip height with <= 254 bins per image
static int bin_rows_for(int H, int TH) {
  int n_strips = (H + TH - 1) / TH;
  return TH * ((n_strips + 253) / 254);
}

struct MeshWs {
  long long *nacc;   // 3 * total_verts fixed-point normal sums
  float4 *vattr;     // 2 * total_verts
  float4 *proj;      // V * total_verts
  unsigned short *bins;  // N * bin_stride
  int bin_stride;
  float *partials;   // N * n_strips * 18
  int *soff;         // N * (n_strips + 1) strip-list offsets (pre-pass)
  int4 *slist;       // N * slist_cap (face, i0, i1, i2) entries
  int *sfront;       // N * n_strips: how many entries of each list are "near" faces (drawn first)
  float *vconst;     // N * 8 per-view shading constants
  int slist_cap;
  size_t bytes;
};

static MeshWs mesh_ws_layout(const mvr_mesh_batch *m, const mvr_mesh_params *p, void *base) {
  MeshWs w;
  int TH = p->strip_rows > 0 ? p->strip_rows : default_strip_rows(p->image_h, p->image_w);
  int n_strips = (p->image_h + TH - 1) / TH;
  size_t N = (size_t)m->n_meshes * p->n_views;
  size_t off = 0;
  w.nacc = (long long *)((char *)base + off); off += align_up((size_t)3 * m->total_verts * sizeof(long long), 256);
  w.vattr = (float4 *)((char *)base + off); off += align_up((size_t)2 * m->total_verts * sizeof(float4), 256);
  w.proj = (float4 *)((char *)base + off); off += align_up((size_t)p->n_views * m->total_verts * sizeof(float4), 256);
  w.bin_stride = (int)align_up((size_t)(m->max_faces > 0 ? m->max_faces : m->total_faces), 8);  // 8-byte vector reads need a multiple of 4
  w.bins = (unsigned short *)((char *)base + off); off += align_up(N * (size_t)w.bin_stride * sizeof(unsigned short), 256);
  w.partials = (float *)((char *)base + off); off += align_up(N * n_strips * MVR_NACC * sizeof(float), 256);
  // strip lists: a face reaches 1.5 strips on average at the reference's sizes; 3 per face + slack for tiny meshes of
  // screen-filling faces.  Views that need more fall back to the scan (decided on the device, per view).
  w.slist_cap = (int)align_up((size_t)3 * (m->max_faces > 0 ? m->max_faces : m->total_faces) + (size_t)16 * n_strips, 64);
  w.soff = (int *)((char *)base + off); off += align_up(N * (size_t)(n_strips + 1) * sizeof(int), 256);
  w.slist = (int4 *)((char *)base + off); off += align_up(N * (size_t)w.slist_cap * sizeof(int4), 256);
  w.sfront = (int *)((char *)base + off); off += align_up(N * (size_t)n_strips * sizeof(int), 256);
  w.vconst = (float *)((char *)base + off); off += align_up(N * 8 * sizeof(float), 256);
  w.bytes = off;
  return w;
}

static int check_mesh_args(const mvr_mesh_batch *m, const mvr_mesh_params *p) {
  if (!m || !p) return set_err(MVR_ERR_INVALID_ARG, "null mesh batch / params");
  if (m->n_meshes <= 0 || m->total_verts <= 0 || m->total_faces <= 0 || m->max_faces <= 0)
    return set_err(MVR_ERR_INVALID_ARG, "empty mesh batch");
  if (!m->verts || !m->faces || !m->vert_off || !m->face_off) return set_err(MVR_ERR_INVALID_ARG, "null mesh pointer");
  if (p->n_views <= 0 || p->image_h <= 0 || p->image_w <= 0) return set_err(MVR_ERR_INVALID_ARG, "bad view count / image size");
  if (p->image_h > 32767 || p->image_w > 32767) return set_err(MVR_ERR_UNSUPPORTED, "image side > 32767");
  if (p->image_h != p->image_w)  // MVTN only renders square images (models/renderer.py:90); PyTorch3D's non-square NDC rule differs
    return set_err(MVR_ERR_UNSUPPORTED, "non-square images are not supported");
  size_t N = (size_t)m->n_meshes * p->n_views;
  if (p->n_light != 1 && (size_t)p->n_light != N) return set_err(MVR_ERR_INVALID_ARG, "n_light must be 1 or B*V");
  return MVR_OK;
}

extern "C" {

#ifdef MVR_DIAG
int mvr_diag_set_buffer(void *dev_ptr) {
  unsigned long long *p = (unsigned long long *)dev_ptr;
  return hip_check(hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &p, sizeof(p)), "mvr_diag_set_buffer");
}
#endif

int mvr_version(void) { return 100; }

int mvr_selftest_rcp(uint32_t lo_bits, uint32_t hi_bits, uint32_t *n_bad, uint32_t *first_bad, mvr_stream_t stream) {
  if (!n_bad || !first_bad || hi_bits < lo_bits) return set_err(MVR_ERR_INVALID_ARG, "mvr_selftest_rcp: bad argument");
  HIP_TRY(hipMemsetAsync(n_bad, 0, sizeof(uint32_t), (hipStream_t)stream));
  hipLaunchKernelGGL(k_selftest_rcp, dim3(4096), dim3(256), 0, (hipStream_t)stream, lo_bits, hi_bits, n_bad, first_bad);
  return hip_check(hipGetLastError(), "k_selftest_rcp");
}

int mvr_profile_begin(void) {
  for (int k = 0; k < PROF_KINDS; ++k) {
    for (auto &p : g_prof_ev[k]) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    g_prof_ev[k].clear();
  }
  g_prof_on = true;
  return MVR_OK;
}

int mvr_profile_end(double *total_ms, int32_t *launches) {
  g_prof_on = false;
  for (int k = 0; k < PROF_KINDS; ++k) {
    double sum = 0.0;
    int cnt = 0;
    for (auto &p : g_prof_ev[k]) {
      float ms = 0.f;
      if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { sum += ms; ++cnt; }
      (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second);
    }
    g_prof_ev[k].clear();
    if (total_ms) total_ms[k] = sum;
    if (launches) launches[k] = cnt;
  }
  return MVR_OK;
}
const char *mvr_last_error(void) { return g_err; }

int mvr_look_at_checked(const float *azim, const float *elev, const float *dist, int32_t n, int32_t max_trials,
                        uint64_t seed, float *R, float *T, float *azim_used, float *elev_used, int32_t *view_valid,
                        int32_t *status, mvr_stream_t stream) {
  if (!azim || !elev || !dist || !R || !T || n <= 0) return set_err(MVR_ERR_INVALID_ARG, "mvr_look_at: bad argument");
  hipLaunchKernelGGL(k_look_at, dim3(1), dim3(MVR_LA_BLOCK), 0, (hipStream_t)stream, azim, elev, dist, n, max_trials,
                     (unsigned long long)seed, R, T, azim_used, elev_used, view_valid, status);
  return hip_check(hipGetLastError(), "k_look_at");
}

int mvr_look_at(const float *azim, const float *elev, const float *dist, int32_t n, float *R, float *T,
                int32_t *view_valid, int32_t *status, mvr_stream_t stream) {
  return mvr_look_at_checked(azim, elev, dist, n, -1, 0ull, R, T, nullptr, nullptr, view_valid, status, stream);
}

int mvr_look_at_backward(const float *azim, const float *elev, const float *dist, int32_t n, const float *grad_R,
                         const float *grad_T, float *grad_azim, float *grad_elev, float *grad_dist, mvr_stream_t stream) {
  if (!azim || !elev || !dist || !grad_azim || !grad_elev || !grad_dist || n <= 0)
    return set_err(MVR_ERR_INVALID_ARG, "mvr_look_at_backward: bad argument");
  hipLaunchKernelGGL(k_look_at_bwd, dim3((n + MVR_BLOCK - 1) / MVR_BLOCK), dim3(MVR_BLOCK), 0, (hipStream_t)stream, azim, elev,
                     dist, n, grad_R, grad_T, grad_azim, grad_elev, grad_dist);
  return hip_check(hipGetLastError(), "k_look_at_bwd");
}

int mvr_camera_position(const float *azim, const float *elev, const float *dist, int32_t n, float *position, mvr_stream_t stream) {
  if (!azim || !elev || !dist || !position || n <= 0) return set_err(MVR_ERR_INVALID_ARG, "mvr_camera_position: bad argument");
  hipLaunchKernelGGL(k_camera_position, dim3((n + MVR_BLOCK - 1) / MVR_BLOCK), dim3(MVR_BLOCK), 0, (hipStream_t)stream, azim, elev,
                     dist, n, position);
  return hip_check(hipGetLastError(), "k_camera_position");
}

int mvr_camera_position_backward(const float *azim, const float *elev, const float *dist, int32_t n, const float *grad_position,
                                 float *grad_azim, float *grad_elev, float *grad_dist, mvr_stream_t stream) {
  if (!azim || !elev || !dist || !grad_position || !grad_azim || !grad_elev || !grad_dist || n <= 0)
    return set_err(MVR_ERR_INVALID_ARG, "mvr_camera_position_backward: bad argument");
  hipLaunchKernelGGL(k_camera_position_bwd, dim3((n + MVR_BLOCK - 1) / MVR_BLOCK), dim3(MVR_BLOCK), 0, (hipStream_t)stream, azim,
                     elev, dist, n, grad_position, grad_azim, grad_elev, grad_dist);
  return hip_check(hipGetLastError(), "k_camera_position_bwd");
}

size_t mvr_mesh_workspace_bytes(const mvr_mesh_batch *meshes, const mvr_mesh_params *params) {
  if (!meshes || !params) return 0;
  return mesh_ws_layout(meshes, params, nullptr).bytes;
}

static int launch_vattr(const mvr_mesh_batch *m, const MeshWs &ws, float *normals_out, hipStream_t st) {
  MeshPtrs mp = {m->verts, m->faces, m->vert_off, m->face_off, m->vert_colors, m->n_meshes};
  const int max_verts = m->max_verts > 0 ? m->max_verts : m->total_verts;
  const int chunks = (max_verts + MVR_NRM_LDS_VERTS - 1) / MVR_NRM_LDS_VERTS;
  if (chunks <= 16) {   // beyond that every workgroup would rescan all faces 16+ times: use the global-atomic path
    size_t lds = (size_t)3 * (max_verts < MVR_NRM_LDS_VERTS ? max_verts : MVR_NRM_LDS_VERTS) * sizeof(unsigned long long);
    if (lds > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute((const void *)k_vattr_mesh_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_vattr_mesh_lds, dim3(m->n_meshes * chunks), dim3(MVR_NRM_BLOCK), lds, st, mp, chunks, ws.vattr, normals_out);
    return hip_check(hipGetLastError(), "k_vattr_mesh_lds");
  }
  HIP_TRY(hipMemsetAsync(ws.nacc, 0, (size_t)3 * m->total_verts * sizeof(long long), st));
  hipLaunchKernelGGL(k_normals_accum, dim3((m->max_faces + MVR_BLOCK - 1) / MVR_BLOCK, m->n_meshes), dim3(MVR_BLOCK), 0, st, mp, ws.nacc);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_vattr_finalize, dim3((m->total_verts + MVR_BLOCK - 1) / MVR_BLOCK), dim3(MVR_BLOCK), 0, st, m->verts,
                     m->total_verts, ws.nacc, ws.vattr, normals_out);
  return hip_check(hipGetLastError(), "k_vattr_finalize");
}

int mvr_vertex_normals(const mvr_mesh_batch *meshes, float *normals, void *workspace, size_t workspace_bytes, mvr_stream_t stream) {
  if (!meshes || !normals || !workspace) return set_err(MVR_ERR_INVALID_ARG, "mvr_vertex_normals: bad argument");
  size_t nacc_b = align_up((size_t)3 * meshes->total_verts * sizeof(long long), 256);
  if (workspace_bytes < nacc_b + align_up((size_t)2 * meshes->total_verts * sizeof(float4), 256))
    return set_err(MVR_ERR_WORKSPACE, "mvr_vertex_normals: workspace too small");
  MeshWs ws;
  ws.nacc = (long long *)workspace;
  ws.vattr = (float4 *)((char *)workspace + nacc_b);
  return launch_vattr(meshes, ws, normals, (hipStream_t)stream);
}

static int mesh_forward_impl(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const float *R, const float *T,
                             const float *light_dir, float *images, int32_t *pix_to_face, void *workspace,
                             size_t workspace_bytes, mvr_stream_t stream, const LookAtFuse *fuse);

int mvr_render_meshes_forward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const float *R, const float *T,
                              const float *light_dir, float *images, int32_t *pix_to_face, void *workspace,
                              size_t workspace_bytes, mvr_stream_t stream) {
  return mesh_forward_impl(meshes, params, R, T, light_dir, images, pix_to_face, workspace, workspace_bytes, stream, nullptr);
}

static int mesh_forward_impl(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const float *R, const float *T,
                             const float *light_dir, float *images, int32_t *pix_to_face, void *workspace,
                             size_t workspace_bytes, mvr_stream_t stream, const LookAtFuse *fuse) {
  int rc = check_mesh_args(meshes, params);
  if (rc != MVR_OK) return rc;
  if (!R || !T || !light_dir || !images || !workspace) return set_err(MVR_ERR_INVALID_ARG, "mvr_render_meshes_forward: null pointer");
  MeshWs ws = mesh_ws_layout(meshes, params, workspace);
  if (workspace_bytes < ws.bytes) return set_err(MVR_ERR_WORKSPACE, "mvr_render_meshes_forward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int V = params->n_views, N = meshes->n_meshes * V, H = params->image_h, W = params->image_w;
  const int TH = params->strip_rows > 0 ? params->strip_rows : default_strip_rows(H, W);
  const int n_strips = (H + TH - 1) / TH;
  MeshPtrs mp = {meshes->verts, meshes->faces, meshes->vert_off, meshes->face_off, meshes->vert_colors, meshes->n_meshes};

  const float s = mvr::fov_scale(params->fov_deg);
  const int max_verts = meshes->max_verts > 0 ? meshes->max_verts : meshes->total_verts;
  const int max_faces = meshes->max_faces > 0 ? meshes->max_faces : meshes->total_faces;
  const int G = bin_rows_for(H, TH);
  const int cullf = (params->flags & MVR_CULL_BACKFACES) ? 1 : 0;
  StripLists sl = {nullptr, nullptr, nullptr, 0, 0};
  const float *vconst = nullptr;
  if (max_verts <= MVR_PB_LDS_VERTS) {
    // one pre-pass launch: per-mesh vertex attributes + per-view projection/binning side by side
    const int chunks = (max_verts + MVR_NRM_LDS_VERTS - 1) / MVR_NRM_LDS_VERTS;
    const size_t lds_attr = (size_t)3 * (max_verts < MVR_NRM_LDS_VERTS ? max_verts : MVR_NRM_LDS_VERTS) * sizeof(unsigned long long);
    const size_t lds_proj = (size_t)3 * max_verts * sizeof(float);
    const size_t lds_pp = lds_attr > lds_proj ? lds_attr : lds_proj;
    if (lds_pp > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute((const void *)k_mesh_prepass, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pp));
    if (G == TH && n_strips <= 254) sl = StripLists{ws.soff, ws.slist, ws.sfront, ws.slist_cap, n_strips};
    hipLaunchKernelGGL(k_mesh_prepass, dim3(meshes->n_meshes * chunks + N), dim3(1024), lds_pp, st, mp, chunks, V, H, W, G,
                       ws.bin_stride, cullf, R, T, s, ws.proj, ws.bins, ws.vattr, sl, ViewConsts{ws.vconst, light_dir, params->n_light},
                       fuse ? *fuse : LookAtFuse{nullptr, nullptr, nullptr, 0, 0ull, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0});
    vconst = ws.vconst;
    HIP_TRY(hipGetLastError());
  } else {
    if (fuse) {   // no fused pre-pass for this mesh size: the look-at runs as its own launch
      hipLaunchKernelGGL(k_look_at, dim3(1), dim3(MVR_LA_BLOCK), 0, st, fuse->azim, fuse->elev, fuse->dist, fuse->N, fuse->max_trials,
                         fuse->seed, fuse->R, fuse->T, fuse->azim_used, fuse->elev_used, fuse->view_valid, fuse->status, fuse->cam_pos);
      HIP_TRY(hipGetLastError());
    }
    rc = launch_vattr(meshes, ws, nullptr, st);
    if (rc != MVR_OK) return rc;
    if (max_verts <= MVR_PB_LDS_VERTS) {
      const size_t lds_pb = (size_t)3 * max_verts * sizeof(float);
      if (lds_pb > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)k_project_bins_view, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pb));
      hipLaunchKernelGGL(k_project_bins_view, dim3(N), dim3(MVR_PB_BLOCK), lds_pb, st, mp, V, H, W, G, ws.bin_stride, cullf, R, T, s,
                         ws.proj, ws.bins);
      HIP_TRY(hipGetLastError());
    } else {
      hipLaunchKernelGGL(k_project_verts, dim3((max_verts + MVR_BLOCK - 1) / MVR_BLOCK, N), dim3(MVR_BLOCK), 0, st, mp, V, R, T, s, ws.proj);
      HIP_TRY(hipGetLastError());
      hipLaunchKernelGGL(k_face_bins, dim3((max_faces + MVR_BLOCK - 1) / MVR_BLOCK, N), dim3(MVR_BLOCK), 0, st, mp, V, H, W, G,
                         ws.bin_stride, cullf, ws.proj, ws.bins);
      HIP_TRY(hipGetLastError());
    }
  }

  RasterArgs a;
  a.bins = ws.bins; a.bin_rows = G; a.bin_stride = ws.bin_stride; a.sl = sl; a.vconst = vconst;
  a.m = mp; a.proj = ws.proj; a.vattr = ws.vattr; a.R = R; a.T = T; a.light = light_dir;
  a.images = images; a.p2f = pix_to_face;
  a.V = V; a.N = N; a.H = H; a.W = W; a.TH = TH; a.n_strips = n_strips;
  a.views_per_xcd = (N + 7) / 8;
  a.n_light = params->n_light; a.flags = params->flags; a.s = s;
  a.ambient = params->ambient; a.diffuse = params->diffuse; a.specular = params->specular; a.shininess = params->shininess;
  for (int k = 0; k < 3; ++k) { a.obj[k] = params->obj_color[k]; a.bg[k] = params->bg_color[k]; }
  size_t lds = (size_t)TH * W * 8 + (size_t)MVR_RBATCH * (sizeof(FaceCoef) + sizeof(int2)) + (size_t)(MVR_RBATCH + 1) * 4 +
               (size_t)MVR_LIST_CAP * 4 + 32 * 4 + (size_t)(W + 1 + TH) * 4 + (size_t)((W + (1 << MVR_TILE_SHIFT) - 1) >> MVR_TILE_SHIFT) * 4;
#ifdef MVR_DIAG
  if (const char *pad = getenv("MVR_LDS_PAD")) lds += (size_t)atoi(pad);   // occupancy experiments only
#endif
  if (lds > 160 * 1024) return set_err(MVR_ERR_UNSUPPORTED, "strip does not fit in LDS; lower strip_rows");
  const bool persp = (params->flags & MVR_PERSPECTIVE_CORRECT) != 0;
  const void *kern = persp ? (const void *)k_mesh_raster_shade<true> : (const void *)k_mesh_raster_shade<false>;
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int grid = 8 * a.views_per_xcd * n_strips;
  {
    ProfScope prof(PROF_MESH_FWD, st);
    if (persp) hipLaunchKernelGGL(k_mesh_raster_shade<true>, dim3(grid), dim3(MVR_RB), lds, st, a);
    else hipLaunchKernelGGL(k_mesh_raster_shade<false>, dim3(grid), dim3(MVR_RB), lds, st, a);
  }
  return hip_check(hipGetLastError(), "k_mesh_raster_shade");
}

struct ViewsBwd { const float *azim, *elev, *dist, *azim_used, *elev_used; int relative; float *g_azim, *g_elev, *g_dist; };

static int mesh_backward_impl(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const float *R, const float *T,
                              const float *light_dir, const int32_t *pix_to_face, const float *grad_images, float *grad_R,
                              float *grad_T, float *grad_light, void *workspace, size_t workspace_bytes, mvr_stream_t stream,
                              const ViewsBwd *vb);

int mvr_render_mesh_views_forward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const float *azim, const float *elev,
                                  const float *dist, int32_t max_trials, uint64_t seed, const float *light_dir, float *R, float *T,
                                  float *azim_used, float *elev_used, float *light_used, int32_t *view_valid, int32_t *status,
                                  float *images, int32_t *pix_to_face, void *workspace, size_t workspace_bytes,
                                  mvr_stream_t stream) {
  int rc = check_mesh_args(meshes, params);
  if (rc != MVR_OK) return rc;
  if (!azim || !elev || !dist || !R || !T || !images || !workspace || (!light_dir && !light_used))
    return set_err(MVR_ERR_INVALID_ARG, "mvr_render_mesh_views_forward: null pointer");
  const int N = meshes->n_meshes * params->n_views;
  if (!light_dir && params->n_light != N) return set_err(MVR_ERR_INVALID_ARG, "relative light needs n_light == B*V");
  const LookAtFuse fuse = {azim, elev, dist, (int)max_trials, (unsigned long long)seed, R, T, azim_used, elev_used,
                           light_dir ? nullptr : light_used, view_valid, status, N};
  return mesh_forward_impl(meshes, params, R, T, light_dir ? light_dir : light_used, images, pix_to_face, workspace,
                           workspace_bytes, stream, &fuse);
}

int mvr_render_meshes_backward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const float *R, const float *T,
                               const float *light_dir, const int32_t *pix_to_face, const float *grad_images, float *grad_R,
                               float *grad_T, float *grad_light, void *workspace, size_t workspace_bytes, mvr_stream_t stream) {
  if (!grad_R || !grad_T) return set_err(MVR_ERR_INVALID_ARG, "mvr_render_meshes_backward: null pointer");
  return mesh_backward_impl(meshes, params, R, T, light_dir, pix_to_face, grad_images, grad_R, grad_T, grad_light, workspace,
                            workspace_bytes, stream, nullptr);
}

int mvr_render_mesh_views_backward(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const float *azim, const float *elev,
                                   const float *dist, const float *azim_used, const float *elev_used, const float *R,
                                   const float *T, const float *light, int32_t relative_light, const int32_t *pix_to_face,
                                   const float *grad_images, float *grad_azim, float *grad_elev, float *grad_dist,
                                   void *workspace, size_t workspace_bytes, mvr_stream_t stream) {
  if (!azim || !elev || !dist || !azim_used || !elev_used || !grad_azim || !grad_elev || !grad_dist)
    return set_err(MVR_ERR_INVALID_ARG, "mvr_render_mesh_views_backward: null pointer");
  const ViewsBwd vb = {azim, elev, dist, azim_used, elev_used, relative_light ? 1 : 0, grad_azim, grad_elev, grad_dist};
  return mesh_backward_impl(meshes, params, R, T, light, pix_to_face, grad_images, nullptr, nullptr, nullptr, workspace,
                            workspace_bytes, stream, &vb);
}

static int mesh_backward_impl(const mvr_mesh_batch *meshes, const mvr_mesh_params *params, const float *R, const float *T,
                              const float *light_dir, const int32_t *pix_to_face, const float *grad_images, float *grad_R,
                              float *grad_T, float *grad_light, void *workspace, size_t workspace_bytes, mvr_stream_t stream,
                              const ViewsBwd *vb) {
  int rc = check_mesh_args(meshes, params);
  if (rc != MVR_OK) return rc;
  if (!R || !T || !light_dir || !pix_to_face || !grad_images || !workspace)
    return set_err(MVR_ERR_INVALID_ARG, "mvr_render_meshes_backward: null pointer");
  MeshWs ws = mesh_ws_layout(meshes, params, workspace);
  if (workspace_bytes < ws.bytes) return set_err(MVR_ERR_WORKSPACE, "mvr_render_meshes_backward: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int V = params->n_views, N = meshes->n_meshes * V, H = params->image_h, W = params->image_w;
  // the backward keeps no depth keys in LDS: taller strips than the forward's amortise its compaction / reduction
  // phases over more pixels (the partials buffer is sized for the forward's, larger, strip count)
  int TH = params->strip_rows > 0 ? params->strip_rows : MVR_BWD_STRIP_MULT * default_strip_rows(H, W);
  if (TH > H) TH = H;
  const int n_strips = (H + TH - 1) / TH;
  MeshPtrs mp = {meshes->verts, meshes->faces, meshes->vert_off, meshes->face_off, meshes->vert_colors, meshes->n_meshes};
  // vertex attributes: reuse the forward's when the caller hands back the same workspace, else recompute
  if (!(params->flags & MVR_WORKSPACE_FROM_FORWARD)) {
    rc = launch_vattr(meshes, ws, nullptr, st);
    if (rc != MVR_OK) return rc;
  }
  MeshBwdArgs a;
  a.m = mp; a.vattr = ws.vattr; a.R = R; a.T = T; a.light = light_dir; a.p2f = pix_to_face; a.gimg = grad_images;
  a.partials = ws.partials;
  a.V = V; a.N = N; a.H = H; a.W = W; a.TH = TH; a.n_strips = n_strips; a.views_per_xcd = (N + 7) / 8;
  a.n_light = params->n_light; a.flags = params->flags; a.s = mvr::fov_scale(params->fov_deg);
  a.ambient = params->ambient; a.diffuse = params->diffuse; a.specular = params->specular; a.shininess = params->shininess;
  for (int k = 0; k < 3; ++k) a.obj[k] = params->obj_color[k];
  int grid = 8 * a.views_per_xcd * n_strips;
  const size_t lds_b = (size_t)TH * W * sizeof(int);
  if (lds_b > 150 * 1024) return set_err(MVR_ERR_UNSUPPORTED, "strip does not fit in LDS; lower strip_rows");
  if (lds_b > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void *)k_mesh_backward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
  {
    ProfScope prof(PROF_MESH_BWD, st);
    hipLaunchKernelGGL(k_mesh_backward, dim3(grid), dim3(MVR_BLOCK), lds_b, st, a);
  }
  HIP_TRY(hipGetLastError());
  const int views_per_block = MVR_BLOCK / 64;
  if (vb) {
    hipLaunchKernelGGL(k_mesh_views_backward_finalize, dim3((N + MVR_BLOCK / 128 - 1) / (MVR_BLOCK / 128)), dim3(MVR_BLOCK), 0, st,
                       ws.partials, N, n_strips, R, T, vb->azim, vb->elev, vb->dist, vb->azim_used, vb->elev_used, vb->relative,
                       vb->g_azim, vb->g_elev, vb->g_dist, light_dir, params->n_light);
    return hip_check(hipGetLastError(), "k_mesh_views_backward_finalize");
  }
  hipLaunchKernelGGL(k_mesh_backward_finalize, dim3((N + views_per_block - 1) / views_per_block), dim3(MVR_BLOCK), 0, st,
                     ws.partials, N, n_strips, R, T, grad_R, grad_T, grad_light, light_dir, params->n_light);
  return hip_check(hipGetLastError(), "k_mesh_backward_finalize");
}

static int points_k(const mvr_points_params *p) { return p->points_per_pixel > 0 ? p->points_per_pixel : 1; }

static int points_strip_rows(const mvr_points_params *p) {
  if (p->strip_rows > 0) return p->strip_rows;
  int th = default_strip_rows(p->image_h, p->image_w);   // measured at C4: 6 / 7 / 8 / 10 / 16 rows -> 78 / 73 / 72 / 89 / 87 us
  return th > p->image_h ? p->image_h : th;
}

struct PointsWs {
  float4 *pv;            // N * pstride
  unsigned short *pbin;  // N * pstride
  float *partials;       // N * n_strips * 13
  int32_t *slots;        // N * H * W * K, only for points_per_pixel > 1 (used when the caller passes no pix_to_point)
  int pstride;
  size_t bytes;
};

static PointsWs points_ws_layout(const mvr_points_params *p, void *base) {
  PointsWs w;
  int TH = points_strip_rows(p);
  int n_strips = (p->image_h + TH - 1) / TH;
  size_t N = (size_t)p->n_clouds * p->n_views;
  w.pstride = (int)align_up((size_t)p->n_points, 8);
  size_t off = 0;
  w.pv = (float4 *)((char *)base + off); off += align_up(N * w.pstride * sizeof(float4), 256);
  w.pbin = (unsigned short *)((char *)base + off); off += align_up(N * w.pstride * sizeof(unsigned short), 256);
  w.partials = (float *)((char *)base + off); off += align_up(N * n_strips * MVR_PACC * sizeof(float), 256);
  w.slots = nullptr;
  if (points_k(p) > 1) {
    w.slots = (int32_t *)((char *)base + off);
    off += align_up(N * (size_t)p->image_h * p->image_w * points_k(p) * sizeof(int32_t), 256);
  }
  w.bytes = off;
  return w;
}

size_t mvr_points_workspace_bytes(const mvr_points_params *p) {
  if (!p) return 0;
  return points_ws_layout(p, nullptr).bytes;
}

static int check_points_args(const mvr_points_params *p) {
  if (!p) return set_err(MVR_ERR_INVALID_ARG, "null points params");
  if (p->n_clouds <= 0 || p->n_points <= 0 || p->n_views <= 0 || p->image_h <= 0 || p->image_w <= 0)
    return set_err(MVR_ERR_INVALID_ARG, "bad points params");
  if (!(p->radius > 0.0)) return set_err(MVR_ERR_INVALID_ARG, "radius must be > 0");
  if (p->points_per_pixel < 0 || p->points_per_pixel > 150 || p->reserved_ != 0)
    return set_err(MVR_ERR_INVALID_ARG, "points_per_pixel must be in [0, 150] and reserved_ 0");
  if (p->image_h > 32767 || p->image_w > 32767) return set_err(MVR_ERR_UNSUPPORTED, "image side > 32767");
  if (p->image_h != p->image_w) return set_err(MVR_ERR_UNSUPPORTED, "non-square images are not supported");
  return MVR_OK;
}

#define MVR_POINT_VIEWS_FUSED_MAX 2048
static int points_forward_impl(const mvr_points_params *p, const float *points, const float *point_colors, const float *R,
                               const float *T, const float *dist, float *images, int32_t *pix_to_point, void *workspace,
                               size_t workspace_bytes, mvr_stream_t stream, const LookAtFuse *fuse) {
  int rc = check_points_args(p);
  if (rc != MVR_OK) return rc;
  if (!points || !R || !T || !dist || !images || !workspace)
    return set_err(MVR_ERR_INVALID_ARG, "mvr_render_points_forward: null pointer");
  PointsWs ws = points_ws_layout(p, workspace);
  if (workspace_bytes < ws.bytes) return set_err(MVR_ERR_WORKSPACE, "mvr_render_points_forward: workspace too small");
  PointsArgs a;
  a.pv = ws.pv; a.pbin = ws.pbin; a.pstride = ws.pstride;
  a.points = points; a.colors = point_colors; a.R = R; a.T = T; a.dist = dist; a.images = images; a.p2p = pix_to_point;
  a.B = p->n_clouds; a.P = p->n_points; a.V = p->n_views; a.N = a.B * a.V; a.H = p->image_h; a.W = p->image_w;
  a.TH = points_strip_rows(p); a.n_strips = (a.H + a.TH - 1) / a.TH; a.views_per_xcd = (a.N + 7) / 8;
  a.radius = (float)p->radius; a.r2 = a.radius * a.radius; a.r2w = (float)(p->radius * p->radius);
  for (int k = 0; k < 3; ++k) { a.obj[k] = p->obj_color[k]; a.bg[k] = p->bg_color[k]; }
  a.bin_rows = bin_rows_for(a.H, a.TH);
  if (fuse) {
    const int vpw = (a.N + 255) / 256;   // <= MVR_POINTS_PREPASS_MAX_VPW (N <= MVR_POINT_VIEWS_FUSED_MAX)
    hipLaunchKernelGGL(k_points_prepass, dim3((a.N + vpw - 1) / vpw), dim3(MVR_PB_BLOCK), 0, (hipStream_t)stream, a, *fuse, vpw, ws.pv, ws.pbin);
  } else
    hipLaunchKernelGGL(k_points_project, dim3((a.P + 4 * MVR_BLOCK - 1) / (4 * MVR_BLOCK), a.N), dim3(MVR_BLOCK), 0, (hipStream_t)stream, a,
                       ws.pv, ws.pbin);
  HIP_TRY(hipGetLastError());
  const int K = points_k(p);
  const size_t list_ints = (size_t)a.TH * (a.W / 4) > MVR_PTS_LIST_INTS ? (size_t)a.TH * (a.W / 4) : MVR_PTS_LIST_INTS;   // hit list / slow-group list
  size_t lds = (size_t)a.TH * a.W * (K > 1 ? 16 : 8) + (size_t)(a.W + a.TH) * 4 + 16 + (K > 1 ? 0 : (list_ints + 4) * 4);
  if (lds > 160 * 1024) return set_err(MVR_ERR_UNSUPPORTED, "strip does not fit in LDS; lower strip_rows");
  const void *kern = K > 1 ? (const void *)k_points_raster_topk : (const void *)k_points_raster_shade;
  if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int grid = 8 * a.views_per_xcd * a.n_strips;
  {
    ProfScope prof(PROF_POINTS_FWD, (hipStream_t)stream);
    if (K > 1)
      hipLaunchKernelGGL(k_points_raster_topk, dim3(grid), dim3(MVR_BLOCK), lds, (hipStream_t)stream, a, K,
                         pix_to_point ? pix_to_point : ws.slots);
    else
      hipLaunchKernelGGL(k_points_raster_shade, dim3(grid), dim3(MVR_BLOCK), lds, (hipStream_t)stream, a);
  }
  return hip_check(hipGetLastError(), "k_points_raster_shade");
}

int mvr_render_points_forward(const mvr_points_params *p, const float *points, const float *point_colors, const float *R,
                              const float *T, const float *dist, float *images, int32_t *pix_to_point, void *workspace,
                              size_t workspace_bytes, mvr_stream_t stream) {
  return points_forward_impl(p, points, point_colors, R, T, dist, images, pix_to_point, workspace, workspace_bytes, stream, nullptr);
}

int mvr_render_point_views_forward(const mvr_points_params *p, const float *points, const float *point_colors, const float *azim,
                                   const float *elev, const float *dist, int32_t max_trials, uint64_t seed, float *R, float *T,
                                   float *azim_used, float *elev_used, int32_t *view_valid, int32_t *status, float *images,
                                   int32_t *pix_to_point, void *workspace, size_t workspace_bytes, mvr_stream_t stream) {
  if (!azim || !elev || !dist || !R || !T) return set_err(MVR_ERR_INVALID_ARG, "mvr_render_point_views_forward: null pointer");
  if (p && (long long)p->n_clouds * p->n_views > MVR_POINT_VIEWS_FUSED_MAX)   // every workgroup evaluates all views' look-at
    return set_err(MVR_ERR_UNSUPPORTED, "mvr_render_point_views_forward: too many views for the fused look-at; use mvr_look_at_checked + mvr_render_points_forward");
  const LookAtFuse fuse = {azim, elev, dist, (int)max_trials, (unsigned long long)seed, R, T, azim_used, elev_used,
                           nullptr, view_valid, status, p ? p->n_clouds * p->n_views : 0};
  return points_forward_impl(p, points, point_colors, R, T, dist, images, pix_to_point, workspace, workspace_bytes, stream, &fuse);
}

int mvr_render_points_backward(const mvr_points_params *p, const float *points, const float *point_colors, const float *R,
                               const float *T, const float *dist, const int32_t *pix_to_point, const float *grad_images,
                               float *grad_R, float *grad_T, float *grad_dist, void *workspace, size_t workspace_bytes,
                               mvr_stream_t stream) {
  int rc = check_points_args(p);
  if (rc != MVR_OK) return rc;
  if (!points || !R || !T || !dist || !pix_to_point || !grad_images || !grad_R || !grad_T || !grad_dist || !workspace)
    return set_err(MVR_ERR_INVALID_ARG, "mvr_render_points_backward: null pointer");
  if (workspace_bytes < mvr_points_workspace_bytes(p)) return set_err(MVR_ERR_WORKSPACE, "mvr_render_points_backward: workspace too small");
  PointsBwdArgs a;
  a.points = points; a.colors = point_colors; a.R = R; a.T = T; a.dist = dist; a.gimg = grad_images; a.p2p = pix_to_point;
  a.partials = points_ws_layout(p, workspace).partials;
  a.B = p->n_clouds; a.P = p->n_points; a.V = p->n_views; a.N = a.B * a.V; a.H = p->image_h; a.W = p->image_w;
  a.TH = points_strip_rows(p); a.n_strips = (a.H + a.TH - 1) / a.TH; a.views_per_xcd = (a.N + 7) / 8;
  a.r2w = (float)(p->radius * p->radius);
  a.K = points_k(p);
  for (int k = 0; k < 3; ++k) a.obj[k] = p->obj_color[k];
  hipStream_t st = (hipStream_t)stream;
  int grid = 8 * a.views_per_xcd * a.n_strips;
  hipLaunchKernelGGL(k_points_backward, dim3(grid), dim3(MVR_BLOCK), 0, st, a);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_points_backward_finalize, dim3((a.N + MVR_BLOCK - 1) / MVR_BLOCK), dim3(MVR_BLOCK), 0, st, a.partials, a.N,
                     a.n_strips, dist, grad_R, grad_T, grad_dist);
  return hip_check(hipGetLastError(), "k_points_backward_finalize");
}

}  // extern "C"
