// mvr_kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4) + the C ABI of include/mvr.h.
//
// Path: models/renderer.py MVRenderer.forward (reference :173-198) -> render_meshes (:65-114) /
// render_points (:116-151) and their autograd backward.  Design:
//
//   mesh forward = 2 launches on the caller's stream (3 through the fine-grained entry points, which run k_look_at first)
//     k_mesh_prepass       : workgroups [0, B): world-space vertex normals once per MESH (64-bit fixed-point LDS sums);
//                            workgroups [B, B+N): one VIEW each: look-at of all views (R, T, validity, device-side
//                            remedy, relative-light camera positions), project the vertices (16 B records, L2 resident),
//                            one LDS word per vertex (row range + depth flags), 2-byte row-bin range per face, per-strip
//                            face lists (near faces first; faces cut by the z-clip plane marked), shading constants
//     k_mesh_raster_fast   : one workgroup per (view, strip of 8 rows), heaviest strips first (mvr_mesh_raster_fast.inc):
//         coverage from the SIGNS of the exact edge functions, depth order from an affine inverse depth whose order is
//         proven per pixel (64-bit LDS atomicMax with the previous occupant returned); what it cannot prove — a few pixels
//         per frame — is settled by the reference's exact test in the same launch, a strip it cannot take at all (coincident
//         faces, faces to cut) by the exact body below; winners' shading planes in an LDS table, planar (B,V,3,H,W) stores.
//     k_mesh_raster_shade  : the exact-arithmetic rasteriser (round 1; MVR_RASTER_EXACT_ONLY, per-vertex colours, and the
//         body the fast kernel's workgroups fall back to): packed-f32 exact inside test / barycentrics / depth, 64-bit LDS
//         atomicMin on (float_bits(z) << 32 | face) — exactly PyTorch3D's (z, face index) order; cuts faces at the clip plane.
//     No per-pixel fragment tensors, no bin tables, no V-fold mesh copies ever touch HBM.
//   mesh backward = k_mesh_backward_tab (per-face data once per strip in an LDS table, 18 partial sums per workgroup; starts
//                   from the forward's pixel lists when it has them) [k_mesh_backward: per-pixel gathers] + one finalize launch
//   soft shaders  = k_mesh_soft_forward / _backward (top-K fragments, softmax / sigmoid blend, coverage gradient)
//   points        = k_points_prepass (look-at + projection; k_points_project when R, T are given) + k_points_raster_shade
//                   (hits compacted, 16-byte stores) [k_points_raster_topk for points_per_pixel > 1]
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
static_assert(MVR_RB == 256, "the rasterisers' LDS layouts and their host-side sizes assume 4 waves per workgroup (128 / 512 measured: wrong results / faults)");
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
#ifndef MVR_FAST_WAVES_PER_SIMD
#define MVR_FAST_WAVES_PER_SIMD 4          // (5: 96 VGPRs and 32 256 B of LDS fit, measured 80.9 vs 77.4 us at C2: the 5th workgroup buys nothing, the spills cost)
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
enum { PROF_MESH_FWD = 0, PROF_POINTS_FWD = 1, PROF_MESH_BWD = 2, PROF_SOFT_FWD = 3, PROF_SOFT_BWD = 4, PROF_MESH_PREPASS = 5, PROF_KINDS = 8 };   // (6, 7: unused)
static bool g_prof_on = false;
static unsigned g_prof_mask = 0xDFu;   // kinds bracketed while profiling; the pre-pass (5) only when MVR_PROF_MASK asks: two event pairs back to back would sit in the gap the other kinds' timings see
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_ev[PROF_KINDS];

struct ProfScope {
  hipStream_t st; int kind; hipEvent_t stop; bool on;
  ProfScope(int kind_, hipStream_t st_) : st(st_), kind(kind_), stop(nullptr), on(g_prof_on && ((g_prof_mask >> kind_) & 1u)) {
    if (!on) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(a, st);
    stop = b;
    g_prof_ev[kind].push_back(std::make_pair(a, b));
  }
  ~ProfScope() { if (on) (void)hipEventRecord(stop, st); }
};

// Scheduling hints — wave priorities (s_setprio) in the mesh rasterisers, the start stagger (s_sleep) of the point rasteriser's first
// resident batch: timing only, never results.  -DMVR_NO_TUNING compiles them out; tests/test_gpu_mesh_forward.py builds that variant and
// checks that mesh and point renders at 128 / 224 / 448 pixels are bit-identical with and without them.
#ifdef MVR_NO_TUNING
#define MVR_SETPRIO(p) ((void)0)
#define MVR_SLEEP(n) ((void)0)
#else
#define MVR_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#define MVR_SLEEP(n) __builtin_amdgcn_s_sleep(n)
#endif

// The sections, in dependency order (one translation unit):
#include "mvr_cameras.inc"
#include "mvr_mesh_prepass.inc"
#include "mvr_mesh_raster.inc"
#include "mvr_mesh_raster_fast.inc"
#include "mvr_mesh_backward.inc"
#include "mvr_mesh_backward_tab.inc"
#include "mvr_mesh_soft.inc"
#include "mvr_points.inc"
#include "mvr_abi.inc"
