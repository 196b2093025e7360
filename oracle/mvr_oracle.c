/*
 * oracle/mvr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Builds the fp32 and fp64 instances of the CPU restatement in
 * mvr_oracle_impl.inc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the resulting liboracle.so; the product
 * (mvtn_amd/) never does.  ** parity unpinned ** (see the .inc header).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* threads used by the literal pixel-major loops (mode 0); 1 = faithful single-threaded stand-in */
static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n > 1 ? n : 1; }

/* Switches for the two places where upstream's op order / precision is uncertain (Weak 1; RECALLED semantics):
 *   normals_three_cross: 1 (default since round 4) = three different cross products per face, (v2-v1)x(v0-v1) -> vertex 1,
 *       (v0-v2)x(v1-v2) -> vertex 2, (v1-v0)x(v2-v0) -> vertex 0, in that order: Meshes._compute_vertex_normals of the
 *       PyTorch3D releases of the reference's era (its README pins pytorch 1.7.1 -> PyTorch3D 0.3 / 0.4; both reviews recall this
 *       form for 0.2 - 0.7); 0 = ONE cross(v2 - v1, v0 - v1) per face added to its three vertices (the later form).
 *       Equal in exact arithmetic, ~1e-7 apart in fp32 (colours only, never coverage).
 *   fov_scale_override: 0 (default) = s = RN(1 / tan(fov / 2)) evaluated in double; > 0: used as s as given — a PyTorch3D
 *       build computes K[0][0] from an fp32 torch.tan whose last bit depends on the device's tan (torch CPU here:
 *       0x1.bb67b0p+0, this restatement: 0x1.bb67aep+0, numpy's tanf: 0x1.bb67acp+0 for fov = 60). */
static int orc_normals_three_cross = 1;
static double orc_fov_scale_override = 0.0;
/*   edge_fma (round 4, VERDICT r3 missing 1b): 0 (default) = the per-pixel test as the plain IEEE sequence of PyTorch3D's C++ source
 *       (what its CPU rasteriser executes with contraction off); 1 = the same expressions CONTRACTED the way nvcc (-fmad=true,
 *       the default) compiles them for the CUDA kernels the reference actually runs (models/renderer.py:67-68,77,85 hard-code
 *       .cuda()): a b - c d -> fma(a, b, -RN(c d)) in EdgeFunctionForward, the sum of the perspective-correction numerators and
 *       the interpolated depth as chains of FMAs (the first product of every sum is rounded, the later ones fused).  One
 *       plausible contraction, not a disassembly: it measures how many pixels such a difference can move. */
static int orc_edge_fma = 0;
void orc_set_options(int normals_three_cross, double fov_scale_override) {
  orc_normals_three_cross = normals_three_cross ? 1 : 0;
  orc_fov_scale_override = fov_scale_override > 0.0 ? fov_scale_override : 0.0;
}
void orc_set_edge_fma(int on) { orc_edge_fma = on ? 1 : 0; }
/*   normals_exact_sum (round 5): 0 (default) = the vertex-normal sums as an index_add in face order; 1 = order-independent
 *       64-bit fixed-point sums (see orc_vertex_normals).  Differs visibly only at vertices whose face normals cancel. */
static int orc_normals_exact_sum = 0;
void orc_set_normals_exact_sum(int on) { orc_normals_exact_sum = on ? 1 : 0; }
/*   projection_bmm (round 6): 0 (default) = world -> view as the closed form ((x r0 + y r1) + z r2) + t, every
 *       product and sum rounded — what the library computes; 1 = the same sum as a K = 4 GPU matrix product accumulates it
 *       (ascending k, fused multiply-adds): PyTorch3D projects vertices through Transform3d.transform_points = torch.bmm.  Which
 *       of two COINCIDENT faces (a shell and its reversed copy) has the smaller interpolated depth is decided in the last bit of
 *       pz, i.e. by the rounding sequence of the projection: this switch measures how many pixels that moves. */
static int orc_projection_bmm = 0;
void orc_set_projection_bmm(int on) { orc_projection_bmm = on ? 1 : 0; }
/* soft path, round 4: which PIECE of a cut face every fragment of the K-nearest lists lies on (clip_face's index; -1: the face
 * is whole) — written by the next orc_render_meshes_soft_* call when `out` is set ((N,H,W,K) int32), and, with a fixed list,
 * read from `fixed` so that the listed fragments are evaluated on the same pieces again (finite differences of the clipped
 * rendering).  Both NULL (the default) = the behaviour of rounds 2-3: fixed lists evaluate whole faces. */
static int32_t *orc_soft_piece_out = NULL;
static const int32_t *orc_soft_fixed_piece = NULL;
void orc_set_soft_pieces(int32_t *out, const int32_t *fixed) { orc_soft_piece_out = out; orc_soft_fixed_piece = fixed; }

#define REAL float
#define SUFFIX _f32
#define SQRT_FN sqrtf
#define FABS_FN fabsf
#define POW_FN powf
#define FMA_FN fmaf
#include "mvr_oracle_impl.inc"
#undef REAL
#undef SUFFIX
#undef SQRT_FN
#undef FABS_FN
#undef POW_FN
#undef FMA_FN

#define REAL double
#define SUFFIX _f64
#define SQRT_FN sqrt
#define FABS_FN fabs
#define POW_FN pow
#define FMA_FN fma
#include "mvr_oracle_impl.inc"
#undef REAL
#undef SUFFIX
#undef SQRT_FN
#undef FABS_FN
#undef POW_FN
#undef FMA_FN

int orc_version(void) { return 1; }
