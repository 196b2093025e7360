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
 *   normals_three_cross: 0 (default) = ONE cross(v2 - v1, v0 - v1) per face added to its three vertices (what
 *       Meshes._compute_vertex_normals does in 0.6+ as this restatement recalls it); 1 = three different cross products,
 *       (v2-v1)x(v0-v1) -> vertex 1, (v0-v2)x(v1-v2) -> vertex 2, (v1-v0)x(v2-v0) -> vertex 0, in that order (the form the
 *       round-2 review recalls).  Equal in exact arithmetic, ~1e-7 apart in fp32.
 *   fov_scale_override: 0 (default) = s = RN(1 / tan(fov / 2)) evaluated in double; > 0: used as s as given — a PyTorch3D
 *       build computes K[0][0] from an fp32 torch.tan whose last bit depends on the device's tan (torch CPU here:
 *       0x1.bb67b0p+0, this restatement: 0x1.bb67aep+0, numpy's tanf: 0x1.bb67acp+0 for fov = 60). */
static int orc_normals_three_cross = 0;
static double orc_fov_scale_override = 0.0;
void orc_set_options(int normals_three_cross, double fov_scale_override) {
  orc_normals_three_cross = normals_three_cross ? 1 : 0;
  orc_fov_scale_override = fov_scale_override > 0.0 ? fov_scale_override : 0.0;
}

#define REAL float
#define SUFFIX _f32
#define SQRT_FN sqrtf
#define FABS_FN fabsf
#define POW_FN powf
#include "mvr_oracle_impl.inc"
#undef REAL
#undef SUFFIX
#undef SQRT_FN
#undef FABS_FN
#undef POW_FN

#define REAL double
#define SUFFIX _f64
#define SQRT_FN sqrt
#define FABS_FN fabs
#define POW_FN pow
#include "mvr_oracle_impl.inc"
#undef REAL
#undef SUFFIX
#undef SQRT_FN
#undef FABS_FN
#undef POW_FN

int orc_version(void) { return 1; }
