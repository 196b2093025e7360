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
