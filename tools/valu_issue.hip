// developer micro-benchmark: how many cycles does one wave64 instruction occupy a SIMD of gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_issue tools/valu_issue.hip ; on the GPU box: tools/valu_issue [out.csv]
// One workgroup per CU (96 KB of LDS each, so that no CU takes two), 256 * W threads = W waves per SIMD, W = 1, 2, 4 (8 for a few).
// Every wave issues N instructions of one kind between two s_memtime reads — independent ones (8 accumulator registers /
// register pairs in rotation: dependent distance 8) or one dependent chain.  Reported: cycles per instruction per wave
// (median and slowest wave of the launch), cycles per instruction per SIMD = that / W, and the shader clock derived from
// s_memrealtime (100 MHz) over the same interval.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float v2f __attribute__((ext_vector_type(2)));

// every kind: NAME, the asm of slot i (a##i: f32 accumulators, p##i: f32 pairs, u##i: u32, q##i: u64; b, c: f32 constants; pb, pc: pairs)
#define KINDS(K) \
  K(0, "v_fma_f32", asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "v"(b), "v"(c))) \
  K(1, "v_pk_fma_f32", asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(P) : "v"(pb), "v"(pc))) \
  K(2, "v_pk_mul_f32", asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(P) : "v"(pb))) \
  K(3, "v_pk_add_f32", asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(P) : "v"(pc))) \
  K(4, "v_add_f32", asm volatile("v_add_f32 %0, %0, %1" : "+v"(A) : "v"(c))) \
  K(5, "v_mul_f32", asm volatile("v_mul_f32 %0, %0, %1" : "+v"(A) : "v"(b))) \
  K(6, "v_fma_f32 (one dependent chain)", asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c))) \
  K(7, "v_pk_fma_f32 (one dependent chain)", asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pb), "v"(pc))) \
  K(8, "v_min3_f32", asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(A) : "v"(b), "v"(c))) \
  K(9, "v_cndmask_b32 (vcc)", asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A) : "v"(b))) \
  K(10, "v_rcp_f32", asm volatile("v_rcp_f32 %0, %0" : "+v"(A))) \
  K(11, "v_mov_b32", asm volatile("v_mov_b32 %0, %1" : "+v"(A) : "v"(b))) \
  K(12, "v_add_u32", asm volatile("v_add_u32 %0, %0, %1" : "+v"(U) : "v"(u7))) \
  K(13, "v_lshl_add_u32", asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(U) : "v"(u7))) \
  K(14, "v_cndmask_b32_e64 (sgpr pair)", asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(A) : "v"(b), "s"(smask))) \
  K(15, "v_cmp_lt_f32 -> vcc", asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(A), "v"(b) : "vcc")) \
  K(16, "v_cmp_lt_f32_e64 -> sgpr pair", asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm2) : "v"(A), "v"(b))) \
  K(17, "v_fma_f32 (sgpr operand)", asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(A) : "s"(sb), "v"(c))) \
  K(18, "v_mov_b32_dpp row_shr:1", asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(A))) \
  K(19, "v_readlane_b32", asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(sr) : "v"(A))) \
  K(20, "v_mad_u32_u24", asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(U) : "v"(u7))) \
  K(21, "v_mul_lo_u32", asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(U) : "v"(u7))) \
  K(22, "v_lshlrev_b64", asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(Q))) \
  K(23, "v_max_f32", asm volatile("v_max_f32 %0, %0, %1" : "+v"(A) : "v"(b))) \
  K(24, "v_sub_f32", asm volatile("v_sub_f32 %0, %0, %1" : "+v"(A) : "v"(c))) \
  K(25, "v_and_b32", asm volatile("v_and_b32 %0, %0, %1" : "+v"(U) : "v"(u7))) \
  K(26, "v_bfe_u32", asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(U))) \
  K(27, "v_cvt_f32_i32", asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(A))) \
  K(28, "v_rsq_f32", asm volatile("v_rsq_f32 %0, %0" : "+v"(A))) \
  K(29, "v_med3_f32", asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(A) : "v"(b), "v"(c))) \
  K(30, "v_add_f32 + v_mul_f32 alternating", asm volatile("v_add_f32 %0, %0, %1\n v_mul_f32 %2, %2, %3" : "+v"(A), "+v"(U) : "v"(c), "v"(b))) \
  K(31, "s_add_u32", asm volatile("s_add_u32 %0, %0, 3" : "+s"(sr) : : "scc")) \
  K(32, "s_and_b64", asm volatile("s_and_b64 %0, %0, %1" : "+s"(sm2) : "s"(smask) : "scc")) \
  K(33, "v_fma_f32 + s_add_u32 alternating", asm volatile("v_fma_f32 %0, %0, %2, %3\n s_add_u32 %1, %1, 3" : "+v"(A), "+s"(sr) : "v"(b), "v"(c) : "scc")) \
  K(34, "ds_read_b32", asm volatile("ds_read_b32 %0, %1" : "=v"(A) : "v"(laddr) : "memory")) \
  K(35, "ds_read_b128", asm volatile("ds_read_b128 %0, %1" : "=v"(F4) : "v"(laddr16) : "memory")) \
  K(36, "ds_write_b32", asm volatile("ds_write_b32 %0, %1" : : "v"(laddr), "v"(A) : "memory")) \
  K(37, "ds_max_rtn_u64 (distinct addresses)", asm volatile("ds_max_rtn_u64 %0, %1, %2" : "=v"(Q) : "v"(laddr8), "v"(q7) : "memory")) \
  K(38, "ds_max_u64 no return (distinct addresses)", asm volatile("ds_max_u64 %0, %1" : : "v"(laddr8), "v"(q7) : "memory")) \
  K(39, "ds_add_rtn_u32 (one address per wave)", asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(U) : "v"(laddr_same), "v"(u7) : "memory")) \
  K(40, "v_cmp_lt_f32 vcc + v_cndmask_b32", asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A) : "v"(b) : "vcc")) \
  K(41, "v_pk_fma_f32 + v_fma_f32 alternating", asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %5" : "+v"(P), "+v"(A) : "v"(pb), "v"(pc), "v"(b), "v"(c))) \
  K(42, "v_permlane32_swap", asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(A), "+v"(U)))
#define N_KINDS 43

template <int KIND>
__global__ void __launch_bounds__(1024) k_issue(long long *out, int iters, float seed, unsigned long long smask_in) {
  extern __shared__ unsigned char lds[];
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  v2f p0 = {seed, seed}, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f, p4 = p0 + 4.0f, p5 = p0 + 5.0f, p6 = p0 + 6.0f, p7 = p0 + 7.0f;
  const float b = 1.0000001f, c = 1e-9f;
  const v2f pb = {b, b}, pc = {c, c};
  unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7;
  unsigned long long q0 = u0, q1 = u1, q2 = u2, q3 = u3, q4 = u4, q5 = u5, q6 = u6, q7 = u7;
  float4 f0 = {0, 0, 0, 0}, f1 = f0, f2 = f0, f3 = f0, f4 = f0, f5 = f0, f6 = f0, f7 = f0;
  const unsigned long long smask = __builtin_amdgcn_readfirstlane((unsigned)smask_in) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(smask_in >> 32)) << 32);
  unsigned long long sm2 = smask;
  const float sb = __builtin_amdgcn_readfirstlane(__float_as_int(b)) ? b : c;
  unsigned sr = __builtin_amdgcn_readfirstlane(iters);
  const unsigned laddr = threadIdx.x * 4u, laddr16 = threadIdx.x * 16u, laddr8 = threadIdx.x * 8u, laddr_same = (threadIdx.x >> 6) * 4u;
  __syncthreads();
  const long long r0 = __builtin_amdgcn_s_memrealtime();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {   // 8 x 8 = 64 slots per iteration
#define K(id, name, stmt) \
      if (KIND == id) { \
        { auto &A = a0; auto &P = p0; auto &U = u0; auto &Q = q0; auto &F4 = f0; stmt; } \
        { auto &A = a1; auto &P = p1; auto &U = u1; auto &Q = q1; auto &F4 = f1; stmt; } \
        { auto &A = a2; auto &P = p2; auto &U = u2; auto &Q = q2; auto &F4 = f2; stmt; } \
        { auto &A = a3; auto &P = p3; auto &U = u3; auto &Q = q3; auto &F4 = f3; stmt; } \
        { auto &A = a4; auto &P = p4; auto &U = u4; auto &Q = q4; auto &F4 = f4; stmt; } \
        { auto &A = a5; auto &P = p5; auto &U = u5; auto &Q = q5; auto &F4 = f5; stmt; } \
        { auto &A = a6; auto &P = p6; auto &U = u6; auto &Q = q6; auto &F4 = f6; stmt; } \
        { auto &A = a7; auto &P = p7; auto &U = u7; auto &Q = q7; auto &F4 = f7; stmt; } \
      }
      KINDS(K)
#undef K
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  const long long r1 = __builtin_amdgcn_s_memrealtime();
  const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y +
                  (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7) + (float)(q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7) + (float)sm2 + (float)sr +
                  f0.x + f1.y + f2.z + f3.w + f4.x + f5.y + f6.z + f7.w;
  if (s == 123.456f) lds[0] = 1;
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[2 * w] = t1 - t0;
    out[2 * w + 1] = r1 - r0;
  }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
static void run_kind(const char *name, int per_slot, long long *d_out, int n_cu, FILE *csv) {
  const int iters = 1000;
  const double n_inst = 64.0 * iters * per_slot;
  for (int W : {1, 2, 4}) {
    const int threads = 256 * W;
    const int waves = n_cu * threads / 64;
    CHECK(hipFuncSetAttribute((const void *)k_issue<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_issue<KIND><<<n_cu, threads, 96 * 1024>>>(d_out, iters, 1.0f, 0x5555aaaa3333ccccull);   // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_issue<KIND><<<n_cu, threads, 96 * 1024>>>(d_out, iters, 1.0f, 0x5555aaaa3333ccccull);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(2 * waves);
    CHECK(hipMemcpy(h.data(), d_out, sizeof(long long) * 2 * waves, hipMemcpyDeviceToHost));
    std::vector<double> cyc(waves), ghz(waves);
    for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w]; ghz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] * 10.0); }   // s_memrealtime: 100 MHz
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double med = cyc[waves / 2], mx = cyc[waves - 1];
    printf("%-42s W=%d  cyc/inst/wave med %7.3f max %7.3f  cyc/inst/SIMD %7.3f  (%.3f GHz, %.0f us)\n", name, W, med / n_inst, mx / n_inst,
           med / n_inst / W, ghz[waves / 2], ms * 1e3);
    if (csv) fprintf(csv, "\"%s\",%d,%.4f,%.4f,%.4f,%.4f,%.2f\n", name, W, med / n_inst, mx / n_inst, med / n_inst / W, ghz[waves / 2], ms * 1e3);
  }
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clockRate %d kHz; one 256*W-thread workgroup per CU, 64 000 instruction slots per wave\n", prop.gcnArchName, n_cu, prop.clockRate);
  long long *d_out;
  CHECK(hipMalloc(&d_out, sizeof(long long) * 2 * n_cu * 16));
  FILE *csv = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (csv) fprintf(csv, "instruction,waves_per_simd,cycles_per_inst_per_wave_median,cycles_per_inst_per_wave_max,cycles_per_inst_per_simd,clock_ghz,kernel_us\n");
#define K(id, name, stmt) run_kind<id>(name, (id == 30 || id == 33 || id == 40 || id == 41) ? 2 : 1, d_out, n_cu, csv);
  KINDS(K)
#undef K
  if (csv) fclose(csv);
  return 0;
}
