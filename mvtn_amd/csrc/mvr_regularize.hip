// The step right after the renderer in the reference's training loop (run_mvtn.py:186-187):
//   regualarize_rendered_views (ops.py:168-176) = dropout2d over whole views (a (B,V,1,1,1) keep mask on the
//   5-D tensor) followed, when augment_training, by applied_transforms (ops.py:138-146) on every image:
//   RandomHorizontalFlip (util.py:168-196) -> ReplicationPad2d(pad) -> RandomCrop(H) (util.py:199-242).
// The reference runs this as ~10 full-frame passes (clone, masked flip, pad to (H+2 pad)^2, advanced-index gather,
// permutes); here it is ONE gather pass forward and one deterministic gather pass backward, both HBM-bound:
//   out[n,c,i,j] = scale[n] * in[n,c, clamp(i + ci[n] - pad), flip_n(clamp(j + cj[n] - pad))]
// The random draws (mask, flips, crop offsets) are made by the caller with torch's generators in the
// reference's order, so a seeded run reproduces the reference's choices; the kernels only apply them.
#include <hip/hip_runtime.h>

#include "../../include/mvr.h"

extern "C" __attribute__((visibility("hidden"))) void mvr_set_error_(const char *msg);

namespace {

constexpr int kBlock = 256;

struct RegArgs {
  const float *scale;      // (N) 0 or 1/(1-p); null: no dropout
  const uint8_t *flip;     // (N) or null
  const int32_t *crop_i;   // (N) row offset into the padded image, 0..2 pad; null: centred (= no shift)
  const int32_t *crop_j;   // (N) column offset
  int32_t pad, C, H, W;
  int64_t rows;            // N * C * H
  // optional per-channel normalisation fused behind the regulariser: out = (regularised - mean[c]) / std[c] — the
  // `Normalize` of the reference's trainer (viewGCN/tools/Trainer_mvt.py:41-49: ImageNet mean 0.456, std 0.225), which
  // otherwise is two more full-frame passes.  The quotient is a true division: bit-identical to the torch expression.
  int32_t norm;
  float mean[4], stdv[4];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one thread = 4 consecutive output pixels of a row (W % 4 == 0) or 1 pixel (VEC = 1)
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_regularize_views_fwd(const float *__restrict__ in, float *__restrict__ out, RegArgs a) {
  const int groups = a.W / VEC;
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= a.rows * groups) return;
  const int xg = (int)(g % groups);
  const int64_t row = g / groups;
  const int y = (int)(row % a.H);
  const int64_t nc = row / a.H;
  const int n = (int)(nc / a.C);
  const float s = a.scale ? a.scale[n] : 1.0f;
  float v[VEC];
  if (a.scale && s == 0.0f) {   // a dropped view: nothing to read
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = 0.0f;
  } else {
    const int sy = clampi(y + (a.crop_i ? a.crop_i[n] - a.pad : 0), 0, a.H - 1);
    const int oj = a.crop_j ? a.crop_j[n] - a.pad : 0;
    const bool f = a.flip && a.flip[n];
    const float *src = in + (nc * a.H + sy) * a.W;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      int sx = clampi(xg * VEC + k + oj, 0, a.W - 1);
      sx = f ? a.W - 1 - sx : sx;
      const float x = src[sx];
      v[k] = a.scale ? x * s : x;
    }
  }
  if (a.norm) {
    const int c = (int)(nc % a.C);
    const float m = a.mean[c], sd = a.stdv[c];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = (v[k] - m) / sd;
  }
  float *dst = out + row * a.W + xg * VEC;
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    dst[0] = v[0];
  }
}

// Output rows / columns that read source index `s` under the clamp: [lo, hi] (empty when lo > hi).
__device__ __forceinline__ void preimage(int s, int off, int size, int &lo, int &hi) {
  lo = (s == 0) ? 0 : s - off;
  hi = (s == size - 1) ? size - 1 : s - off;
  lo = lo < 0 ? 0 : lo;
  hi = hi > size - 1 ? size - 1 : hi;
}

// Backward as a gather with a fixed summation order.  Interior source pixels have at most one reader; a pixel of
// the first / last row is read by up to pad+1 output rows (the whole wave walks them together: coalesced); a pixel
// of the first / last COLUMN is read by up to pad+1 output columns (corners: (pad+1)^2) — one lane of a wave would
// walk those alone, so k_regularize_views_bwd_columns gives each of them a whole wave and overwrites what this
// kernel stored there.  One thread = VEC consecutive source pixels of a row.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_regularize_views_bwd(const float *__restrict__ gout, float *__restrict__ gin, RegArgs a) {
  const int groups = a.W / VEC;
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= a.rows * groups) return;
  const int xg = (int)(g % groups);
  const int64_t row = g / groups;
  const int y = (int)(row % a.H);
  const int64_t nc = row / a.H;
  const int n = (int)(nc / a.C);
  const float s = a.scale ? a.scale[n] : 1.0f;
  float v[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) v[k] = 0.0f;
  if (!(a.scale && s == 0.0f)) {
    const bool f = a.flip && a.flip[n];
    const int oj = a.crop_j ? a.crop_j[n] - a.pad : 0;
    int ilo, ihi;
    preimage(y, a.crop_i ? a.crop_i[n] - a.pad : 0, a.H, ilo, ihi);
    const float *src = gout + nc * a.H * a.W;
    const float sd = a.norm ? a.stdv[(int)(nc % a.C)] : 1.0f;
    int jc[VEC];
    bool in[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int x = xg * VEC + k;
      const int xs = f ? a.W - 1 - x : x;   // the column of the flipped image this source pixel sits in
      const int j = xs - oj;                  // its only reader in an output row (first / last column: see above)
      in[k] = j >= 0 && j < a.W;
      jc[k] = clampi(j, 0, a.W - 1);          // loads stay unconditional (and in flight together)
    }
    int i = ilo;
    for (; i + 3 <= ihi; i += 4) {            // first / last source row: up to pad+1 reader rows, 4 rows of loads in flight
      float t[4][VEC];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < VEC; ++k) { t[r][k] = src[(int64_t)(i + r) * a.W + jc[k]]; if (a.norm) t[r][k] = t[r][k] / sd; }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] += in[k] ? t[r][k] : 0.0f;
    }
    for (; i <= ihi; ++i) {                   // every other source row has one reader row (or none)
      float t[VEC];
#pragma unroll
      for (int k = 0; k < VEC; ++k) { t[k] = src[(int64_t)i * a.W + jc[k]]; if (a.norm) t[k] = t[k] / sd; }
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[k] += in[k] ? t[k] : 0.0f;
    }
    if (a.scale) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) v[k] *= s;
    }
  }
  float *dst = gin + row * a.W + xg * VEC;
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4 *>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    dst[0] = v[0];
  }
}

// sum of p[0 .. nj) by 16 lanes, 8 loads in flight per lane, fixed order
__device__ __forceinline__ float row_sum16(const float *__restrict__ p, int nj, int lane, bool norm, float sd) {
  float t = 0.0f;
  for (int e0 = 0; e0 < nj; e0 += 128) {
    float u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e = e0 + lane + 16 * k;
      u[k] = e < nj ? p[e] : 0.0f;
      if (norm) u[k] = u[k] / sd;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) t += u[k];
  }
  return t;
}

__device__ __forceinline__ float reduce16(float x) {
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) x += __shfl_down(x, off, 16);
  return x;
}

// The first or last column of the flipped image — whichever the crop window hangs over: column 0 when the window
// starts left of the image (crop_j < pad), column W-1 when it ends right of it — is read by up to pad+1 output
// columns (times up to pad+1 rows in the corner).  16 source pixels per workgroup, 16 lanes each, striding over the
// contiguous reader columns; the corner pixel's rows are shared out over the 16 groups.  Fixed summation order.
__global__ __launch_bounds__(kBlock) void k_regularize_views_bwd_columns(const float *__restrict__ gout, float *__restrict__ gin, RegArgs a) {
  __shared__ float s_red[16];
  const int blocks_per_plane = (a.H + 15) / 16;
  const int64_t nc = blockIdx.x / blocks_per_plane;
  const int y0 = (blockIdx.x % blocks_per_plane) * 16, grp = threadIdx.x >> 4, lane = threadIdx.x & 15;
  const int n = (int)(nc / a.C);
  const int oj = a.crop_j[n] - a.pad, oi = a.crop_i[n] - a.pad;
  if (oj == 0) return;                      // centred window: every column has its single reader, already summed
  const int xs = oj < 0 ? 0 : a.W - 1;
  const float s = a.scale ? a.scale[n] : 1.0f;
  const bool dropped = a.scale && s == 0.0f;
  int jlo, jhi;
  preimage(xs, oj, a.W, jlo, jhi);
  const int nj = jhi - jlo + 1;
  const float *src = gout + nc * a.H * a.W + jlo;
  const bool f = a.flip && a.flip[n];
  const bool nrm = a.norm != 0;
  const float sd = nrm ? a.stdv[(int)(nc % a.C)] : 1.0f;
  float *dst = gin + nc * a.H * a.W + (f ? a.W - 1 - xs : xs);
  const int y = y0 + grp;
  if (y < a.H) {                            // source rows with at most one reader row
    int ilo, ihi;
    preimage(y, oi, a.H, ilo, ihi);
    if (ihi <= ilo) {
      float acc = 0.0f;
      if (!dropped && ihi == ilo && nj > 0) acc = row_sum16(src + (int64_t)ilo * a.W, nj, lane, nrm, sd);
      acc = reduce16(acc);
      if (lane == 0) dst[(int64_t)y * a.W] = a.scale ? acc * s : acc;
    }
  }
  for (int c = 0; c < 2; ++c) {             // the corner: first / last source row with several reader rows
    const int yc = c ? a.H - 1 : 0;
    if ((c && a.H == 1) || yc < y0 || yc >= y0 + 16) continue;
    int ilo, ihi;
    preimage(yc, oi, a.H, ilo, ihi);
    if (ihi <= ilo) continue;
    float acc = 0.0f;
    if (!dropped && nj > 0)
      for (int i = ilo + grp; i <= ihi; i += 16) acc += row_sum16(src + (int64_t)i * a.W, nj, lane, nrm, sd);
    acc = reduce16(acc);
    if (lane == 0) s_red[grp] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.0f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += s_red[g];
      dst[(int64_t)yc * a.W] = a.scale ? t * s : t;
    }
    __syncthreads();
  }
}

int check(hipError_t e, const char *what) {
  if (e == hipSuccess) return MVR_OK;
  char buf[200];
  snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  mvr_set_error_(buf);
  return MVR_ERR_HIP;
}

int bad(const char *msg) {
  mvr_set_error_(msg);
  return MVR_ERR_INVALID_ARG;
}

int fill_args(RegArgs &a, int32_t n_images, int32_t channels, int32_t h, int32_t w, const float *scale, const uint8_t *flip,
              const int32_t *crop_i, const int32_t *crop_j, int32_t pad) {
  if (n_images <= 0 || channels <= 0 || h <= 0 || w <= 0) return bad("mvr_regularize_views: bad image shape");
  if (pad < 0 || ((crop_i == nullptr) != (crop_j == nullptr))) return bad("mvr_regularize_views: crop_i and crop_j go together, pad >= 0");
  a = RegArgs{scale, flip, crop_i, crop_j, pad, channels, h, w, (int64_t)n_images * channels * h, 0, {0, 0, 0, 0}, {1, 1, 1, 1}};
  return MVR_OK;
}

int fill_norm(RegArgs &a, const float *mean, const float *stdv) {
  if (!mean && !stdv) return MVR_OK;
  if (!mean || !stdv) return bad("mvr_regularize_normalize_views: mean and std go together");
  if (a.C > 4) return bad("mvr_regularize_normalize_views: at most 4 channels");
  a.norm = 1;
  for (int c = 0; c < a.C; ++c) {
    if (!(stdv[c] != 0.0f)) return bad("mvr_regularize_normalize_views: std must be non-zero");
    a.mean[c] = mean[c]; a.stdv[c] = stdv[c];
  }
  return MVR_OK;
}

int launch_fwd(const float *images, float *out, const RegArgs &a, mvr_stream_t stream) {
  if (a.W % 4 == 0) {
    const int64_t total = a.rows * (a.W / 4);
    hipLaunchKernelGGL(k_regularize_views_fwd<4>, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, images, out, a);
  } else {
    const int64_t total = a.rows * a.W;
    hipLaunchKernelGGL(k_regularize_views_fwd<1>, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, images, out, a);
  }
  return check(hipGetLastError(), "k_regularize_views_fwd");
}

int launch_bwd(const float *grad_out, float *grad_images, const RegArgs &a, int32_t n_images, mvr_stream_t stream) {
  if (a.W % 4 == 0) {
    const int64_t total = a.rows * (a.W / 4);
    hipLaunchKernelGGL(k_regularize_views_bwd<4>, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, grad_out, grad_images, a);
  } else {
    const int64_t total = a.rows * a.W;
    hipLaunchKernelGGL(k_regularize_views_bwd<1>, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, grad_out, grad_images, a);
  }
  if (int rc = check(hipGetLastError(), "k_regularize_views_bwd")) return rc;
  if (!a.crop_j) return MVR_OK;   // no shift: every source pixel has exactly one reader, the kernel above is complete
  const int64_t blocks = (int64_t)n_images * a.C * ((a.H + 15) / 16);
  hipLaunchKernelGGL(k_regularize_views_bwd_columns, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, grad_out, grad_images, a);
  return check(hipGetLastError(), "k_regularize_views_bwd_columns");
}

}  // namespace

extern "C" int mvr_regularize_views_forward(const float *images, int32_t n_images, int32_t channels, int32_t h, int32_t w,
                                            const float *scale, const uint8_t *flip, const int32_t *crop_i,
                                            const int32_t *crop_j, int32_t pad, float *out, mvr_stream_t stream) {
  RegArgs a;
  if (!images || !out || images == out) return bad("mvr_regularize_views_forward: images and out must be distinct buffers");
  if (int rc = fill_args(a, n_images, channels, h, w, scale, flip, crop_i, crop_j, pad)) return rc;
  return launch_fwd(images, out, a, stream);
}

extern "C" int mvr_regularize_views_backward(const float *grad_out, int32_t n_images, int32_t channels, int32_t h, int32_t w,
                                             const float *scale, const uint8_t *flip, const int32_t *crop_i,
                                             const int32_t *crop_j, int32_t pad, float *grad_images, mvr_stream_t stream) {
  RegArgs a;
  if (!grad_out || !grad_images || grad_out == grad_images) return bad("mvr_regularize_views_backward: grad_out and grad_images must be distinct buffers");
  if (int rc = fill_args(a, n_images, channels, h, w, scale, flip, crop_i, crop_j, pad)) return rc;
  return launch_bwd(grad_out, grad_images, a, n_images, stream);
}

// The same pass with the trainer's per-channel `Normalize` fused behind it: out = (regularised - mean[c]) / std[c];
// mean / std: `channels` HOST floats each (both NULL: no normalisation).  Any of scale / flip / crop may be NULL, so this
// is also the stand-alone Normalize.
extern "C" int mvr_regularize_normalize_views_forward(const float *images, int32_t n_images, int32_t channels, int32_t h, int32_t w,
                                                      const float *scale, const uint8_t *flip, const int32_t *crop_i,
                                                      const int32_t *crop_j, int32_t pad, const float *mean, const float *stdv,
                                                      float *out, mvr_stream_t stream) {
  RegArgs a;
  if (!images || !out || images == out) return bad("mvr_regularize_normalize_views_forward: images and out must be distinct buffers");
  if (int rc = fill_args(a, n_images, channels, h, w, scale, flip, crop_i, crop_j, pad)) return rc;
  if (int rc = fill_norm(a, mean, stdv)) return rc;
  return launch_fwd(images, out, a, stream);
}

extern "C" int mvr_regularize_normalize_views_backward(const float *grad_out, int32_t n_images, int32_t channels, int32_t h, int32_t w,
                                                       const float *scale, const uint8_t *flip, const int32_t *crop_i,
                                                       const int32_t *crop_j, int32_t pad, const float *mean, const float *stdv,
                                                       float *grad_images, mvr_stream_t stream) {
  RegArgs a;
  if (!grad_out || !grad_images || grad_out == grad_images) return bad("mvr_regularize_normalize_views_backward: grad_out and grad_images must be distinct buffers");
  if (int rc = fill_args(a, n_images, channels, h, w, scale, flip, crop_i, crop_j, pad)) return rc;
  if (int rc = fill_norm(a, mean, stdv)) return rc;
  return launch_bwd(grad_out, grad_images, a, n_images, stream);
}
