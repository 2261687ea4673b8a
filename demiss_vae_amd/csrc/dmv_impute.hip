// DeMissVAE imputation update: device-resident imputation buffer (gather / scatter by imputation
// index), the pseudo-Gibbs / LAIR sampling tail (draw x_m ~ p(.|z), clamp, non-finite fallback, blend
// with the observed values, write back) and the per-datapoint categorical resampling of LAIR, with
// on-device Philox noise or caller-provided noise.  All HBM-bound row kernels.
#include "dmv_common.h"
#include "dmv_rng.h"

// ------------------------------------------------------------------------------------------
// buffer rows (vgiwae/utils/imputation_distribution.py:145-157, 180-187)
// ------------------------------------------------------------------------------------------
__global__ void k_rows_gather(const float* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, int D,
                              float* __restrict__ dst) {
  const int64_t total = n * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    dst[i] = src[idx[r] * D + (i - r * D)];
  }
}

__global__ void k_rows_scatter(float* __restrict__ dst, const int64_t* __restrict__ idx, int64_t n, int D,
                               const float* __restrict__ src) {
  const int64_t total = n * D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    dst[idx[r] * D + (i - r * D)] = src[i];
  }
}

// ------------------------------------------------------------------------------------------
// sampling tail (vgiwae/shared/vae_imputation.py:38-52 pseudo-Gibbs; :526-561, :657-666 LAIR)
// ------------------------------------------------------------------------------------------
struct TailArgs {
  int64_t N;   // output rows (b, k)
  int D;
  int gen_dist;
  const float* dec;          // decoder outputs, row r = row_map ? row_map[n] : n
  int64_t ld_dec;
  const int64_t* row_map;    // nullable
  const float* X_old;        // [N][D]
  const uint8_t* M;          // [N][D]
  const float* noise;        // nullable [N][D]: standard normal (Gaussian decoder) or U[0,1) (Bernoulli)
  uint64_t seed, offset;
  const float* clip_min;     // nullable [D]
  const float* clip_max;     // nullable [D]
  int clip_after_blend;      // LAIR clips the blended row (observed entries included)
  float* X_out;              // [N][D]
  float* buf;                // nullable: imputation buffer, rows of D floats
  const int64_t* idx_imp;    // [N] rows of buf to overwrite
};

// torch.max / torch.min propagate NaN (fmaxf / fminf would drop it)
__device__ __forceinline__ float clamp_like_torch(float v, const float* lo, const float* hi, int d) {
  if (v != v) return v;
  if (lo) v = fmaxf(v, lo[d]);
  if (hi) v = fminf(v, hi[d]);
  return v;
}

__global__ void k_impute_tail(TailArgs a) {
  const int64_t total = a.N * a.D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / a.D;
    const int d = (int)(i - n * a.D);
    const int64_t r = a.row_map ? a.row_map[n] : n;
    const float* dec = a.dec + r * a.ld_dec;
    const float xo = a.X_old[i];
    const bool obs = a.M[i] != 0;
    float xm;
    if (a.gen_dist <= DMV_GEN_NORMAL_EPS) {
      const float e = a.noise ? a.noise[i] : dmv_philox_normal(a.seed, a.offset + (uint64_t)i);
      const float s = dmv_softplus(dec[a.D + d]) + (a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f);
      xm = dec[d] + e * s;
    } else {
      const float u = a.noise ? a.noise[i] : dmv_philox_uniform(a.seed, a.offset + (uint64_t)i);
      const float b = u < dmv_sigmoid(dec[d]) ? 1.f : 0.f;
      xm = a.gen_dist == DMV_GEN_BERN ? b : 2.f * b - 1.f;
    }
    float out;
    if (!a.clip_after_blend) {
      xm = clamp_like_torch(xm, a.clip_min, a.clip_max, d);
      if (!isfinite(xm)) xm = xo;
      out = obs ? xo : xm;
    } else {
      out = obs ? xo : xm;
      if (a.clip_min || a.clip_max) {
        out = clamp_like_torch(out, a.clip_min, a.clip_max, d);
        if (!isfinite(out)) out = xo;
      }
    }
    a.X_out[i] = out;
    if (a.buf) a.buf[a.idx_imp[n] * a.D + d] = out;
  }
}

// ------------------------------------------------------------------------------------------
// categorical draws per datapoint (vgiwae/shared/vae_imputation.py:181-186, :679-680):
// idx[b][k] = smallest p with cumsum(softmax(logits[b]))[p] > u[b][k]; one wavefront per datapoint
// ------------------------------------------------------------------------------------------
struct DrawArgs {
  int B, P, K;
  const float* logits;  // [B][P]
  const float* u;       // nullable [B][K] in [0,1)
  uint64_t seed, offset;
  int64_t* idx;         // [B][K]
};

__global__ void k_categorical_draw(DrawArgs a) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const float* lw = a.logits + (int64_t)b * a.P;
  float mx = -INFINITY;
  for (int p = lane; p < a.P; p += 64) mx = fmaxf(mx, lw[p]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float se = 0.f;
  for (int p = lane; p < a.P; p += 64) se += expf(lw[p] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
  for (int k = lane; k < a.K; k += 64) {
    const float u = a.u ? a.u[(int64_t)b * a.K + k] : dmv_philox_uniform(a.seed, a.offset + (uint64_t)b * a.K + k);
    const float target = u * se;
    float c = 0.f;
    int pick = a.P - 1;
    for (int p = 0; p < a.P; ++p) {
      c += expf(lw[p] - mx);
      if (c > target) { pick = p; break; }
    }
    a.idx[(int64_t)b * a.K + k] = pick;
  }
}

// Bernoulli(1 - p) keep masks for the residual blocks' dropout (vgiwae/shared/neural_nets.py:65,70)
__global__ void k_dropout_masks(uint8_t* __restrict__ keep, int64_t n, float p, uint64_t seed, uint64_t offset) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    keep[i] = dmv_philox_uniform(seed, offset + (uint64_t)i) >= p ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
static inline unsigned grid_for(int64_t work, int block, unsigned cap = 8192) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}

extern "C" int dmv_rows_gather(const float* src, const int64_t* idx, int64_t n, int32_t D, float* dst, void* stream) {
  if (!src || !idx || !dst || n < 0 || D < 1) return DMV_E_BADARG;
  if (n == 0) return DMV_OK;
  hipLaunchKernelGGL(k_rows_gather, dim3(grid_for(n * D, 256)), dim3(256), 0, (hipStream_t)stream, src, idx, n, (int)D, dst);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_rows_scatter(float* dst, const int64_t* idx, int64_t n, int32_t D, const float* src, void* stream) {
  if (!src || !idx || !dst || n < 0 || D < 1) return DMV_E_BADARG;
  if (n == 0) return DMV_OK;
  hipLaunchKernelGGL(k_rows_scatter, dim3(grid_for(n * D, 256)), dim3(256), 0, (hipStream_t)stream, dst, idx, n, (int)D, src);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_impute_tail(int64_t N, int32_t D, int32_t gen_dist, const float* dec, int64_t ld_dec,
                               const int64_t* row_map, const float* X_old, const uint8_t* M, const float* noise,
                               uint64_t seed, uint64_t offset, const float* clip_min, const float* clip_max,
                               int32_t clip_after_blend, float* X_out, float* buf, const int64_t* idx_imp,
                               void* stream) {
  if (N < 0 || D < 1 || !dec || !X_old || !M || !X_out) return DMV_E_BADARG;
  if (gen_dist < DMV_GEN_NORMAL || gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
  const int P = gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * D : D;
  if (ld_dec < P || (buf && !idx_imp)) return DMV_E_BADARG;
  if (N == 0) return DMV_OK;
  TailArgs a;
  a.N = N; a.D = D; a.gen_dist = gen_dist; a.dec = dec; a.ld_dec = ld_dec; a.row_map = row_map;
  a.X_old = X_old; a.M = M; a.noise = noise; a.seed = seed; a.offset = offset;
  a.clip_min = clip_min; a.clip_max = clip_max; a.clip_after_blend = clip_after_blend;
  a.X_out = X_out; a.buf = buf; a.idx_imp = idx_imp;
  hipLaunchKernelGGL(k_impute_tail, dim3(grid_for(N * D, 256)), dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_categorical_draw(int32_t B, int32_t P, int32_t K, const float* logits, const float* u, uint64_t seed,
                                    uint64_t offset, int64_t* idx, void* stream) {
  if (B < 1 || P < 1 || K < 1 || !logits || !idx) return DMV_E_BADARG;
  DrawArgs a;
  a.B = B; a.P = P; a.K = K; a.logits = logits; a.u = u; a.seed = seed; a.offset = offset; a.idx = idx;
  hipLaunchKernelGGL(k_categorical_draw, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_dropout_masks(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
  if (!keep || n < 0 || !(p >= 0.f && p < 1.f)) return DMV_E_BADARG;
  if (n == 0) return DMV_OK;
  hipLaunchKernelGGL(k_dropout_masks, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, keep, n, p, seed, offset);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}
