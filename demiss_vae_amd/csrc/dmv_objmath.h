// Formulas shared by the objective kernels: encoder-output addressing, log densities, sub-wave reductions.
#pragma once
#include "dmv_common.h"

#define LOG_SQRT_2PI 0.9189385332046727f

struct QAddr {
  int64_t sb, sg, raw_off, logit_off;
  float eps;
};

__device__ __forceinline__ float q_mu(const float* enc, const QAddr& q, int64_t b, int g, int j) {
  return enc[b * q.sb + g * q.sg + j];
}
__device__ __forceinline__ float q_sigma(const float* enc, const QAddr& q, int64_t b, int g, int j) {
  return dmv_softplus(enc[b * q.sb + g * q.sg + q.raw_off + j]) + q.eps;
}
__device__ __forceinline__ float log_normal(float x, float mu, float sigma) {
  const float d = x - mu;
  return -(d * d) / (2.f * sigma * sigma) - dmv_log(sigma) - LOG_SQRT_2PI;
}
// -BCEWithLogits(l, v) = -((1 - v) l - logsigmoid(l)),  logsigmoid(l) = min(l, 0) - log1p(exp(-|l|))
__device__ __forceinline__ float log_bern(float v, float l) {
  const float ls = fminf(l, 0.f) - dmv_log1p_pos(dmv_exp(-fabsf(l)));
  return -((1.f - v) * l - ls);
}

// sub-wave sum over R consecutive lanes (R power of two <= 64)
template <int R>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = R / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <int R>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = R / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// Sub-wave sum over R consecutive lanes on the data-parallel-primitive path (no LDS round trip as in __shfl_xor):
// quad permutes for the offsets 1 and 2, then row_half_mirror (lane i <-> 7 - i within 8) and row_mirror
// (i <-> 15 - i within 16): once every lane of a quad holds the quad's sum, a mirror pairs each lane with a lane of
// the other half, so all lanes end with the total.  Offsets 16 and 32 fall back to __shfl_xor.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int R>
__device__ __forceinline__ float gsum(float v) {
  if (R >= 2) v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  if (R >= 4) v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  if (R >= 8) v += dpp_move<0x141>(v);   // row_half_mirror
  if (R >= 16) v += dpp_move<0x140>(v);  // row_mirror
  if (R >= 32) v += __shfl_xor(v, 16);
  if (R >= 64) v += __shfl_xor(v, 32);
  return v;
}
// log N(x; mu, sigma) with the division replaced by a reciprocal (1 ulp): -((x - mu) / sigma)^2 / 2 - log sigma - c
__device__ __forceinline__ float log_normal_r(float x, float mu, float sigma) {
  const float d = (x - mu) * __builtin_amdgcn_rcpf(sigma);
  return -0.5f * d * d - dmv_log(sigma) - LOG_SQRT_2PI;
}

// One entry of the LDS table of q's parameters: log N(z; mu, sigma) = -(z - mu)^2 * inv2s2 - lnorm.  The softplus, the
// logarithm and the reciprocal are paid once per (datapoint, component, dimension) of the workgroup instead of once
// per sample row that evaluates the component.
struct __attribute__((aligned(16))) QEntry { float mu, inv2s2, lnorm, pad; };

__device__ __forceinline__ float q_logp(const QEntry& e, float z) {
  const float d = z - e.mu;
  return -(d * d) * e.inv2s2 - e.lnorm;
}


// Sum of an array over the threads of a workgroup, thread t taking elements t, t + nthr, ... (16-byte loads, four in
// flight per thread): fixed summation order for a given workgroup size, so results are reproducible bit for bit.
__device__ __forceinline__ float sum_strided(const float* __restrict__ v, int64_t n) {
  float s = 0.f;
  if (!v) return s;
  const int64_t tid = threadIdx.x, nt = blockDim.x;
  if ((((uintptr_t)v) & 15) == 0) {
    const f32x4* __restrict__ v4 = (const f32x4*)v;
    const int64_t n4 = n >> 2;
    int64_t i = tid;
    for (; i + 3 * nt < n4; i += 4 * nt) {
      const f32x4 a = v4[i], b = v4[i + nt], c = v4[i + 2 * nt], d = v4[i + 3 * nt];
      s += ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3])) + ((c[0] + c[1]) + (c[2] + c[3])) +
           ((d[0] + d[1]) + (d[2] + d[3]));
    }
    for (; i < n4; i += nt) { const f32x4 a = v4[i]; s += (a[0] + a[1]) + (a[2] + a[3]); }
    for (int64_t k = (n4 << 2) + tid; k < n; k += nt) s += v[k];
  } else {
    for (int64_t i = tid; i < n; i += nt) s += v[i];
  }
  return s;
}

// means of one or two arrays by the calling workgroup (every thread calls; sh: 32 floats of LDS)
__device__ __forceinline__ void block_mean2(float* sh, const float* __restrict__ v1, const float* __restrict__ v2, int64_t n,
                                            float scale, float* __restrict__ o1, float* __restrict__ o2,
                                            float* __restrict__ o1_neg) {
  const float s1 = gsum<64>(sum_strided(v1, n)), s2 = gsum<64>(sum_strided(v2, n));
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = s1; sh[16 + (threadIdx.x >> 6)] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t1 = 0.f, t2 = 0.f;
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) { t1 += sh[w]; t2 += sh[16 + w]; }
    if (o1) o1[0] = t1 * scale;
    if (o1_neg) o1_neg[0] = -(t1 * scale);
    if (o2) o2[0] = t2 * scale;
  }
}

// The batch mean of per-datapoint bounds inside the launch that produced them: every workgroup makes its results
// visible and takes a ticket; the one that draws the last sums all bounds in a fixed order and leaves the counter at
// zero for the next launch.  Every thread of every workgroup must call it (sh: 32 floats of LDS).
__device__ __forceinline__ void ticket_mean(unsigned* ticket, float* sh, unsigned* last, const float* bound, int64_t n,
                                            float* mean, float* neg_mean) {
  if (!ticket) return;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) *last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!*last) return;
  __threadfence();
  block_mean2(sh, bound, nullptr, n, 1.f / (float)n, mean, nullptr, neg_mean);
  if (threadIdx.x == 0) *ticket = 0u;
}
