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

