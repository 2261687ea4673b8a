// The sampling tail of the imputation updates (vgiwae/shared/vae_imputation.py:38-52 pseudo-Gibbs; :526-561, :657-666
// LAIR): draw x_m ~ p(. | z), clamp, non-finite fallback, blend with the observed values.  Shared by the row kernels and
// the one-launch LAIR update.
#pragma once
#include "dmv_common.h"

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
  const uint64_t* offset_base;  // nullable device counter (a multiple of 4) added to offset
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

__device__ __forceinline__ float tail_element(const TailArgs& a, const float* dec, int d, float noise, float xo, bool obs) {
  float xm;
  if (a.gen_dist <= DMV_GEN_NORMAL_EPS) {
    const float s = dmv_softplus(dec[a.D + d]) + (a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f);
    xm = dec[d] + noise * s;
  } else {
    const float b = noise < dmv_sigmoid(dec[d]) ? 1.f : 0.f;
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
  return out;
}
