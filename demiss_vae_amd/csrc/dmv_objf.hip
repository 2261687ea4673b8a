// The masked objective in one launch: per-sample terms, bound (logsumexp over the importance / component axes),
// softmax weights for the backward and the mixture-logit gradient, for a block of datapoints at a time
// (vgiwae/shared/iwae.py:9-132, vgiwae/models/iwae.py:70-123, models/multiple_iwae.py:124-199,
// models/mvbvae.py:83-137).  HBM-bound.
//
// Work decomposition.  Sample rows are n = s*B*G + b*G + g (s = z*I + i).  A workgroup owns DPB consecutive
// datapoints, i.e. Q = DPB*G consecutive (b, g) pairs; for every s their rows are consecutive in memory, so a wave
// reads consecutive rows.  A group of R lanes walks one (b, g) pair through all S samples: the dimensions of a row are
// spread over the R lanes (consecutive lanes read consecutive floats) and summed with DPP reductions; lane 0 of the
// group leaves the row's log-weight in LDS.  After a barrier the logsumexp over (i, g) of each datapoint runs out of
// LDS on a sub-wave group of T lanes, the normalised weights replace the log-weights in LDS, and a last sweep
// writes them to HBM with consecutive threads on consecutive rows.  Nothing but the decoder outputs, the latent
// samples and the [B]-sized inputs is read, and nothing is read twice; no 64-bit division anywhere.
#include "dmv_common.h"
#include "dmv_objmath.h"

namespace {

struct FwdArgs {
  int B, G, S, nz, I, D, Zd;
  int gen_dist, xpc, q_kind, mix_C;
  int kind;        // DMV_BOUND_LSE / _WEIGHTED / _PER_COMPONENT, or -1: terms only
  int q_term;      // 3: own, 4: mixture (which log q enters the log-weight)
  int use_logpi, dreg;
  int DPB, T;      // datapoints per workgroup; lanes per datapoint in the logsumexp stage
  int vec4;
  int nC;          // components per datapoint in the LDS table of q's parameters (0: no table, q evaluated from enc)
  QAddr q;
  const float* __restrict__ dec; int64_t ld_dec;
  const float* __restrict__ X; const uint8_t* __restrict__ M;
  const float* __restrict__ Z; int64_t ldz;
  const float* __restrict__ enc;
  float* __restrict__ terms;    // nullable [5][N]
  float* __restrict__ comp_lp;  // nullable [N][mix_C]
  float* __restrict__ log_w;    // nullable [N]
  float* __restrict__ bound;    // [B] (LSE / WEIGHTED)
  float* __restrict__ wgt;      // nullable [N]
  float* __restrict__ zscale;   // nullable [N]
  float* __restrict__ dlogit;   // nullable [B][G]
  float cA[5], cB[5];           // PER_COMPONENT: objA = mean_z sum_t cA[t] terms[t], objB likewise (nullable)
  float* __restrict__ objA; float* __restrict__ objB;
  // means in the same launch: the workgroup that finishes last (a ticket counter in device memory, zero between launches)
  // sums the per-datapoint results.  mean1 / mean2: means of bound (or objA) / of objB; neg1 = -mean1.
  unsigned* ticket;
  float* mean1; float* mean2; float* neg1;
};

// Batch means in the launch that forms the per-datapoint results: every workgroup takes a ticket when its results are
// out, and the one that draws the last ticket sums what all of them wrote (then leaves the counter at zero for the next
// launch).  Making ordinary stores visible to that workgroup would take a device-scope release, i.e. a write-back of every
// dirty L2 line the kernels before this one have left on this multi-die part (10 us in a small batch's step).  The few
// results that are summed therefore go out as device-scope atomic exchanges, which are performed at the level all compute
// dies share; their old values come back into `parked`, and finish_means uses `parked` before the barrier in front of the
// ticket, so every exchange of the workgroup has been performed by then.
__device__ __forceinline__ void put_result(const FwdArgs& a, float* dst, float v, float& parked) {
  if (a.ticket) parked += atomicExch(dst, v);
  else *dst = v;
}

// Every thread of the workgroup must call it.
__device__ __forceinline__ void finish_means(const FwdArgs& a, float parked) {
  if (!a.ticket) return;
  __shared__ float sh[32];
  __shared__ unsigned last;
  sh[threadIdx.x & 31] = parked;            // (waits for the exchanges; the values are never read)
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (a.kind == DMV_BOUND_PER_COMPONENT) {
    const int64_t pairs = (int64_t)a.B * a.G;
    block_mean2(sh, a.objA, a.objB, pairs, 1.f / (float)pairs, a.mean1, a.mean2, nullptr);
  } else {
    block_mean2(sh, a.bound, nullptr, (int64_t)a.B, 1.f / (float)a.B, a.mean1, nullptr, a.neg1);
  }
  if (threadIdx.x == 0) *a.ticket = 0u;
}

// the five terms of one sample row, summed over the R lanes of the group (every lane returns the totals).
// qt: the workgroup's table [DPB][nC][Zd] (null: evaluate q from the encoder output), bl: datapoint within the workgroup
template <int R>
__device__ __forceinline__ void row_terms(const FwdArgs& a, int r, int64_t n, int64_t b, int bl, int g, const QEntry* qt,
                                          const float* __restrict__ X, const uint8_t* __restrict__ M, float& gm, float& gf,
                                          float& pr, float& qo, float& qm) {
  const float* __restrict__ dec = a.dec + n * a.ld_dec;
  gm = 0.f; gf = 0.f;
  if (a.vec4) {
    // wide rows (Bernoulli images): 16 bytes of logits / data and 4 mask bytes per lane and iteration
    for (int q = r; q < (a.D >> 2); q += R) {
      const f32x4 l4 = *(const f32x4*)(dec + 4 * q);
      const f32x4 x4 = *(const f32x4*)(X + 4 * q);
      const uint32_t m4 = *(const uint32_t*)(M + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lp = log_bern(a.gen_dist == DMV_GEN_BERN ? x4[e] : (x4[e] + 1.f) * 0.5f, l4[e]);
        gf += lp;
        gm += lp * (((m4 >> (8 * e)) & 0xffu) ? 1.f : 0.f);
      }
    }
  } else {
    for (int d = r; d < a.D; d += R) {
      float lp;
      const float x = X[d];
      if (a.gen_dist <= DMV_GEN_NORMAL_EPS) {
        const float s = dmv_softplus(dec[a.D + d]) + (a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f);
        lp = log_normal_r(x, dec[d], s);
      } else if (a.gen_dist == DMV_GEN_BERN) {
        lp = log_bern(x, dec[d]);
      } else {
        lp = log_bern((x + 1.f) * 0.5f, dec[d]);
      }
      gf += lp;
      gm += lp * (M[d] ? 1.f : 0.f);     // the product the reference forms (shared/iwae.py:20): a non-finite lp at a
                                         // missing entry makes the row's log-weight NaN there and here alike
    }
  }
  const float* __restrict__ z = a.Z + n * a.ldz;
  pr = 0.f; qo = 0.f;
  const bool own = a.q_kind != DMV_Q_STRATIFIED || a.G > 1;   // rows sampled from the mixture itself have no own component
  const QEntry* qb = qt ? qt + (size_t)bl * a.nC * a.Zd : nullptr;
  for (int j = r; j < a.Zd; j += R) {
    const float zj = z[j];
    pr += -(zj * zj) * 0.5f - LOG_SQRT_2PI;
    if (own) qo += qb ? q_logp(qb[g * a.Zd + j], zj) : log_normal_r(zj, q_mu(a.enc, a.q, b, g, j), q_sigma(a.enc, a.q, b, g, j));
  }
  gm = gsum<R>(gm); gf = gsum<R>(gf); pr = gsum<R>(pr); qo = gsum<R>(qo);
  qm = qo;
  if (a.q_kind != DMV_Q_OWN) {
    float mx = -INFINITY, se = 0.f, lse_logit = 0.f;
    const float* __restrict__ lg = a.enc + b * a.q.sb + a.q.logit_off;
    if (a.q_kind == DMV_Q_STRATIFIED) {
      float lm = -INFINITY;
      for (int c = 0; c < a.mix_C; ++c) lm = fmaxf(lm, lg[c]);
      float ss = 0.f;
      for (int c = 0; c < a.mix_C; ++c) ss += dmv_exp(lg[c] - lm);
      lse_logit = lm + dmv_log(ss);
    }
    for (int c = 0; c < a.mix_C; ++c) {
      float lp = 0.f;
      if (qb) {
        for (int j = r; j < a.Zd; j += R) lp += q_logp(qb[c * a.Zd + j], z[j]);
      } else {
        for (int j = r; j < a.Zd; j += R) lp += log_normal_r(z[j], q_mu(a.enc, a.q, b, c, j), q_sigma(a.enc, a.q, b, c, j));
      }
      lp = gsum<R>(lp);
      if (a.comp_lp && r == 0) a.comp_lp[n * a.mix_C + c] = lp;
      const float t = lp + (a.q_kind == DMV_Q_STRATIFIED ? lg[c] - lse_logit : 0.f);
      const float nm = fmaxf(mx, t);
      se = se * dmv_exp(mx - nm) + dmv_exp(t - nm);
      mx = nm;
    }
    qm = mx + dmv_log(se) - (a.q_kind == DMV_Q_IMPUTATION ? dmv_log((float)a.mix_C) : 0.f);
  }
}

__device__ __forceinline__ void bound_phases(const FwdArgs& a, float* lw, float* aux, int tid, int nthr, int b0, int nb, int Qn,
                                             int Q, int64_t BG, int64_t row0, float& parked);

__device__ __forceinline__ float tsum(float v, int T) {
  for (int o = T >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float tmax(float v, int T) {
  for (int o = T >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

template <int R>
__global__ void __launch_bounds__(256) k_obj_fwd(FwdArgs a) {
  extern __shared__ float sm[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int NG = nthr / R;
  const int r = tid % R, grp = tid / R;
  const int b0 = blockIdx.x * a.DPB;
  const int nb = min(a.DPB, a.B - b0);
  const int Qn = nb * a.G, Q = a.DPB * a.G;
  const int64_t BG = (int64_t)a.B * a.G;
  const int64_t N = BG * a.S;
  const int64_t row0 = (int64_t)b0 * a.G;
  const bool per_comp = a.kind == DMV_BOUND_PER_COMPONENT;
  // LDS: the q table first (16-byte entries), then the log-weights and the mixture bookkeeping
  QEntry* qt = a.nC ? (QEntry*)sm : nullptr;
  float* lw = sm + (size_t)4 * a.DPB * a.nC * a.Zd;   // [S][Q] log-weights, later the weights (PER_COMPONENT: objA summands)
  float* aux = lw + (a.kind < 0 ? 0 : (size_t)a.S * Q);   // PER_COMPONENT: [S][Q] objB summands; else [3][Q]: log pi, L_g, dlogit
  if (qt) {
    const int per_dp = a.nC * a.Zd;
    for (int e = tid; e < nb * per_dp; e += nthr) {
      const int bl = e / per_dp, rem = e - bl * per_dp;
      const int c = rem / a.Zd, j = rem - c * a.Zd;
      const float sg = q_sigma(a.enc, a.q, b0 + bl, c, j);
      const float rs = __builtin_amdgcn_rcpf(sg);
      qt[e] = QEntry{q_mu(a.enc, a.q, b0 + bl, c, j), 0.5f * rs * rs, dmv_log(sg) + LOG_SQRT_2PI, 0.f};
    }
    __syncthreads();
  }

  // ---- phase 1: terms of every row of the block's datapoints
  for (int q = grp; q < Qn; q += NG) {
    const int bl = q / a.G, g = q - bl * a.G;
    const int64_t b = b0 + bl;
    const int64_t xrow = a.xpc ? row0 + q : b;
    const float* __restrict__ X = a.X + xrow * a.D;
    const uint8_t* __restrict__ M = a.M + xrow * a.D;
    for (int s = blockIdx.y; s < a.S; s += gridDim.y) {     // gridDim.y > 1 only when no bound is formed (terms only)
      const int64_t n = (int64_t)s * BG + row0 + q;
      float gm, gf, pr, qo, qm;
      row_terms<R>(a, r, n, b, bl, g, qt, X, M, gm, gf, pr, qo, qm);
      if (r == 0) {
        if (a.terms) {
          a.terms[n] = gm; a.terms[N + n] = gf; a.terms[2 * N + n] = pr; a.terms[3 * N + n] = qo; a.terms[4 * N + n] = qm;
        }
        if (per_comp) {
          const float t5[5] = {gm, gf, pr, qo, qm};
          float vA = 0.f, vB = 0.f;
#pragma unroll
          for (int t = 0; t < 5; ++t) {
            if (a.cA[t] != 0.f) vA += a.cA[t] * t5[t];
            if (a.cB[t] != 0.f) vB += a.cB[t] * t5[t];
          }
          lw[s * Q + q] = vA;
          aux[s * Q + q] = vB;
        } else if (a.kind >= 0) {
          const float w = gm + pr - (a.q_term == 3 ? qo : qm);
          lw[s * Q + q] = w;
          if (a.log_w) a.log_w[n] = w;
        }
      }
    }
  }
  if (a.kind < 0) return;
  float parked = 0.f;
  bound_phases(a, lw, aux, tid, nthr, b0, nb, Qn, Q, BG, row0, parked);
  finish_means(a, parked);
}

// Everything after the log-weights of a workgroup's rows are in LDS: bounds by logsumexp over (i, g), weights of the
// backward, mixture-logit gradients, per-component objectives.  lw [S][Q], aux as laid out by the kernels.
__device__ __forceinline__ void bound_phases(const FwdArgs& a, float* lw, float* aux, int tid, int nthr, int b0, int nb, int Qn,
                                             int Q, int64_t BG, int64_t row0, float& parked) {
  const bool per_comp = a.kind == DMV_BOUND_PER_COMPONENT;
  // ---- normalised log mixture weights of the block's datapoints
  const bool need_pi = !per_comp && (a.use_logpi || a.kind == DMV_BOUND_WEIGHTED);
  float* lpi = aux;
  if (need_pi) {
    for (int bl = tid; bl < nb; bl += nthr) {
      const float* __restrict__ lg = a.enc + (int64_t)(b0 + bl) * a.q.sb + a.q.logit_off;
      float lm = -INFINITY;
      for (int c = 0; c < a.G; ++c) lm = fmaxf(lm, lg[c]);
      float ss = 0.f;
      for (int c = 0; c < a.G; ++c) ss += expf(lg[c] - lm);
      const float lse = lm + logf(ss);
      for (int c = 0; c < a.G; ++c) lpi[bl * a.G + c] = lg[c] - lse;
    }
  }
  __syncthreads();

  const float inv_nz = 1.f / (float)a.nz;
  const float logI = logf((float)a.I);
  if (per_comp) {
    // objectives per (b, g): mean over the outer samples (I == 1)
    for (int q = tid; q < Qn; q += nthr) {
      float sA = 0.f, sB = 0.f;
      for (int s = 0; s < a.S; ++s) { sA += lw[s * Q + q]; sB += aux[s * Q + q]; }
      if (a.objA) put_result(a, a.objA + row0 + q, sA / (float)a.nz, parked);
      if (a.objB) put_result(a, a.objB + row0 + q, sB / (float)a.nz, parked);
    }
    return;
  }

  const bool need_w = a.wgt || a.zscale || a.dlogit;
  if (a.kind == DMV_BOUND_LSE) {
    // logsumexp over (i, g) per datapoint and outer sample, T lanes per datapoint
    const int T = a.T, lt = tid % T, sub = tid / T, nsub = nthr / T;
    const int IG = a.I * a.G;
    for (int bl = sub; bl < nb; bl += nsub) {
      float bacc = 0.f;
      for (int z = 0; z < a.nz; ++z) {
        float mx = -INFINITY;
        for (int t = lt; t < IG; t += T) {
          const int i = t / a.G, g = t - i * a.G;
          mx = fmaxf(mx, lw[(z * a.I + i) * Q + bl * a.G + g] + (a.use_logpi ? lpi[bl * a.G + g] : 0.f));
        }
        mx = tmax(mx, T);
        float se = 0.f;
        for (int t = lt; t < IG; t += T) {
          const int i = t / a.G, g = t - i * a.G;
          se += expf(lw[(z * a.I + i) * Q + bl * a.G + g] + (a.use_logpi ? lpi[bl * a.G + g] : 0.f) - mx);
        }
        se = tsum(se, T);
        const float lse = mx + logf(se);
        bacc += lse - logI;
        if (need_w) {
          for (int t = lt; t < IG; t += T) {
            const int i = t / a.G, g = t - i * a.G;
            float* slot = &lw[(z * a.I + i) * Q + bl * a.G + g];
            *slot = expf(*slot + (a.use_logpi ? lpi[bl * a.G + g] : 0.f) - lse);
          }
        }
      }
      if (lt == 0) put_result(a, a.bound + b0 + bl, bacc * inv_nz, parked);
    }
  } else {
    // sum_g pi_g (logsumexp_i log_w - log I): one thread per datapoint (G and I are small here)
    float* Lg = aux + Q;
    float* dl = aux + 2 * Q;
    for (int bl = tid; bl < nb; bl += nthr) {
      float bacc = 0.f;
      for (int g = 0; g < a.G; ++g) dl[bl * a.G + g] = 0.f;
      for (int z = 0; z < a.nz; ++z) {
        float tot = 0.f;
        for (int g = 0; g < a.G; ++g) {
          const float pig = expf(lpi[bl * a.G + g]);
          float mx = -INFINITY;
          for (int i = 0; i < a.I; ++i) mx = fmaxf(mx, lw[(z * a.I + i) * Q + bl * a.G + g]);
          float se = 0.f;
          for (int i = 0; i < a.I; ++i) se += expf(lw[(z * a.I + i) * Q + bl * a.G + g] - mx);
          const float lse = mx + logf(se);
          tot += pig * (lse - logI);
          Lg[bl * a.G + g] = lse - logI;
          if (need_w)
            for (int i = 0; i < a.I; ++i) {
              float* slot = &lw[(z * a.I + i) * Q + bl * a.G + g];
              *slot = pig * expf(*slot - lse);
            }
        }
        bacc += tot;
        for (int g = 0; g < a.G; ++g) dl[bl * a.G + g] += expf(lpi[bl * a.G + g]) * (Lg[bl * a.G + g] - tot) * inv_nz;
      }
      put_result(a, a.bound + b0 + bl, bacc * inv_nz, parked);
    }
  }
  if (!need_w) return;
  __syncthreads();

  // ---- gradient of sum_b bound_b with respect to the mixture logits through the pi factors
  if (a.dlogit) {
    for (int q = tid; q < Qn; q += nthr) {
      float v = 0.f;
      if (a.kind == DMV_BOUND_WEIGHTED) {
        v = aux[2 * Q + q];
      } else if (a.use_logpi) {
        // d/d logit_c of sum_n p_n log pi_{g(n)} = (sum_{n: g(n) = c} p_n) - pi_c, per outer sample
        float acc = 0.f;
        for (int s = 0; s < a.S; ++s) acc += lw[s * Q + q];
        v = (acc - (float)a.nz * expf(lpi[q])) * inv_nz;
      }
      a.dlogit[row0 + q] = v;
    }
  }
  // ---- weights of the backward, consecutive threads on consecutive rows
  for (int s = 0; s < a.S; ++s)
    for (int q = tid; q < Qn; q += nthr) {
      const int64_t n = (int64_t)s * BG + row0 + q;
      const float p = lw[s * Q + q];
      if (a.wgt) a.wgt[n] = p * inv_nz;
      if (a.zscale) a.zscale[n] = (a.dreg && a.kind == DMV_BOUND_LSE) ? p : 1.f;
    }
}

// Short, 16-byte aligned rows of a Gaussian decoder (D and Zd multiples of 4: the UCI-gas shapes of the headline
// configuration): one THREAD per sample row, D4 = D / 4 and Z4 = Zd / 4 known at compile time.  With a group of lanes per
// row most instructions of a wave are address arithmetic, reductions and loop control around a handful of useful
// operations per lane, and the kernel is bound by instruction issue; one lane per row spends its instructions on the
// row, and because the loops unroll completely every load of the row (2 D4 + Z4 of 16 bytes, plus data and mask) is
// issued before the first is used.  The logsumexp stage is the same as in the general kernel.
template <int D4, int Z4>
__global__ void __launch_bounds__(256) k_obj_fwd_rows4(FwdArgs a) {
  extern __shared__ float sm[];
  constexpr int D = 4 * D4, Zd = 4 * Z4;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int b0 = blockIdx.x * a.DPB;
  const int nb = min(a.DPB, a.B - b0);
  const int Qn = nb * a.G, Q = a.DPB * a.G;
  const int64_t BG = (int64_t)a.B * a.G;
  const int64_t N = BG * a.S;
  const int64_t row0 = (int64_t)b0 * a.G;
  const bool per_comp = a.kind == DMV_BOUND_PER_COMPONENT;
  // a datapoint's table entries are followed by one unused entry: 64 lanes on 64 datapoints then read 16-byte entries at
  // a stride that is not a multiple of the bank width
  const int qstride = a.nC * Zd + 1;
  QEntry* qt = (QEntry*)sm;
  float* lw = sm + (size_t)4 * a.DPB * qstride;
  float* aux = lw + (a.kind < 0 ? 0 : (size_t)a.S * Q);

  // this thread's row: loads first
  const int t = tid;
  const bool live = t < a.S * Qn;
  const int s = live ? t / Qn : 0, q = live ? t - s * Qn : 0;
  const int bl = q / a.G, g = q - bl * a.G;
  const int64_t b = b0 + bl;
  const int64_t n = (int64_t)s * BG + row0 + q;
  const int64_t xrow = a.xpc ? row0 + q : b;
  f32x4 dm[D4], dr[D4], xx[D4], zz[Z4];
  uint32_t mm[D4];
  {
    const f32x4* dp = (const f32x4*)(a.dec + n * a.ld_dec);
    const f32x4* xp = (const f32x4*)(a.X + xrow * D);
    const uint32_t* mp = (const uint32_t*)(a.M + xrow * D);
    const f32x4* zp = (const f32x4*)(a.Z + n * a.ldz);
#pragma unroll
    for (int k = 0; k < D4; ++k) { dm[k] = dp[k]; dr[k] = dp[D4 + k]; xx[k] = xp[k]; mm[k] = mp[k]; }
#pragma unroll
    for (int k = 0; k < Z4; ++k) zz[k] = zp[k];
  }
  {
    const int per_dp = a.nC * Zd;
    for (int e = tid; e < nb * per_dp; e += nthr) {
      const int el = e / per_dp, rem = e - el * per_dp;
      const int c = rem / Zd, j = rem - c * Zd;
      const float sg = q_sigma(a.enc, a.q, b0 + el, c, j);
      const float rs = __builtin_amdgcn_rcpf(sg);
      qt[el * qstride + rem] = QEntry{q_mu(a.enc, a.q, b0 + el, c, j), 0.5f * rs * rs, dmv_log(sg) + LOG_SQRT_2PI, 0.f};
    }
  }
  __syncthreads();

  if (live) {
    const float geps = a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f;
    const bool own = a.q_kind != DMV_Q_STRATIFIED || a.G > 1;
    float gm = 0.f, gf = 0.f;
#pragma unroll
    for (int k = 0; k < D4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lp = log_normal_r(xx[k][e], dm[k][e], dmv_softplus(dr[k][e]) + geps);
        gf += lp;
        gm += lp * (((mm[k] >> (8 * e)) & 0xffu) ? 1.f : 0.f);
      }
    const QEntry* qb = qt + (size_t)bl * qstride;
    float pr = 0.f, qo = 0.f;
#pragma unroll
    for (int k = 0; k < Z4; ++k)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float zj = zz[k][e];
        pr += -(zj * zj) * 0.5f - LOG_SQRT_2PI;
        if (own) qo += q_logp(qb[g * Zd + 4 * k + e], zj);
      }
    float qm = qo;
    if (a.q_kind != DMV_Q_OWN) {
      float mx = -INFINITY, se = 0.f, lse_logit = 0.f;
      const float* __restrict__ lg = a.enc + b * a.q.sb + a.q.logit_off;
      if (a.q_kind == DMV_Q_STRATIFIED) {
        float lm = -INFINITY;
        for (int c = 0; c < a.mix_C; ++c) lm = fmaxf(lm, lg[c]);
        float ss = 0.f;
        for (int c = 0; c < a.mix_C; ++c) ss += dmv_exp(lg[c] - lm);
        lse_logit = lm + dmv_log(ss);
      }
      for (int c = 0; c < a.mix_C; ++c) {
        float lp = 0.f;
#pragma unroll
        for (int k = 0; k < Z4; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) lp += q_logp(qb[c * Zd + 4 * k + e], zz[k][e]);
        if (a.comp_lp) a.comp_lp[n * a.mix_C + c] = lp;
        const float tt = lp + (a.q_kind == DMV_Q_STRATIFIED ? lg[c] - lse_logit : 0.f);
        const float nm = fmaxf(mx, tt);
        se = se * dmv_exp(mx - nm) + dmv_exp(tt - nm);
        mx = nm;
      }
      qm = mx + dmv_log(se) - (a.q_kind == DMV_Q_IMPUTATION ? dmv_log((float)a.mix_C) : 0.f);
    }
    if (a.terms) {
      a.terms[n] = gm; a.terms[N + n] = gf; a.terms[2 * N + n] = pr; a.terms[3 * N + n] = qo; a.terms[4 * N + n] = qm;
    }
    if (per_comp) {
      const float t5[5] = {gm, gf, pr, qo, qm};
      float vA = 0.f, vB = 0.f;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        if (a.cA[k] != 0.f) vA += a.cA[k] * t5[k];
        if (a.cB[k] != 0.f) vB += a.cB[k] * t5[k];
      }
      lw[s * Q + q] = vA;
      aux[s * Q + q] = vB;
    } else if (a.kind >= 0) {
      const float w = gm + pr - (a.q_term == 3 ? qo : qm);
      lw[s * Q + q] = w;
      if (a.log_w) a.log_w[n] = w;
    }
  }
  if (a.kind < 0) return;
  float parked = 0.f;
  bound_phases(a, lw, aux, tid, nthr, b0, nb, Qn, Q, BG, row0, parked);
  finish_means(a, parked);
}

// the same means as a launch of their own (callers that bring no ticket counter): one workgroup of 1024 threads
__global__ void __launch_bounds__(1024) k_mean2(const float* __restrict__ v1, const float* __restrict__ v2, int64_t n, float scale,
                                                float* __restrict__ o1, float* __restrict__ o2, float* __restrict__ o1_neg) {
  __shared__ float sh[32];
  block_mean2(sh, v1, v2, n, scale, o1, o2, o1_neg);
}

int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

extern "C" int dmv_obj_forward(const DmvObjCfg* cfg, const DmvObjFwd* f, const float* dec, int64_t ld_dec, const float* X,
                               const uint8_t* M, const float* Z, int64_t ldz, const float* enc, void* stream) {
  if (!cfg || !f) return DMV_E_BADARG;
  if (cfg->B < 1 || cfg->G < 1 || cfg->nz < 1 || cfg->I < 1 || cfg->D < 1 || cfg->Zd < 1) return DMV_E_BADARG;
  if (cfg->G > 64) return DMV_E_UNSUPPORTED;
  if (cfg->gen_dist < DMV_GEN_NORMAL || cfg->gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
  if (!dec || !X || !M || !Z || !enc) return DMV_E_BADARG;
  if (f->q_kind < DMV_Q_OWN || f->q_kind > DMV_Q_IMPUTATION) return DMV_E_UNSUPPORTED;
  if (f->q_kind != DMV_Q_OWN && (f->mix_C < 1 || f->mix_C > 64)) return DMV_E_UNSUPPORTED;
  const int P = cfg->gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * cfg->D : cfg->D;
  if (ld_dec < P || ldz < cfg->Zd) return DMV_E_BADARG;
  const int kind = f->kind;
  if (kind < -1 || kind > DMV_BOUND_PER_COMPONENT) return DMV_E_UNSUPPORTED;
  if ((kind == DMV_BOUND_LSE || kind == DMV_BOUND_WEIGHTED) && !f->bound) return DMV_E_BADARG;
  if (kind == DMV_BOUND_PER_COMPONENT && (cfg->I != 1 || (!f->objA && !f->objB))) return DMV_E_BADARG;
  if (kind == -1 && !f->terms) return DMV_E_BADARG;

  FwdArgs a;
  a.B = cfg->B; a.G = cfg->G; a.nz = cfg->nz; a.I = cfg->I; a.S = cfg->nz * cfg->I; a.D = cfg->D; a.Zd = cfg->Zd;
  a.gen_dist = cfg->gen_dist; a.xpc = cfg->x_per_component; a.q_kind = f->q_kind; a.mix_C = f->mix_C;
  a.kind = kind; a.q_term = f->q_kind == DMV_Q_OWN ? 3 : 4; a.use_logpi = f->use_logpi; a.dreg = f->dreg;
  a.q.sb = cfg->enc_sb; a.q.sg = cfg->enc_sg; a.q.raw_off = cfg->raw_off; a.q.logit_off = cfg->logit_off;
  a.q.eps = cfg->q_eps ? DMV_EPSILON : 0.f;
  a.dec = dec; a.ld_dec = ld_dec; a.X = X; a.M = M; a.Z = Z; a.ldz = ldz; a.enc = enc;
  a.terms = f->terms; a.comp_lp = f->comp_lp; a.log_w = f->log_w; a.bound = f->bound; a.wgt = f->wgt;
  a.zscale = f->zscale; a.dlogit = f->dlogit; a.objA = f->objA; a.objB = f->objB;
  for (int t = 0; t < 5; ++t) { a.cA[t] = f->coefA[t]; a.cB[t] = f->coefB[t]; }
  a.vec4 = cfg->gen_dist >= DMV_GEN_BERN && (cfg->D & 3) == 0 && (ld_dec & 3) == 0 &&
           ((((uintptr_t)dec | (uintptr_t)X) & 15) == 0) && (((uintptr_t)M & 3) == 0);

  const int nC_full = f->q_kind == DMV_Q_OWN ? cfg->G : f->mix_C;
  const size_t lds_max = 60 * 1024;
  const int64_t pairs = (int64_t)cfg->B * cfg->G;
  hipStream_t st = (hipStream_t)stream;
  const bool want_means = f->mean_out || f->meanA_out || f->meanB_out || f->neg_mean_out;
  // (decided per launch below: the ticket's last workgroup sums the whole batch on its own, 6 us at 512 workgroups of 64
  // threads against 5 us for k_mean2 in a launch of its own: it pays for itself only in the short launches of small
  // batches)
  a.ticket = nullptr;
  auto use_ticket = [&](unsigned workgroups) {
    a.ticket = (want_means && kind >= 0 && workgroups <= 256) ? (unsigned*)f->ticket : nullptr;
  };
  a.mean1 = kind == DMV_BOUND_PER_COMPONENT ? f->meanA_out : f->mean_out;
  a.mean2 = kind == DMV_BOUND_PER_COMPONENT ? f->meanB_out : nullptr;
  a.neg1 = kind == DMV_BOUND_PER_COMPONENT ? nullptr : f->neg_mean_out;
  auto means = [&]() -> int {
    if (want_means && !a.ticket) {
      if (kind == DMV_BOUND_PER_COMPONENT) {
        hipLaunchKernelGGL(k_mean2, dim3(1), dim3(1024), 0, st, (const float*)f->objA, (const float*)f->objB, pairs,
                           1.f / (float)pairs, f->meanA_out, f->meanB_out, (float*)nullptr);
      } else if (kind >= 0) {
        hipLaunchKernelGGL(k_mean2, dim3(1), dim3(1024), 0, st, (const float*)f->bound, (const float*)nullptr, (int64_t)cfg->B,
                           1.f / (float)cfg->B, f->mean_out, (float*)nullptr, f->neg_mean_out);
      }
      DMV_HIP_CHECK_LAUNCH();
    }
    return DMV_OK;
  };

  // ---- short aligned Gaussian rows: one thread per row (every sample of a workgroup's datapoints in one pass)
  const int SG = a.S * cfg->G;
  const bool aligned4 = cfg->gen_dist <= DMV_GEN_NORMAL_EPS && (cfg->D & 3) == 0 && (cfg->Zd & 3) == 0 && (ld_dec & 3) == 0 &&
                        (ldz & 3) == 0 && ((((uintptr_t)dec | (uintptr_t)X | (uintptr_t)Z) & 15) == 0) && (((uintptr_t)M & 3) == 0);
  const int D4 = cfg->D >> 2, Z4 = cfg->Zd >> 2;
  if (aligned4 && SG <= 256 && (D4 == 1 || D4 == 2 || D4 == 4) && (Z4 == 1 || Z4 == 2 || Z4 == 4 || Z4 == 8)) {
    auto rows_lds = [&](int dpb) {
      const size_t Q = (size_t)dpb * cfg->G;
      const size_t bound_part = kind < 0 ? 0 : (size_t)a.S * Q + (kind == DMV_BOUND_PER_COMPONENT ? (size_t)a.S * Q : 3 * Q);
      return sizeof(float) * (bound_part + (size_t)4 * dpb * (nC_full * cfg->Zd + 1));
    };
    int nthr = pairs * a.S >= 256 * 512 ? 256 : 64;       // small batches: more, smaller workgroups
    int DPB = nthr / SG;
    if (DPB < 1) { nthr = 256; DPB = 256 / SG; }
    // a mixture posterior on a batch that does not fill the chip anyway: a workgroup's time is its chain of component
    // table, rows (every component per row) and logsumexp, which shortens with the datapoints it holds, so they are
    // spread over up to 256 workgroups (idle lanes cost nothing here; LAIR's weights at B = 512, 6 components: 18 -> 11 us.
    // Without a mixture the rows are short and the spread gains nothing: 14.0 against 14.7 us)
    if (nthr == 64 && f->q_kind != DMV_Q_OWN) { const int want = (cfg->B + 255) / 256; if (DPB > want) DPB = want; }
    while (DPB > 1 && rows_lds(DPB) > lds_max) --DPB;
    if (rows_lds(DPB) <= lds_max) {
      a.DPB = DPB; a.nC = nC_full;
      int T = pow2_ceil(cfg->I * cfg->G);
      if (T > 64) T = 64;
      if (T > nthr) T = nthr;
      a.T = T;
      const dim3 grid((unsigned)((cfg->B + DPB - 1) / DPB)), block(nthr);
      const size_t lds = rows_lds(DPB);
      use_ticket(grid.x);
#define DMV_ROWS4(d4, z4) \
  if (D4 == d4 && Z4 == z4) hipLaunchKernelGGL((k_obj_fwd_rows4<d4, z4>), grid, block, lds, st, a)
      DMV_ROWS4(1, 1); DMV_ROWS4(1, 2); DMV_ROWS4(1, 4); DMV_ROWS4(1, 8);
      DMV_ROWS4(2, 1); DMV_ROWS4(2, 2); DMV_ROWS4(2, 4); DMV_ROWS4(2, 8);
      DMV_ROWS4(4, 1); DMV_ROWS4(4, 2); DMV_ROWS4(4, 4); DMV_ROWS4(4, 8);
#undef DMV_ROWS4
      DMV_HIP_CHECK_LAUNCH();
      return means();
    }
  }

  // lanes per row by the longer of the two row kinds; workgroup size by how many datapoints there are
  const int span = cfg->D > cfg->Zd ? cfg->D : cfg->Zd;
  const int R = span <= 8 ? 1 : span <= 32 ? 8 : span <= 96 ? 16 : 64;
  int nthr = 256;
  if (pairs * R < 256 * 256) nthr = 64;            // small batches: more, smaller workgroups
  const int NG = nthr / R;
  int DPB = NG / cfg->G;
  if (DPB < 1) DPB = 1;
  {
    // keep every group busy in every round: if the (b, g) pairs of a workgroup do not fill whole rounds of NG groups
    // (less than 90 % of them), take the smallest count of datapoints that does
    const int Q0 = DPB * cfg->G, rounds = (Q0 + NG - 1) / NG;
    if (Q0 * 10 < rounds * NG * 9) {
      int x = NG, y = cfg->G;
      while (y) { const int t = x % y; x = y; y = t; }
      if (NG / x <= 64) DPB = NG / x;
    }
  }
  // ... unless the batch is too small to fill the chip anyway (as on the four-dimensions-per-lane path above): a
  // workgroup's time is then its chain of rounds, which shortens with the datapoints it holds, so a mixture posterior's
  // datapoints are spread over up to 512 workgroups (hepmass shapes at B = 512, 5 components: 8 datapoints = 5 rounds per
  // workgroup on 64 workgroups took 44 us, one datapoint per workgroup 14 us + 5 us for the mean in a launch of its own)
  if (nthr == 64 && f->q_kind != DMV_Q_OWN) { const int want = (cfg->B + 511) / 512; if (DPB > want) DPB = want; }
  // LDS: the q table (16 bytes per (datapoint, component, dimension)), S*Q log-weights (twice for the two objectives of
  // the per-component kind) and 3*Q floats of mixture bookkeeping
  auto lds_bytes = [&](int dpb, int nC) {
    const size_t Q = (size_t)dpb * cfg->G;
    const size_t bound_part = kind < 0 ? 0 : (size_t)a.S * Q * (kind == DMV_BOUND_PER_COMPONENT ? 2 : 1) + 3 * Q;
    return sizeof(float) * (bound_part + (size_t)4 * dpb * nC * cfg->Zd);
  };
  int nC = nC_full;
  while (DPB > 1 && lds_bytes(DPB, nC) > lds_max) DPB >>= 1;
  if (lds_bytes(DPB, nC) > lds_max) nC = 0;                       // too many components for a table: evaluate q directly
  if (lds_bytes(DPB, nC) > lds_max) return DMV_E_UNSUPPORTED;     // nz*I*G beyond ~15 000: stream with dmv_lse_update
  a.DPB = DPB; a.nC = nC;
  int T = pow2_ceil(cfg->I * cfg->G);
  if (T > 64) T = 64;
  if (T > nthr) T = nthr;
  a.T = T;
  dim3 grid((unsigned)((cfg->B + DPB - 1) / DPB));
  const size_t lds = lds_bytes(DPB, nC);
  if (kind < 0) {
    // terms only: nothing couples the samples of a datapoint, so few datapoints with many samples (the K-sample
    // log-likelihood) spread over the grid's second dimension
    const int64_t want = 2048 / grid.x;
    grid.y = (unsigned)(want < 1 ? 1 : want > a.S ? a.S : want > 65535 ? 65535 : want);
  }
  use_ticket(grid.x);
  switch (R) {
    case 1: hipLaunchKernelGGL(k_obj_fwd<1>, grid, dim3(nthr), lds, st, a); break;
    case 8: hipLaunchKernelGGL(k_obj_fwd<8>, grid, dim3(nthr), lds, st, a); break;
    case 16: hipLaunchKernelGGL(k_obj_fwd<16>, grid, dim3(nthr), lds, st, a); break;
    default: hipLaunchKernelGGL(k_obj_fwd<64>, grid, dim3(nthr), lds, st, a); break;
  }
  DMV_HIP_CHECK_LAUNCH();
  return means();
}
