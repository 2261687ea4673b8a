// Masked VAE / IWAE / mixture objectives: latent sampling, per-sample log-weight terms, bounds with
// wavefront logsumexp, and the hand-derived backward.  HBM-bound elementwise / reduction kernels.
//
// Sample rows are ordered n = (s*B + b)*G + g, s = z*I + i (the reference's [nz, I, B, (G,) .] tensors
// flattened).  See include/dmv.h for the encoder-output addressing.
#include "dmv_common.h"
#include "dmv_rng.h"
#include "dmv_objmath.h"

// ------------------------------------------------------------------------------------------
// z = mu + sigma * eps
// ------------------------------------------------------------------------------------------
struct SampleArgs {
  int64_t N;       // sample rows
  int S;           // N / (B G): outer x importance samples
  int B, G, Zd;
  QAddr q;
  const float* enc;
  const float* eps;        // [N][ldz] or null (Philox)
  float* eps_out;          // nullable: noise actually used
  const int64_t* comp_idx; // nullable: component per row (G must be 1)
  int64_t ldz;
  float* Z;
  uint64_t seed, offset;
  const uint64_t* offset_base;   // nullable device counter added to offset (advanced by the caller between replays)
};

// Launch geometry of the row kernels below: blockIdx.y = s (outer x importance sample), blockIdx.x * blockDim.y +
// threadIdx.y = the (b, g) pair, threadIdx.x = position inside the row.  Rows of one s are consecutive in memory, so
// consecutive threads touch consecutive addresses, and b, g come from one 32-bit division.
__global__ void k_latent_sample(SampleArgs a) {
  const int BG = a.B * a.G;
  const int bg = blockIdx.x * blockDim.y + threadIdx.y;
  if (bg >= BG) return;
  const int64_t b = bg / a.G;
  const int g0 = bg - (int)b * a.G;
  const uint64_t off = a.offset + (a.offset_base ? a.offset_base[0] : 0ull);
  for (int j = threadIdx.x; j < (int)a.ldz; j += blockDim.x) {
    // without per-row components the parameters of (b, g, j) serve every sample s: one softplus for all of them
    float mu = 0.f, sg = 0.f;
    if (j < a.Zd && !a.comp_idx) { mu = q_mu(a.enc, a.q, b, g0, j); sg = q_sigma(a.enc, a.q, b, g0, j); }
    for (int s = blockIdx.y; s < a.S; s += gridDim.y) {
      const int64_t n = (int64_t)s * BG + bg;
      const int64_t i = n * a.ldz + j;
      if (j >= a.Zd) { a.Z[i] = 0.f; if (a.eps_out) a.eps_out[i] = 0.f; continue; }
      if (a.comp_idx) { const int g = (int)a.comp_idx[n]; mu = q_mu(a.enc, a.q, b, g, j); sg = q_sigma(a.enc, a.q, b, g, j); }
      const float e = a.eps ? a.eps[i] : dmv_philox_normal(a.seed, off + (uint64_t)i);
      if (a.eps_out) a.eps_out[i] = e;
      a.Z[i] = mu + sg * e;
    }
  }
}

// ------------------------------------------------------------------------------------------
// per-sample terms
// ------------------------------------------------------------------------------------------
struct TermsArgs {
  int64_t N;
  int B, G, D, Zd;
  int gen_dist;
  int x_per_component;
  int q_kind, mix_C;
  QAddr q;
  const float* dec;
  int64_t ld_dec;
  const float* X;
  const uint8_t* M;
  const float* Z;
  int64_t ldz;
  const float* enc;
  float* terms;     // [5][N]
  float* comp_lp;   // nullable [N][mix_C]: log N(z_n; mu_{b,c}, sigma_{b,c}) summed over j
  int vec4;         // Bernoulli decoder with D % 4 == 0 and 16-byte aligned rows
};

template <int R>
__global__ void k_obj_terms(TermsArgs a) {
  const int r = threadIdx.x % R;
  const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / R;
  const bool live = n < a.N;
  const int64_t nn = live ? n : a.N - 1;
  const int g = (int)(nn % a.G);
  const int64_t b = (nn / a.G) % a.B;
  const int64_t xrow = a.x_per_component ? (b * a.G + g) : b;
  const float* dec = a.dec + nn * a.ld_dec;
  const float* X = a.X + xrow * a.D;
  const uint8_t* M = a.M + xrow * a.D;

  float gm = 0.f, gf = 0.f;
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
  } else
  for (int d = r; d < a.D; d += R) {
    float lp;
    const float x = X[d];
    if (a.gen_dist <= DMV_GEN_NORMAL_EPS) {
      const float s = dmv_softplus(dec[a.D + d]) + (a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f);
      lp = log_normal(x, dec[d], s);
    } else if (a.gen_dist == DMV_GEN_BERN) {
      lp = log_bern(x, dec[d]);
    } else {
      lp = log_bern((x + 1.f) * 0.5f, dec[d]);
    }
    gf += lp;
    gm += lp * (M[d] ? 1.f : 0.f);   // the product the reference forms (shared/iwae.py:20): a non-finite lp at a
                                     // missing entry makes the row's log-weight NaN there and here alike
  }
  const float* z = a.Z + nn * a.ldz;
  float pr = 0.f, qo = 0.f;
  for (int j = r; j < a.Zd; j += R) {
    const float zj = z[j];
    pr += -(zj * zj) * 0.5f - LOG_SQRT_2PI;
    if (a.q_kind != DMV_Q_STRATIFIED || a.G > 1)  // rows sampled from the mixture itself have no own component
      qo += log_normal(zj, q_mu(a.enc, a.q, b, g, j), q_sigma(a.enc, a.q, b, g, j));
  }
  gm = group_sum<R>(gm); gf = group_sum<R>(gf); pr = group_sum<R>(pr); qo = group_sum<R>(qo);

  float qm = qo;
  if (a.q_kind != DMV_Q_OWN) {
    // online logsumexp over the mixture components
    float mx = -INFINITY, se = 0.f, lse_logit = 0.f;
    if (a.q_kind == DMV_Q_STRATIFIED) {
      float lm = -INFINITY;
      for (int c = 0; c < a.mix_C; ++c) lm = fmaxf(lm, a.enc[b * a.q.sb + a.q.logit_off + c]);
      float ss = 0.f;
      for (int c = 0; c < a.mix_C; ++c) ss += expf(a.enc[b * a.q.sb + a.q.logit_off + c] - lm);
      lse_logit = lm + logf(ss);
    }
    for (int c = 0; c < a.mix_C; ++c) {
      float lp = 0.f;
      for (int j = r; j < a.Zd; j += R) lp += log_normal(z[j], q_mu(a.enc, a.q, b, c, j), q_sigma(a.enc, a.q, b, c, j));
      lp = group_sum<R>(lp);
      if (a.comp_lp && live && r == 0) a.comp_lp[nn * a.mix_C + c] = lp;
      const float t = lp + (a.q_kind == DMV_Q_STRATIFIED ? a.enc[b * a.q.sb + a.q.logit_off + c] - lse_logit : 0.f);
      const float nm = fmaxf(mx, t);
      se = se * expf(mx - nm) + expf(t - nm);
      mx = nm;
    }
    qm = mx + logf(se) - (a.q_kind == DMV_Q_IMPUTATION ? logf((float)a.mix_C) : 0.f);
  }
  if (live && r == 0) {
    a.terms[0 * a.N + n] = gm;
    a.terms[1 * a.N + n] = gf;
    a.terms[2 * a.N + n] = pr;
    a.terms[3 * a.N + n] = qo;
    a.terms[4 * a.N + n] = qm;
  }
}

// ------------------------------------------------------------------------------------------
// bounds: one wavefront per datapoint, logsumexp over the importance axis with wave shuffles
// ------------------------------------------------------------------------------------------
struct BoundArgs {
  int B, G, nz, I;
  int kind;        // DMV_BOUND_*
  int q_term;      // 3: own, 4: mixture  (index into terms)
  int use_logpi;   // LSE kind: add log pi_g of the stratified mixture
  int dreg;        // also emit zscale = softmax over i (G == 1)
  QAddr q;
  const float* enc;
  const float* terms;   // [5][N]
  int64_t N;
  float* log_w;         // nullable [N]
  float* bound;         // [B]  (per-datapoint bound; PER_COMPONENT: unused)
  float* wgt;           // nullable [N]: d(sum_b bound_b)/d log_w_n
  float* zscale;        // nullable [N]
  float* dlogit;        // nullable [B][G]: d(sum_b bound_b)/d logit_{b,g} through the log pi / pi factors
};

__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }
__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }

__global__ void k_obj_bound(BoundArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= a.B) return;
  const float* tgm = a.terms;
  const float* tpr = a.terms + 2 * a.N;
  const float* tq = a.terms + (int64_t)a.q_term * a.N;
  const int IG = a.I * a.G;
  const float logI = logf((float)a.I);

  // normalised log mixture weights of this datapoint (lane c holds component c, G <= 64)
  float logpi = 0.f;
  const bool need_pi = a.use_logpi || a.kind == DMV_BOUND_WEIGHTED;
  if (need_pi) {
    const float lg = lane < a.G ? a.enc[b * a.q.sb + a.q.logit_off + lane] : -INFINITY;
    const float mx = wave_max(lg);
    const float se = wave_sum(lane < a.G ? expf(lg - mx) : 0.f);
    logpi = lg - (mx + logf(se));
  }
  float bound_acc = 0.f;
  float dlog_acc = 0.f;   // lane c: gradient w.r.t. logit c

  // every loop below has a wave-uniform trip count (lanes past the end are predicated, not exited),
  // so the cross-lane broadcasts of log pi always read an active lane
  auto logw = [&](int64_t n) { return tgm[n] + tpr[n] - tq[n]; };
  for (int z = 0; z < a.nz; ++z) {
    if (a.kind == DMV_BOUND_LSE) {
      // logsumexp over (i, g)
      float mx = -INFINITY;
      for (int t0 = 0; t0 < IG; t0 += 64) {
        const int t = t0 + lane;
        const bool ok = t < IG;
        const int i = ok ? t / a.G : 0, g = ok ? t % a.G : 0;
        const int64_t n = ((int64_t)(z * a.I + i) * a.B + b) * a.G + g;
        const float lp = __shfl(logpi, g);
        if (ok) {
          float w = logw(n);
          if (a.log_w) a.log_w[n] = w;
          if (a.use_logpi) w += lp;
          mx = fmaxf(mx, w);
        }
      }
      mx = wave_max(mx);
      float se = 0.f;
      for (int t0 = 0; t0 < IG; t0 += 64) {
        const int t = t0 + lane;
        const bool ok = t < IG;
        const int i = ok ? t / a.G : 0, g = ok ? t % a.G : 0;
        const int64_t n = ((int64_t)(z * a.I + i) * a.B + b) * a.G + g;
        const float lp = __shfl(logpi, g);
        if (ok) se += expf(logw(n) + (a.use_logpi ? lp : 0.f) - mx);
      }
      se = wave_sum(se);
      const float lse = mx + logf(se);
      bound_acc += lse - logI;
      if (a.wgt || a.zscale || a.dlogit) {
        float pc_acc = 0.f;  // lane c: sum of the weights of component c's samples
        for (int t0 = 0; t0 < IG; t0 += 64) {
          const int t = t0 + lane;
          const bool ok = t < IG;
          const int i = ok ? t / a.G : 0, g = ok ? t % a.G : 0;
          const int64_t n = ((int64_t)(z * a.I + i) * a.B + b) * a.G + g;
          const float lp = __shfl(logpi, g);
          const float p = ok ? expf(logw(n) + (a.use_logpi ? lp : 0.f) - lse) : 0.f;
          if (ok) {
            if (a.wgt) a.wgt[n] = p / (float)a.nz;
            if (a.zscale) a.zscale[n] = a.dreg ? p : 1.f;
          }
          if (a.dlogit && a.use_logpi) {
            for (int c = 0; c < a.G; ++c) {
              const float s = wave_sum((ok && g == c) ? p : 0.f);
              if (lane == c) pc_acc += s;
            }
          }
        }
        // d/d logit_c of sum_n p_n log pi_{g(n)} = (sum_{n: g(n) = c} p_n) - pi_c
        if (a.dlogit && a.use_logpi && lane < a.G) dlog_acc += (pc_acc - expf(logpi)) / (float)a.nz;
      }
    } else if (a.kind == DMV_BOUND_WEIGHTED) {
      // sum_g pi_g (logsumexp_i log_w - log I)
      float tot = 0.f;
      float Lc = 0.f;   // lane c keeps L_c
      for (int g = 0; g < a.G; ++g) {
        const float pig = expf(__shfl(logpi, g));
        float mx = -INFINITY;
        for (int i = lane; i < a.I; i += 64) {
          const int64_t n = ((int64_t)(z * a.I + i) * a.B + b) * a.G + g;
          const float w = logw(n);
          if (a.log_w) a.log_w[n] = w;
          mx = fmaxf(mx, w);
        }
        mx = wave_max(mx);
        float se = 0.f;
        for (int i = lane; i < a.I; i += 64) {
          const int64_t n = ((int64_t)(z * a.I + i) * a.B + b) * a.G + g;
          se += expf(logw(n) - mx);
        }
        se = wave_sum(se);
        const float lse = mx + logf(se);
        tot += pig * (lse - logI);
        if (lane == g) Lc = lse - logI;
        if (a.wgt) {
          for (int i = lane; i < a.I; i += 64) {
            const int64_t n = ((int64_t)(z * a.I + i) * a.B + b) * a.G + g;
            a.wgt[n] = pig * expf(logw(n) - lse) / (float)a.nz;
            if (a.zscale) a.zscale[n] = 1.f;
          }
        }
      }
      bound_acc += tot;
      if (a.dlogit && lane < a.G) dlog_acc += expf(logpi) * (Lc - tot) / (float)a.nz;
    }
  }
  if (lane == 0) a.bound[b] = bound_acc / (float)a.nz;
  if (a.dlogit && lane < a.G) a.dlogit[b * a.G + lane] = dlog_acc;
}

// mean over B of bound -> out[0]  (single block; B is at most a few hundred thousand)
__global__ void k_mean(const float* v, int64_t n, float scale, float* out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) s += v[i];
  s = group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sh[0] + sh[1] + sh[2] + sh[3]) * scale;
}

// PER_COMPONENT objectives (DeMissVAE): obj[b*G+g] = mean_z( cm*gen_m + cf*gen_f + cp*prior + co*q_own + cx*q_mix )
struct CombineArgs {
  int64_t N;
  int B, G, nz;
  float c[5];
  const float* terms;
  float* obj;  // [B*G]
};
__global__ void k_obj_combine(CombineArgs a) {
  const int64_t bg = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t BG = (int64_t)a.B * a.G;
  if (bg >= BG) return;
  float s = 0.f;
  for (int z = 0; z < a.nz; ++z) {
    const int64_t n = (int64_t)z * BG + bg;
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < 5; ++t)
      if (a.c[t] != 0.f) v += a.c[t] * a.terms[(int64_t)t * a.N + n];
    s += v;
  }
  a.obj[bg] = s / (float)a.nz;
}

// ------------------------------------------------------------------------------------------
// backward, decoder side: d_dec[n][:] = gs * wgt[n] * (base_m * M_d + base_f) * d log p(x_d|z) / d params
// ------------------------------------------------------------------------------------------
struct BwdDecArgs {
  int64_t N;
  int B, G, D, S;
  int gen_dist, x_per_component;
  const float* dec;
  int64_t ld_dec;
  const float* X;
  const uint8_t* M;
  const float* wgt;     // nullable
  const float* wgt_f;   // nullable: separate per-row weight of the unmasked term (wgt then weighs the masked one only)
  const float* gb;      // nullable [B]: upstream gradient of each datapoint's bound, multiplied into the row weights
  const float* gscale;  // nullable device scalar
  float base_m, base_f;
  float* d_dec;
  int64_t ld_ddec;
};

// generic form: one thread per output element (row geometry of k_latent_sample: threadIdx.x walks the row)
__global__ void k_obj_bwd_dec(BwdDecArgs a) {
  const int BG = a.B * a.G;
  const int bg = blockIdx.x * blockDim.y + threadIdx.y;
  if (bg >= BG) return;
  const float gs = a.gscale ? a.gscale[0] : 1.f;
  const int b = bg / a.G;
  const int64_t xrow = a.x_per_component ? bg : b;
  const float gbb = gs * (a.gb ? a.gb[b] : 1.f);
  const bool normal = a.gen_dist <= DMV_GEN_NORMAL_EPS;
  const int P = normal ? 2 * a.D : a.D;
  for (int s = blockIdx.y; s < a.S; s += gridDim.y) {
    const int64_t n = (int64_t)s * BG + bg;
    const float wm = a.wgt ? a.wgt[n] : 1.f;
    const float wf = a.wgt_f ? a.wgt_f[n] : wm;
    const float* dec = a.dec + n * a.ld_dec;
    float* out = a.d_dec + n * a.ld_ddec;
    for (int c = threadIdx.x; c < (int)a.ld_ddec; c += blockDim.x) {
      if (c >= P) { out[c] = 0.f; continue; }
      const int d = c < a.D ? c : c - a.D;
      const float x = a.X[xrow * a.D + d];
      const float coef = gbb * ((a.M[xrow * a.D + d] ? wm * a.base_m : 0.f) + wf * a.base_f);
      float gr;
      if (normal) {
        const float raw = dec[a.D + d];
        const float sg = dmv_softplus(raw) + (a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f);
        const float df = x - dec[d];
        if (c < a.D) gr = df / (sg * sg);
        else gr = (df * df / (sg * sg * sg) - 1.f / sg) * (raw > 20.f ? 1.f : dmv_sigmoid(raw));
      } else {
        const float v = a.gen_dist == DMV_GEN_BERN ? x : (x + 1.f) * 0.5f;
        gr = v - dmv_sigmoid(dec[d]);
      }
      out[c] = coef * gr;
    }
  }
}

// Gaussian decoders: thread (row, d) reads the mean, the raw scale, the datum and the mask byte of dimension d once and
// writes both of its gradients (columns d and D + d); the padding columns [2D, pitch) are zeroed by the same threads.
// These rows are short (the kernel is bound by its arithmetic, not by HBM), so every lane gets one dimension's worth of
// transcendentals instead of a few lanes getting four; reciprocals instead of divisions.
__global__ void k_obj_bwd_dec_normal(BwdDecArgs a) {
  const int BG = a.B * a.G;
  const int bg = blockIdx.x * blockDim.y + threadIdx.y;
  if (bg >= BG) return;
  const float gs = a.gscale ? a.gscale[0] : 1.f;
  const int b = bg / a.G;
  const int64_t xrow = a.x_per_component ? bg : b;
  const float gbb = gs * (a.gb ? a.gb[b] : 1.f);
  const float eps = a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f;
  const int ld = (int)a.ld_ddec;
  for (int s = blockIdx.y; s < a.S; s += gridDim.y) {
    const int64_t n = (int64_t)s * BG + bg;
    const float wm = gbb * (a.wgt ? a.wgt[n] : 1.f);
    const float wf = a.wgt_f ? gbb * a.wgt_f[n] : wm;
    const float* dec = a.dec + n * a.ld_dec;
    float* out = a.d_dec + n * a.ld_ddec;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
      const float mu = dec[d], raw = dec[a.D + d];
      const float x = a.X[xrow * a.D + d];
      const float coef = (a.M[xrow * a.D + d] ? wm * a.base_m : 0.f) + wf * a.base_f;
      const float sg = dmv_softplus(raw) + eps;
      const float rs = __builtin_amdgcn_rcpf(sg);
      const float u = (x - mu) * rs;                       // (x - mu) / sigma
      out[d] = coef * (u * rs);                            // (x - mu) / sigma^2
      out[a.D + d] = coef * ((u * u - 1.f) * rs * (raw > 20.f ? 1.f : dmv_sigmoid(raw)));   // ((x-mu)^2/sigma^3 - 1/sigma) sigma'(raw)
    }
    for (int c = 2 * a.D + threadIdx.x; c < ld; c += blockDim.x) out[c] = 0.f;
  }
}

// Bernoulli decoders with wide, aligned rows (D % 4 == 0, ld_ddec == ld_dec rows of whole float4): blockIdx.y
// walks the (b, g) pairs, threads walk a row's float4 columns -- no per-element division, 16-byte accesses
__global__ void k_obj_bwd_dec_bern4(BwdDecArgs a) {
  const float gs = a.gscale ? a.gscale[0] : 1.f;
  const int D4 = a.D >> 2, W4 = (int)(a.ld_ddec >> 2);
  const int BG = a.B * a.G;
  for (int bg = blockIdx.y; bg < BG; bg += gridDim.y) {
    const int b = bg / a.G;
    const int64_t xrow = a.x_per_component ? bg : b;
    const float gbb = gs * (a.gb ? a.gb[b] : 1.f);
    const float* X = a.X + xrow * a.D;
    const uint8_t* M = a.M + xrow * a.D;
    for (int s = 0; s < a.S; ++s) {
      const int64_t n = (int64_t)s * BG + bg;
      const float w = gbb * (a.wgt ? a.wgt[n] : 1.f);
      const float wf = a.wgt_f ? gbb * a.wgt_f[n] : w;
      const float* dec = a.dec + n * a.ld_dec;
      f32x4* out = (f32x4*)(a.d_dec + n * a.ld_ddec);
      for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < W4; q += gridDim.x * blockDim.x) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (q < D4) {
          const f32x4 l4 = *(const f32x4*)(dec + 4 * q);
          const f32x4 x4 = *(const f32x4*)(X + 4 * q);
          const uint32_t m4 = *(const uint32_t*)(M + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = a.gen_dist == DMV_GEN_BERN ? x4[e] : (x4[e] + 1.f) * 0.5f;
            const float coef = (((m4 >> (8 * e)) & 0xffu) ? w * a.base_m : 0.f) + wf * a.base_f;
            o[e] = coef * (v - dmv_sigmoid(l4[e]));
          }
        }
        out[q] = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, latent side: gradient w.r.t. the encoder output (mu, raw sigma, mixture logits)
// one thread per (b, g', j) plus one per (b, g') logit
// ------------------------------------------------------------------------------------------
struct BwdLatArgs {
  int B, G, S, Zd;            // S = nz*I
  int64_t N;
  QAddr q;
  int mix_kind;               // DMV_Q_STRATIFIED / DMV_Q_IMPUTATION (used when base_qmix != 0)
  float base_prior, base_qown, base_qmix, base_w;
  int own_live, mix_live, use_zscale;
  float kl_coef;              // loss += kl_coef * sum_{b,g} kl_wgt[b,g] * KL(N(mu,sigma) || N(0,1))
  const float* kl_wgt;        // nullable [B][G] (1 when null)
  const float* enc;
  const float* Z;
  const float* eps;
  int64_t ldz;
  const float* dZ;            // nullable, decoder input gradient [N][ld_dz]
  int64_t ld_dz;
  const float* wgt;           // nullable
  const float* wgt_qown;      // nullable: per-row weight of the q_own term (wgt when null)
  const float* wgt_qmix;      // nullable: per-row weight of the q_mix term (wgt when null)
  const float* zscale;        // nullable
  const float* comp_lp;       // [N][G] when base_qmix != 0
  const float* qmix;          // terms[4]
  const float* dlogit;        // nullable [B][G]
  const float* gb;            // nullable [B]: upstream gradient per datapoint, multiplied into every row weight and
                              // into dlogit (not into dZ, which carries it already, nor into the KL term)
  const float* gscale;        // nullable device scalar: multiplies what gb multiplies, and the KL term
  float* d_enc;               // same addressing as enc
  // entries of d_enc that no (b, g, j) owns are zeroed here: zf_nseg segments of zf_len floats per datapoint, the
  // first zf_used of each hold gradients
  int zf_nseg, zf_len, zf_used;
};

__global__ void k_obj_bwd_latent(BwdLatArgs a) {
  const int per_b = a.G * (a.Zd + 1);
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // B * per_b < 2^31 (checked by the caller)
  if (idx >= a.B * per_b) return;
  const int64_t b = idx / per_b;
  const int rem = idx - (int)b * per_b;
  const int gp = rem / (a.Zd + 1);
  const int j = rem - gp * (a.Zd + 1);
  const float gs = a.gscale ? a.gscale[0] : 1.f;
  const float gbb = gs * (a.gb ? a.gb[b] : 1.f);
  {
    const int pad = a.zf_len - a.zf_used;
    for (int e = rem; e < a.zf_nseg * pad; e += per_b) {
      const int seg = e / pad;
      a.d_enc[b * a.q.sb + (int64_t)seg * a.zf_len + a.zf_used + (e - seg * pad)] = 0.f;
    }
  }
  const bool strat = a.mix_kind == DMV_Q_STRATIFIED;
  const bool has_mix = a.base_qmix != 0.f;

  // normalised log mixture weights (uniform for the imputation mixture)
  float lse_logit = 0.f;
  if (has_mix && strat) {
    float lm = -INFINITY;
    for (int c = 0; c < a.G; ++c) lm = fmaxf(lm, a.enc[b * a.q.sb + a.q.logit_off + c]);
    float ss = 0.f;
    for (int c = 0; c < a.G; ++c) ss += expf(a.enc[b * a.q.sb + a.q.logit_off + c] - lm);
    lse_logit = lm + logf(ss);
  }
  const float log_unif = -logf((float)a.G);
  auto logpi = [&](int c) { return strat ? a.enc[b * a.q.sb + a.q.logit_off + c] - lse_logit : log_unif; };

  if (j == a.Zd) {
    // ---- mixture logit (b, gp)
    if (!strat) return;
    float acc = a.dlogit ? a.base_w * gbb * a.dlogit[b * a.G + gp] : 0.f;
    if (has_mix && a.mix_live) {
      const float pi = expf(logpi(gp));
      for (int s = 0; s < a.S; ++s)
        for (int g = 0; g < a.G; ++g) {
          const int64_t n = ((int64_t)s * a.B + b) * a.G + g;
          const float cw = a.base_qmix * gbb * (a.wgt_qmix ? a.wgt_qmix[n] : a.wgt ? a.wgt[n] : 1.f);
          const float r = expf(logpi(gp) + a.comp_lp[n * a.G + gp] - a.qmix[n]);
          acc += cw * (r - pi);
        }
    }
    a.d_enc[b * a.q.sb + a.q.logit_off + gp] = acc;
    return;
  }

  const float mu = q_mu(a.enc, a.q, b, gp, j);
  const float raw = a.enc[b * a.q.sb + gp * a.q.sg + a.q.raw_off + j];
  const float sg = dmv_softplus(raw) + a.q.eps;
  float acc_mu = 0.f, acc_sg = 0.f;
  if (!has_mix) {
    // Own-Gaussian q (no mixture term): the operands of eight samples are requested before the first is used, so a
    // thread waits for memory once per eight samples instead of once per sample (small batches are bound by that
    // latency chain, not by bandwidth).  Same operations in the same order as the general loop below.
    // every load below is unconditional (a clamped sample index, and a valid stand-in address where an optional array
    // is absent) so that the compiler issues all of them before the first wait; the selects come afterwards
    const float* dZp = a.dZ ? a.dZ : a.Z;
    const int64_t ld_dzp = a.dZ ? a.ld_dz : a.ldz;
    const float* wp = a.wgt ? a.wgt : a.Z;
    const float* wop = a.wgt_qown ? a.wgt_qown : wp;
    const bool zs_on = a.use_zscale && a.zscale;
    const float* zsp = zs_on ? a.zscale : a.Z;
    for (int s0 = 0; s0 < a.S; s0 += 8) {
      float zv[8], dzv[8], ev[8], wv[8], wov[8], zsv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int sk = min(s0 + k, a.S - 1);
        const int64_t n = ((int64_t)sk * a.B + b) * a.G + gp;
        zv[k] = a.Z[n * a.ldz + j];
        dzv[k] = dZp[n * ld_dzp + j];
        ev[k] = a.eps[n * a.ldz + j];
        wv[k] = wp[n];
        wov[k] = wop[n];
        zsv[k] = zsp[n];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool ok = s0 + k < a.S;
        if (!a.dZ) dzv[k] = 0.f;
        if (!a.wgt) wv[k] = 1.f;
        if (!a.wgt_qown) wov[k] = wv[k];
        if (!zs_on) zsv[k] = 1.f;
        if (!ok) { zv[k] = 0.f; dzv[k] = 0.f; ev[k] = 0.f; wv[k] = 0.f; wov[k] = 0.f; zsv[k] = 1.f; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float w = gbb * wv[k], wo = gbb * wov[k], z = zv[k];
        float dz = dzv[k];
        dz += a.base_prior * w * (-z);
        if (a.base_qown != 0.f) dz += a.base_qown * wo * (-(z - mu) / (sg * sg));
        dz *= zsv[k];
        acc_mu += dz;
        acc_sg += dz * ev[k];
        if (a.own_live && a.base_qown != 0.f) {
          const float df = z - mu;
          acc_mu += a.base_qown * wo * (df / (sg * sg));
          acc_sg += a.base_qown * wo * (df * df / (sg * sg * sg) - 1.f / sg);
        }
      }
    }
  } else
  for (int s = 0; s < a.S; ++s) {
    // (1) reparameterised path: only samples drawn from component gp depend on (mu, sigma)_{gp}
    {
      const int64_t n = ((int64_t)s * a.B + b) * a.G + gp;
      const float w = gbb * (a.wgt ? a.wgt[n] : 1.f);
      const float wo = a.wgt_qown ? gbb * a.wgt_qown[n] : w;
      const float z = a.Z[n * a.ldz + j];
      float dz = a.dZ ? a.dZ[n * a.ld_dz + j] : 0.f;
      dz += a.base_prior * w * (-z);
      if (a.base_qown != 0.f) dz += a.base_qown * wo * (-(z - mu) / (sg * sg));
      if (has_mix) {
        float t = 0.f;
        for (int c = 0; c < a.G; ++c) {
          const float r = expf(logpi(c) + a.comp_lp[n * a.G + c] - a.qmix[n]);
          const float mc = q_mu(a.enc, a.q, b, c, j), sc = q_sigma(a.enc, a.q, b, c, j);
          t += r * (-(z - mc) / (sc * sc));
        }
        dz += a.base_qmix * (a.wgt_qmix ? gbb * a.wgt_qmix[n] : w) * t;
      }
      if (a.use_zscale && a.zscale) dz *= a.zscale[n];
      acc_mu += dz;
      acc_sg += dz * a.eps[n * a.ldz + j];
      // (2) direct dependence of log q_own on its parameters
      if (a.own_live && a.base_qown != 0.f) {
        const float df = z - mu;
        acc_mu += a.base_qown * wo * (df / (sg * sg));
        acc_sg += a.base_qown * wo * (df * df / (sg * sg * sg) - 1.f / sg);
      }
    }
    // (3) direct dependence of log q_mix (evaluated at every sample of the datapoint) on component gp
    if (has_mix && a.mix_live) {
      for (int g = 0; g < a.G; ++g) {
        const int64_t n = ((int64_t)s * a.B + b) * a.G + g;
        const float w = gbb * (a.wgt_qmix ? a.wgt_qmix[n] : a.wgt ? a.wgt[n] : 1.f);
        const float z = a.Z[n * a.ldz + j];
        const float r = expf(logpi(gp) + a.comp_lp[n * a.G + gp] - a.qmix[n]);
        const float df = z - mu;
        acc_mu += a.base_qmix * w * r * (df / (sg * sg));
        acc_sg += a.base_qmix * w * r * (df * df / (sg * sg * sg) - 1.f / sg);
      }
    }
  }
  if (a.kl_coef != 0.f) {
    const float kc = a.kl_coef * gs * (a.kl_wgt ? a.kl_wgt[b * a.G + gp] : 1.f);
    acc_mu += kc * mu;
    acc_sg += kc * (sg - 1.f / sg);
  }
  a.d_enc[b * a.q.sb + gp * a.q.sg + j] = acc_mu;
  a.d_enc[b * a.q.sb + gp * a.q.sg + a.q.raw_off + j] = acc_sg * (raw > 20.f ? 1.f : dmv_sigmoid(raw));
}

// The same gradient when a mixture term is present (stratified mixture: shared/iwae.py:102-118,
// models/multiple_iwae.py:124-140; imputation mixture: shared/iwae.py:119-132, models/mvbvae.py:95-109), for
// datapoints with G * Zd <= 1024.  A workgroup owns nb consecutive datapoints, one thread per (datapoint, g', j).
// What the G * Zd threads of a datapoint share is computed once and exchanged through LDS: the components' mu and
// 1 / sigma^2, the normalised log mixture weights, and per sample group s the latent samples z[g][j], the mixture
// term's row weights and the responsibilities r[g][c] = pi_c N_c(z_g) / q_mix(z_g).  All global operands of a sample
// group are requested before the first barrier, so a thread waits for memory twice per group (operands, then the
// component log-densities of responsibilities beyond the first per thread), not once per (sample, component).
// Same terms in the same order as the mixture branch of k_obj_bwd_latent, which stays for larger datapoints.
template <bool STRAT>
__global__ __launch_bounds__(1024) void k_obj_bwd_latent_mix(BwdLatArgs a, int nb, int per_b) {
  extern __shared__ float lds[];
  const int G = a.G, Zd = a.Zd;
  const int stride = 3 * per_b + 2 * G + G * G;
  const int lb = (int)threadIdx.x / per_b;
  const int e = (int)threadIdx.x - lb * per_b;
  const int64_t b = (int64_t)blockIdx.x * nb + lb;
  const bool active = lb < nb && b < a.B;
  float* smu = lds + (active ? lb : 0) * stride;      // [G][Zd]
  float* sis2 = smu + per_b;                          // [G][Zd]  1 / sigma^2
  float* sz = sis2 + per_b;                           // [G][Zd]  samples of the current group
  float* slp = sz + per_b;                            // [G]      log pi
  float* sw = slp + G;                                // [G]      base_qmix * upstream * row weight of the mixture term
  float* sr = sw + G;                                 // [G][G]   responsibilities of the current group
  const int gp = e / Zd, j = e - gp * Zd;
  const int GG = G * G;
  const float gs = a.gscale ? a.gscale[0] : 1.f;
  float gbb = gs, mu = 0.f, raw = 0.f, sg = 1.f, is2 = 1.f, isg = 1.f;
  float acc_mu = 0.f, acc_sg = 0.f, acc_lg = 0.f;
  for (int s = 0; s < a.S; ++s) {
    // ---- requests of this sample group
    float z = 0.f, dzv = 0.f, ep = 0.f, wv = 1.f, wov = 1.f, zs = 1.f, wq = 1.f, clp = 0.f, qm = 0.f;
    if (active) {
      const int64_t n = ((int64_t)s * a.B + b) * G + gp;
      z = a.Z[n * a.ldz + j];
      ep = a.eps[n * a.ldz + j];
      if (a.dZ) dzv = a.dZ[n * a.ld_dz + j];
      if (a.wgt) wv = a.wgt[n];
      wov = a.wgt_qown ? a.wgt_qown[n] : wv;
      if (a.use_zscale && a.zscale) zs = a.zscale[n];
      if (e < G) {
        const int64_t ng = ((int64_t)s * a.B + b) * G + e;
        wq = a.wgt_qmix ? a.wgt_qmix[ng] : a.wgt ? a.wgt[ng] : 1.f;
      }
      if (e < GG) {
        const int g = e / G;
        const int64_t ng = ((int64_t)s * a.B + b) * G + g;
        clp = a.comp_lp[ng * G + (e - g * G)];
        qm = a.qmix[ng];
      }
    }
    if (s == 0 && active) {
      // ---- once per datapoint: its padding entries, parameters and log mixture weights
      if (a.gb) gbb = gs * a.gb[b];
      const int pad = a.zf_len - a.zf_used;
      for (int k = e; k < a.zf_nseg * pad; k += per_b) {
        const int seg = k / pad;
        a.d_enc[b * a.q.sb + (int64_t)seg * a.zf_len + a.zf_used + (k - seg * pad)] = 0.f;
      }
      mu = q_mu(a.enc, a.q, b, gp, j);
      raw = a.enc[b * a.q.sb + gp * a.q.sg + a.q.raw_off + j];
      sg = dmv_softplus(raw) + a.q.eps;
      isg = 1.f / sg;
      is2 = 1.f / (sg * sg);
      smu[e] = mu;
      sis2[e] = is2;
      if (e < G) {
        if (STRAT) {
          const float* lg = a.enc + b * a.q.sb + a.q.logit_off;
          float lm = -INFINITY;
          for (int c = 0; c < G; ++c) lm = fmaxf(lm, lg[c]);
          float ss = 0.f;
          for (int c = 0; c < G; ++c) ss += expf(lg[c] - lm);
          slp[e] = lg[e] - (lm + logf(ss));
          if (a.dlogit) acc_lg = a.base_w * gbb * a.dlogit[b * G + e];
        } else {
          slp[e] = -logf((float)G);
        }
      }
    }
    if (active) {
      sz[e] = z;
      if (e < G) sw[e] = a.base_qmix * gbb * wq;
    }
    __syncthreads();
    if (active) {
      if (e < GG) sr[e] = expf(slp[e - (e / G) * G] + clp - qm);
      for (int k = e + per_b; k < GG; k += per_b) {
        const int g = k / G, c = k - g * G;
        const int64_t ng = ((int64_t)s * a.B + b) * G + g;
        sr[k] = expf(slp[c] + a.comp_lp[ng * G + c] - a.qmix[ng]);
      }
    }
    __syncthreads();
    if (active) {
      // (1) reparameterised path: only samples drawn from component gp depend on (mu, sigma)_{gp}
      const float w = gbb * wv, wo = gbb * wov;
      float dz = dzv + a.base_prior * w * (-z);
      if (a.base_qown != 0.f) dz += a.base_qown * wo * (-(z - mu) * is2);
      float t = 0.f;
      for (int c = 0; c < G; ++c) t += sr[gp * G + c] * (-(z - smu[c * Zd + j]) * sis2[c * Zd + j]);
      dz += sw[gp] * t;
      dz *= zs;
      acc_mu += dz;
      acc_sg += dz * ep;
      // (2) direct dependence of log q_own on its parameters
      if (a.own_live && a.base_qown != 0.f) {
        const float df = z - mu;
        acc_mu += a.base_qown * wo * (df * is2);
        acc_sg += a.base_qown * wo * (df * df * is2 * isg - isg);
      }
      // (3) direct dependence of log q_mix (evaluated at every sample of the datapoint) on component gp
      if (a.mix_live) {
        for (int g = 0; g < G; ++g) {
          const float wr = sw[g] * sr[g * G + gp];
          const float df = sz[g * Zd + j] - mu;
          acc_mu += wr * (df * is2);
          acc_sg += wr * (df * df * is2 * isg - isg);
        }
        if (STRAT && e < G) {
          const float pi = expf(slp[e]);
          for (int g = 0; g < G; ++g) acc_lg += sw[g] * (sr[g * G + e] - pi);
        }
      }
    }
    if (s + 1 < a.S) __syncthreads();
  }
  if (!active) return;
  if (a.kl_coef != 0.f) {
    const float kc = a.kl_coef * gs * (a.kl_wgt ? a.kl_wgt[b * G + gp] : 1.f);
    acc_mu += kc * mu;
    acc_sg += kc * (sg - isg);
  }
  a.d_enc[b * a.q.sb + gp * a.q.sg + j] = acc_mu;
  a.d_enc[b * a.q.sb + gp * a.q.sg + a.q.raw_off + j] = acc_sg * (raw > 20.f ? 1.f : dmv_sigmoid(raw));
  if (STRAT && e < G) a.d_enc[b * a.q.sb + a.q.logit_off + e] = acc_lg;
}

// ------------------------------------------------------------------------------------------
// Implicit reparameterisation gradients of mixture samples
// (vgiwae/shared/reparametrized_mixture_same_family.py:27-191).  z ~ sum_k pi_k N(mu_k, diag sigma_k^2) is drawn
// ancestrally; its pathwise derivative is dz/dphi = -J^-1 dT/dphi with the distributional transform
//   T_i(z) = sum_k w_ik Phi((z_i - mu_ki) / sigma_ki),   w_ik = softmax_k(logit_k + sum_{j<i} log N(z_j; mu_kj, sigma_kj))
// and J = dT/dz (lower triangular).  For a loss gradient g = dL/dz the parameter gradient is -u^T dT/dphi with
// J^T u = g.  Both the back substitution and the vector-Jacobian product collapse into one sweep j = Zd-1 .. 0
// with K running sums A_k = sum_{i>j} u_i w_ik (Phi_ik - T_i):
//   u_j        = (g_j - sum_k d_kj A_k) / sum_k w_jk pdf_jk,          d_kj = -(z_j - mu_kj) / sigma_kj^2
//   dL/dmu_kj   = u_j w_jk pdf_jk + A_k d_kj
//   dL/dsig_kj  = u_j w_jk pdf_jk a_jk - A_k ((z_j - mu_kj)^2 / sigma_kj^3 - 1 / sigma_kj),   a = (z - mu) / sigma
//   dL/dlogit_k = -sum_j u_j w_jk (Phi_jk - T_j)
// i.e. O(K Zd) per sample instead of the reference's Zd x Zd Jacobian and triangular solve.
// g_j = dZ[n][j] + wgt[n] (base_prior (-z_j) + base_qmix dlog q_mix(z)/dz_j).
// One thread per datapoint walks its S samples in order (fixed summation order, no atomics); the prefix sums
// live in ws [B][K][Zd + 1].
// ------------------------------------------------------------------------------------------
struct BwdImplicitArgs {
  int B, S, Zd, K;
  QAddr q;
  float base_prior, base_qmix;
  int mix_live;         // also differentiate log q_mix(z) in q's parameters at fixed z
  const float* enc;
  const float* Z;
  int64_t ldz;
  const float* dZ;      // nullable
  int64_t ld_dz;
  const float* wgt;     // nullable
  float* ws;
  float* d_enc;
};

__global__ void k_obj_bwd_latent_implicit(BwdImplicitArgs a) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= a.B) return;
  const int K = a.K, Zd = a.Zd;
  const float* enc = a.enc + b * a.q.sb;
  float* out = a.d_enc + b * a.q.sb;
  float* pre = a.ws + b * (int64_t)K * (Zd + 1);
  for (int k = 0; k < K; ++k) {
    out[a.q.logit_off + k] = 0.f;
    for (int j = 0; j < Zd; ++j) { out[k * a.q.sg + j] = 0.f; out[k * a.q.sg + a.q.raw_off + j] = 0.f; }
  }
  float A[64];
  for (int s = 0; s < a.S; ++s) {
    const int64_t n = (int64_t)s * a.B + b;
    const float* z = a.Z + n * a.ldz;
    const float w_n = a.wgt ? a.wgt[n] : 1.f;
    // prefix sums of the components' log densities
    for (int k = 0; k < K; ++k) {
      float acc = 0.f;
      pre[k * (Zd + 1)] = 0.f;
      for (int j = 0; j < Zd; ++j) {
        const float sg = dmv_softplus(enc[k * a.q.sg + a.q.raw_off + j]) + a.q.eps;
        acc += log_normal(z[j], enc[k * a.q.sg + j], sg);
        pre[k * (Zd + 1) + j + 1] = acc;
      }
    }
    // responsibilities of the whole sample (for d log q_mix / dz)
    float tm = -INFINITY;
    for (int k = 0; k < K; ++k) tm = fmaxf(tm, enc[a.q.logit_off + k] + pre[k * (Zd + 1) + Zd]);
    float tz = 0.f;
    for (int k = 0; k < K; ++k) tz += expf(enc[a.q.logit_off + k] + pre[k * (Zd + 1) + Zd] - tm);
    for (int k = 0; k < K; ++k) A[k] = 0.f;
    // at fixed z: d log q_mix / d logit_k = r_k - pi_k, and r_k d log N_k / d(mu, sigma) inside the loop over j
    const float live = a.mix_live ? w_n * a.base_qmix : 0.f;
    if (a.mix_live) {
      float lm = -INFINITY;
      for (int k = 0; k < K; ++k) lm = fmaxf(lm, enc[a.q.logit_off + k]);
      float lz = 0.f;
      for (int k = 0; k < K; ++k) lz += expf(enc[a.q.logit_off + k] - lm);
      for (int k = 0; k < K; ++k) {
        const float r = expf(enc[a.q.logit_off + k] + pre[k * (Zd + 1) + Zd] - tm) / tz;
        out[a.q.logit_off + k] += live * (r - expf(enc[a.q.logit_off + k] - lm) / lz);
      }
    }
    for (int j = Zd - 1; j >= 0; --j) {
      float wm = -INFINITY;
      for (int k = 0; k < K; ++k) wm = fmaxf(wm, enc[a.q.logit_off + k] + pre[k * (Zd + 1) + j]);
      float wz = 0.f;
      for (int k = 0; k < K; ++k) wz += expf(enc[a.q.logit_off + k] + pre[k * (Zd + 1) + j] - wm);
      float T = 0.f, pj = 0.f, dq = 0.f, dA = 0.f;
      for (int k = 0; k < K; ++k) {
        const float mu = enc[k * a.q.sg + j];
        const float sg = dmv_softplus(enc[k * a.q.sg + a.q.raw_off + j]) + a.q.eps;
        const float al = (z[j] - mu) / sg;
        const float w = expf(enc[a.q.logit_off + k] + pre[k * (Zd + 1) + j] - wm) / wz;
        const float cdf = 0.5f * (1.f + erff(al * 0.70710678118654752f));
        const float pdf = expf(-0.5f * al * al) * 0.3989422804014327f / sg;
        const float d = -al / sg;
        const float r = expf(enc[a.q.logit_off + k] + pre[k * (Zd + 1) + Zd] - tm) / tz;
        T += w * cdf; pj += w * pdf; dq += r * d; dA += d * A[k];
      }
      const float g = (a.dZ ? a.dZ[n * a.ld_dz + j] : 0.f) + w_n * (a.base_prior * (-z[j]) + a.base_qmix * dq);
      const float u = (g - dA) / pj;
      for (int k = 0; k < K; ++k) {
        const float mu = enc[k * a.q.sg + j];
        const float raw = enc[k * a.q.sg + a.q.raw_off + j];
        const float sg = dmv_softplus(raw) + a.q.eps;
        const float al = (z[j] - mu) / sg;
        const float w = expf(enc[a.q.logit_off + k] + pre[k * (Zd + 1) + j] - wm) / wz;
        const float cdf = 0.5f * (1.f + erff(al * 0.70710678118654752f));
        const float pdf = expf(-0.5f * al * al) * 0.3989422804014327f / sg;
        const float d = -al / sg;
        const float rl = a.mix_live ? live * expf(enc[a.q.logit_off + k] + pre[k * (Zd + 1) + Zd] - tm) / tz : 0.f;
        out[k * a.q.sg + j] += u * w * pdf + (A[k] - rl) * d;
        const float gs = u * w * pdf * al - (A[k] - rl) * (al * al / sg - 1.f / sg);
        out[k * a.q.sg + a.q.raw_off + j] += gs * (raw > 20.f ? 1.f : dmv_sigmoid(raw));
        const float e = u * w * (cdf - T);
        out[a.q.logit_off + k] -= e;
        A[k] += e;
      }
    }
  }
}

// analytic -KL(N(mu,sigma)||N(0,1)) summed over j, per (b, g):  models/vae.py:404-406
struct KlArgs { int B, G, Zd; QAddr q; const float* enc; float* kl; };
__global__ void k_kl(KlArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)a.B * a.G) return;
  const int64_t b = i / a.G; const int g = (int)(i % a.G);
  float s = 0.f;
  for (int j = 0; j < a.Zd; ++j) {
    const float mu = q_mu(a.enc, a.q, b, g, j), sg = q_sigma(a.enc, a.q, b, g, j);
    const float v = sg * sg;
    s += 0.5f * (v + mu * mu - 1.f - logf(v));
  }
  a.kl[i] = s;
}

// ------------------------------------------------------------------------------------------
// streaming logsumexp for the K-sample log-likelihood (test/refit_...py:189-222): fold a chunk of
// log-weights [Kc][B] into running (max, sum exp) per datapoint.  One block per datapoint.
// ------------------------------------------------------------------------------------------
struct LseArgs {
  int B; int64_t Kc;
  const float* gm; const float* pr; const float* q;   // term arrays of the chunk, each [Kc*B]
  float* run_max; float* run_sum;   // [B]; initialise to -inf / 0
};
__global__ void k_lse_update(LseArgs a) {
  const int b = blockIdx.x;
  __shared__ float shm[4], shs[4];
  float mx = -INFINITY;
  for (int64_t k = threadIdx.x; k < a.Kc; k += blockDim.x) {
    const int64_t n = k * a.B + b;
    mx = fmaxf(mx, a.gm[n] + a.pr[n] - a.q[n]);
  }
  mx = group_max<64>(mx);
  if ((threadIdx.x & 63) == 0) shm[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(shm[0], shm[1]), fmaxf(shm[2], shm[3]));
  const float old = a.run_max[b];
  const float nm = fmaxf(mx, old);
  float se = 0.f;
  for (int64_t k = threadIdx.x; k < a.Kc; k += blockDim.x) {
    const int64_t n = k * a.B + b;
    se += expf(a.gm[n] + a.pr[n] - a.q[n] - nm);
  }
  se = group_sum<64>(se);
  if ((threadIdx.x & 63) == 0) shs[threadIdx.x >> 6] = se;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float prev = (old == -INFINITY) ? 0.f : a.run_sum[b] * expf(old - nm);
    a.run_sum[b] = prev + shs[0] + shs[1] + shs[2] + shs[3];
    a.run_max[b] = nm;
  }
}

// ------------------------------------------------------------------------------------------
// Marginal log-probability by numerical integration over a latent grid
// (vgiwae/shared/vae_marginal_logprob.py:194-307): for every datapoint
//   log p(x_o) ~= logsumexp_z [ log p(z) + log dz + sum_d M_d log p(x_d | z) ]      (and the same without the mask)
// with the decoder evaluated once per grid point.  k_grid_prep turns the decoder outputs into three numbers per
// (grid point, dimension) so that the B x Nz x D inner loop is a fused multiply-add; k_grid_marginal runs one
// block per datapoint with an online (max, sum-exp) per thread and a block reduction.
// ------------------------------------------------------------------------------------------
struct GridArgs {
  int64_t Nz;
  int D, B, gen_dist;
  const float* dec;
  int64_t ld_dec;
  const float* logw;     // [Nz] log prior + log cell volume
  const float* X;
  const uint8_t* M;
  float* ws;             // [Nz][D][3]
  float* out_obs;
  float* out_full;       // nullable
};

__global__ void k_grid_prep(GridArgs a) {
  const int64_t total = a.Nz * a.D;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t z = i / a.D;
    const int d = (int)(i - z * a.D);
    const float* dec = a.dec + z * a.ld_dec;
    float p0, p1, c;
    if (a.gen_dist <= DMV_GEN_NORMAL_EPS) {
      const float s = dmv_softplus(dec[a.D + d]) + (a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f);
      p0 = dec[d]; p1 = 1.f / (2.f * s * s); c = -logf(s) - LOG_SQRT_2PI;
    } else {
      const float l = dec[d];
      p0 = l; p1 = 0.f; c = fminf(-l, 0.f) - log1pf(expf(-fabsf(l)));      // log(1 - sigmoid(l)): log p(x = 0)
    }
    a.ws[3 * i + 0] = p0; a.ws[3 * i + 1] = p1; a.ws[3 * i + 2] = c;
  }
}

__global__ void k_grid_marginal(GridArgs a) {
  const int b = blockIdx.x;
  const float* X = a.X + (int64_t)b * a.D;
  const uint8_t* M = a.M + (int64_t)b * a.D;
  const bool normal = a.gen_dist <= DMV_GEN_NORMAL_EPS;
  float mo = -INFINITY, so = 0.f, mf = -INFINITY, sf = 0.f;
  for (int64_t z = threadIdx.x; z < a.Nz; z += blockDim.x) {
    const float* w = a.ws + z * a.D * 3;
    float lo = a.logw[z], lf = lo;
    for (int d = 0; d < a.D; ++d) {
      const float x = X[d];
      float lp;
      if (normal) { const float df = x - w[3 * d]; lp = w[3 * d + 2] - df * df * w[3 * d + 1]; }
      else { const float v = a.gen_dist == DMV_GEN_BERN ? x : (x + 1.f) * 0.5f; lp = w[3 * d + 2] + v * w[3 * d]; }
      lf += lp;
      lo += lp * (M[d] ? 1.f : 0.f);
    }
    float nm = fmaxf(mo, lo);
    if (nm > -INFINITY) { so = so * expf(mo - nm) + expf(lo - nm); mo = nm; }
    nm = fmaxf(mf, lf);
    if (nm > -INFINITY) { sf = sf * expf(mf - nm) + expf(lf - nm); mf = nm; }
  }
  __shared__ float sh[4][256];
  sh[0][threadIdx.x] = mo; sh[1][threadIdx.x] = so; sh[2][threadIdx.x] = mf; sh[3][threadIdx.x] = sf;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      for (int q = 0; q < 2; ++q) {
        const float m1 = sh[2 * q][threadIdx.x], m2 = sh[2 * q][threadIdx.x + o];
        const float nm = fmaxf(m1, m2);
        if (nm > -INFINITY) {
          sh[2 * q + 1][threadIdx.x] = sh[2 * q + 1][threadIdx.x] * expf(m1 - nm) + sh[2 * q + 1][threadIdx.x + o] * expf(m2 - nm);
          sh[2 * q][threadIdx.x] = nm;
        }
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.out_obs[b] = sh[0][0] + logf(sh[1][0]);
    if (a.out_full) a.out_full[b] = sh[2][0] + logf(sh[3][0]);
  }
}

// ------------------------------------------------------------------------------------------
// Divergences between the model's posterior over the latent grid and a variational mixture
// (vgiwae/shared/vae_marginal_logprob.py:91-192).  For row r with data x_r (mask optional):
//   log p_z = log p(z) + sum_d [M_d] log p(x_d | z) - log_marg[r]        (posterior density on the grid)
//   log q_z = logsumexp_k sum_j log N(z_j; mu_rkj, sigma_rkj) - log Kq   (uniform mixture of Kq Gaussians)
//   KL(p||q) = sum_z p (log p - log q) dz,  KL(q||p) = sum_z q (log q - log p) dz,
//   JSD = (sum_z q (log q - log m) dz + sum_z p (log p - log m) dz) / 2,  log m = log((p + q) / 2).
// One block per row over the grid prepared by k_grid_prep; fp32 like the reference's tensors.
// ------------------------------------------------------------------------------------------
struct GridDivArgs {
  int64_t Nz;
  int D, Zd, R, Kq, gen_dist;
  const float* ws;        // [Nz][D][3]
  const float* log_prior; // [Nz]
  const float* Zg;        // [Nz][Zd]
  float dz;
  const float* X;         // [R][D]
  const uint8_t* M;       // nullable [R][D]
  const float* log_marg;  // [R]
  const float* q_mu;      // [R][Kq][Zd]
  const float* q_sigma;
  float* out;             // [3][R]
};

__global__ void k_grid_divs(GridDivArgs a) {
  const int r = blockIdx.x;
  const float* X = a.X + (int64_t)r * a.D;
  const uint8_t* M = a.M ? a.M + (int64_t)r * a.D : nullptr;
  const float* mu = a.q_mu + (int64_t)r * a.Kq * a.Zd;
  const float* sg = a.q_sigma + (int64_t)r * a.Kq * a.Zd;
  const bool normal = a.gen_dist <= DMV_GEN_NORMAL_EPS;
  const float lmarg = a.log_marg[r], logK = logf((float)a.Kq);
  float fow = 0.f, rev = 0.f, jq = 0.f, jp = 0.f;
  for (int64_t z = threadIdx.x; z < a.Nz; z += blockDim.x) {
    const float* w = a.ws + z * a.D * 3;
    float lp = a.log_prior[z];
    for (int d = 0; d < a.D; ++d) {
      float t;
      if (normal) { const float df = X[d] - w[3 * d]; t = w[3 * d + 2] - df * df * w[3 * d + 1]; }
      else { const float v = a.gen_dist == DMV_GEN_BERN ? X[d] : (X[d] + 1.f) * 0.5f; t = w[3 * d + 2] + v * w[3 * d]; }
      lp += (!M || M[d]) ? t : 0.f;
    }
    lp -= lmarg;
    const float* zz = a.Zg + z * a.Zd;
    float qm = -INFINITY, qs = 0.f;
    for (int k = 0; k < a.Kq; ++k) {
      float l = 0.f;
      for (int j = 0; j < a.Zd; ++j) l += log_normal(zz[j], mu[k * a.Zd + j], sg[k * a.Zd + j]);
      const float nm = fmaxf(qm, l);
      if (nm > -INFINITY) { qs = qs * expf(qm - nm) + expf(l - nm); qm = nm; }
    }
    const float lq = qm + logf(qs) - logK;
    const float hi = fmaxf(lp, lq), lo = fminf(lp, lq);
    const float lm = hi + log1pf(expf(lo - hi)) - 0.69314718056f;
    const float p = expf(lp), q = expf(lq);
    fow += p * (lp - lq) * a.dz;
    rev += q * (lq - lp) * a.dz;
    jq += q * (lq - lm) * a.dz;
    jp += p * (lp - lm) * a.dz;
  }
  __shared__ float sh[4][4];
  float v4[4] = {fow, rev, jq, jp};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float s = group_sum<64>(v4[i]);
    if ((threadIdx.x & 63) == 0) sh[i][threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4];
    for (int i = 0; i < 4; ++i) t[i] = sh[i][0] + sh[i][1] + sh[i][2] + sh[i][3];
    a.out[r] = t[0];
    a.out[a.R + r] = t[1];
    a.out[2 * a.R + r] = 0.5f * (t[2] + t[3]);
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int check_cfg(const DmvObjCfg* c) {
  if (!c) return DMV_E_BADARG;
  if (c->B < 1 || c->G < 1 || c->nz < 1 || c->I < 1 || c->D < 1 || c->Zd < 1) return DMV_E_BADARG;
  if (c->G > 64) return DMV_E_UNSUPPORTED;
  if (c->gen_dist < DMV_GEN_NORMAL || c->gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
  return DMV_OK;
}
static QAddr qaddr(const DmvObjCfg* c) {
  QAddr q;
  q.sb = c->enc_sb; q.sg = c->enc_sg; q.raw_off = c->raw_off; q.logit_off = c->logit_off;
  q.eps = c->q_eps ? DMV_EPSILON : 0.f;
  return q;
}
static inline int64_t nrows(const DmvObjCfg* c) { return (int64_t)c->nz * c->I * c->B * c->G; }
static inline unsigned grid_for(int64_t work, int block, unsigned cap = 8192) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}

// four consecutive latent dimensions per thread (Zd, ldz, offset multiples of 4, 16-byte aligned rows): one Philox call
// and one 16-byte access per array; same values as the scalar kernel
__global__ void k_latent_sample4(SampleArgs a) {
  const int BG = a.B * a.G;
  const int bg = blockIdx.x * blockDim.y + threadIdx.y;
  if (bg >= BG) return;
  const int64_t b = bg / a.G;
  const int g0 = bg - (int)b * a.G;
  const uint64_t off = a.offset + (a.offset_base ? a.offset_base[0] : 0ull);
  const int ld4 = (int)(a.ldz >> 2);
  for (int q = threadIdx.x; q < ld4; q += blockDim.x) {
    const int j = q << 2;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < a.Zd && !a.comp_idx) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { mu[k] = q_mu(a.enc, a.q, b, g0, j + k); sg[k] = q_sigma(a.enc, a.q, b, g0, j + k); }
    }
    for (int s = blockIdx.y; s < a.S; s += gridDim.y) {
      const int64_t n = (int64_t)s * BG + bg;
      const int64_t i = n * a.ldz + j;
      f32x4 z = {0.f, 0.f, 0.f, 0.f}, e = {0.f, 0.f, 0.f, 0.f};
      if (j < a.Zd) {
        if (a.comp_idx) {
          const int g = (int)a.comp_idx[n];
#pragma unroll
          for (int k = 0; k < 4; ++k) { mu[k] = q_mu(a.enc, a.q, b, g, j + k); sg[k] = q_sigma(a.enc, a.q, b, g, j + k); }
        }
        if (a.eps) {
          e = *(const f32x4*)(a.eps + i);
        } else {
          const dmv_f4 r = dmv_philox_normal4(a.seed, (off + (uint64_t)i) >> 2);
          e = f32x4{r.v[0], r.v[1], r.v[2], r.v[3]};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = mu[k] + sg[k] * e[k];
      }
      *(f32x4*)(a.Z + i) = z;
      if (a.eps_out) *(f32x4*)(a.eps_out + i) = e;
    }
  }
}

// threads along a row (power of two up to 64, at least `want`), rows per workgroup, grid of a row kernel
static inline void row_geometry(int want, int BG, int S, dim3& grid, dim3& block) {
  int bx = 1;
  while (bx < want && bx < 64) bx <<= 1;
  const int by = 256 / bx;
  block = dim3((unsigned)bx, (unsigned)by);
  grid = dim3((unsigned)((BG + by - 1) / by), (unsigned)(S > 65535 ? 65535 : S));   // the kernels stride over s
}

extern "C" int dmv_latent_sample(const DmvObjCfg* cfg, const float* enc, const float* eps, uint64_t seed,
                                 uint64_t offset, const uint64_t* offset_base, int64_t ldz, const int64_t* comp_idx,
                                 float* Z, float* eps_out, void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!enc || !Z || ldz < cfg->Zd) return DMV_E_BADARG;
  if (comp_idx && cfg->G != 1) return DMV_E_BADARG;
  SampleArgs a;
  a.N = nrows(cfg); a.B = cfg->B; a.G = cfg->G; a.Zd = cfg->Zd;
  a.q = qaddr(cfg); a.enc = enc; a.eps = eps; a.eps_out = eps_out; a.comp_idx = comp_idx;
  a.ldz = ldz; a.Z = Z; a.seed = seed; a.offset = offset; a.offset_base = offset_base;
  const uintptr_t al = (uintptr_t)Z | (uintptr_t)eps | (uintptr_t)eps_out;
  const int S = cfg->nz * cfg->I;
  if ((int64_t)cfg->B * cfg->G > 0x7fffffff) return DMV_E_UNSUPPORTED;
  a.S = S;
  dim3 grid, block;
  // a thread walks the samples of its (b, g, position): the grid's second dimension only has to bring the number of
  // workgroups to a few thousand when there are few datapoints and many samples
  auto slices = [&](unsigned gx) { const int64_t want = (4096 + gx - 1) / gx; return (unsigned)(want > S ? S : want); };
  if ((cfg->Zd & 3) == 0 && (ldz & 3) == 0 && (offset & 3) == 0 && (al & 15) == 0) {
    row_geometry((int)(ldz >> 2), cfg->B * cfg->G, S, grid, block);
    grid.y = slices(grid.x);
    hipLaunchKernelGGL(k_latent_sample4, grid, block, 0, (hipStream_t)stream, a);
  } else {
    row_geometry((int)ldz, cfg->B * cfg->G, S, grid, block);
    grid.y = slices(grid.x);
    hipLaunchKernelGGL(k_latent_sample, grid, block, 0, (hipStream_t)stream, a);
  }
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_obj_terms(const DmvObjCfg* cfg, int32_t q_kind, int32_t mix_C, const float* dec, int64_t ld_dec,
                             const float* X, const uint8_t* M, const float* Z, int64_t ldz, const float* enc,
                             float* terms, float* comp_lp, void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!dec || !X || !M || !Z || !enc || !terms) return DMV_E_BADARG;
  if (q_kind < DMV_Q_OWN || q_kind > DMV_Q_IMPUTATION) return DMV_E_UNSUPPORTED;
  if (q_kind != DMV_Q_OWN && (mix_C < 1 || mix_C > 64)) return DMV_E_UNSUPPORTED;
  const int P = cfg->gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * cfg->D : cfg->D;
  if (ld_dec < P || ldz < cfg->Zd) return DMV_E_BADARG;
  TermsArgs a;
  a.N = nrows(cfg); a.B = cfg->B; a.G = cfg->G; a.D = cfg->D; a.Zd = cfg->Zd;
  a.gen_dist = cfg->gen_dist; a.x_per_component = cfg->x_per_component;
  a.q_kind = q_kind; a.mix_C = mix_C; a.q = qaddr(cfg);
  a.dec = dec; a.ld_dec = ld_dec; a.X = X; a.M = M; a.Z = Z; a.ldz = ldz; a.enc = enc;
  a.terms = terms; a.comp_lp = comp_lp;
  a.vec4 = cfg->gen_dist >= DMV_GEN_BERN && (cfg->D & 3) == 0 && (ld_dec & 3) == 0 &&
           ((((uintptr_t)dec | (uintptr_t)X) & 15) == 0) && (((uintptr_t)M & 3) == 0);
  hipStream_t st = (hipStream_t)stream;
  const int span = cfg->D > cfg->Zd ? cfg->D : cfg->Zd;
  if (span <= 8) {
    hipLaunchKernelGGL(k_obj_terms<1>, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, st, a);
  } else if (span <= 96) {
    hipLaunchKernelGGL(k_obj_terms<8>, dim3((unsigned)((a.N * 8 + 255) / 256)), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL(k_obj_terms<64>, dim3((unsigned)((a.N * 64 + 255) / 256)), dim3(256), 0, st, a);
  }
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_obj_bound(const DmvObjCfg* cfg, int32_t kind, int32_t q_kind, int32_t use_logpi, int32_t dreg,
                             const float* enc, const float* terms, float* log_w, float* bound, float* mean_out,
                             float* wgt, float* zscale, float* dlogit, void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!terms || !bound) return DMV_E_BADARG;
  if (kind != DMV_BOUND_LSE && kind != DMV_BOUND_WEIGHTED) return DMV_E_UNSUPPORTED;
  if ((use_logpi || kind == DMV_BOUND_WEIGHTED) && !enc) return DMV_E_BADARG;
  BoundArgs a;
  a.B = cfg->B; a.G = cfg->G; a.nz = cfg->nz; a.I = cfg->I;
  a.kind = kind; a.q_term = (q_kind == DMV_Q_OWN) ? 3 : 4; a.use_logpi = use_logpi; a.dreg = dreg;
  a.q = qaddr(cfg); a.enc = enc; a.terms = terms; a.N = nrows(cfg);
  a.log_w = log_w; a.bound = bound; a.wgt = wgt; a.zscale = zscale; a.dlogit = dlogit;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_obj_bound, dim3((unsigned)((cfg->B + 3) / 4)), dim3(256), 0, st, a);
  DMV_HIP_CHECK_LAUNCH();
  if (mean_out) {
    hipLaunchKernelGGL(k_mean, dim3(1), dim3(256), 0, st, (const float*)bound, (int64_t)cfg->B, 1.f / (float)cfg->B, mean_out);
    DMV_HIP_CHECK_LAUNCH();
  }
  return DMV_OK;
}

extern "C" int dmv_obj_combine(const DmvObjCfg* cfg, const float* coef5, const float* terms, float* obj,
                               float* mean_out, void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!coef5 || !terms || !obj || cfg->I != 1) return DMV_E_BADARG;
  CombineArgs a;
  a.N = nrows(cfg); a.B = cfg->B; a.G = cfg->G; a.nz = cfg->nz;
  for (int t = 0; t < 5; ++t) a.c[t] = coef5[t];
  a.terms = terms; a.obj = obj;
  hipStream_t st = (hipStream_t)stream;
  const int64_t BG = (int64_t)cfg->B * cfg->G;
  hipLaunchKernelGGL(k_obj_combine, dim3((unsigned)((BG + 255) / 256)), dim3(256), 0, st, a);
  DMV_HIP_CHECK_LAUNCH();
  if (mean_out) {
    hipLaunchKernelGGL(k_mean, dim3(1), dim3(256), 0, st, (const float*)obj, BG, 1.f / (float)BG, mean_out);
    DMV_HIP_CHECK_LAUNCH();
  }
  return DMV_OK;
}

extern "C" size_t dmv_obj_implicit_ws_floats(const DmvObjCfg* cfg, int32_t mix_C) {
  if (check_cfg(cfg) != DMV_OK || mix_C < 1) return 0;
  return (size_t)cfg->B * mix_C * (cfg->Zd + 1);
}

extern "C" int dmv_obj_backward_latent_implicit(const DmvObjCfg* cfg, int32_t mix_C, const float* enc, const float* Z,
                                                int64_t ldz, const float* dZ, int64_t ld_dz, const float* wgt,
                                                float base_prior, float base_qmix, int32_t mix_live, float* ws, float* d_enc,
                                                void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!enc || !Z || !ws || !d_enc || ldz < cfg->Zd || (dZ && ld_dz < cfg->Zd)) return DMV_E_BADARG;
  if (cfg->G != 1) return DMV_E_BADARG;              // one ancestral sample per row
  if (mix_C < 1 || mix_C > 64) return DMV_E_UNSUPPORTED;
  BwdImplicitArgs a;
  a.B = cfg->B; a.S = cfg->nz * cfg->I; a.Zd = cfg->Zd; a.K = mix_C; a.q = qaddr(cfg);
  a.base_prior = base_prior; a.base_qmix = base_qmix; a.mix_live = mix_live != 0;
  a.enc = enc; a.Z = Z; a.ldz = ldz; a.dZ = dZ; a.ld_dz = ld_dz; a.wgt = wgt; a.ws = ws; a.d_enc = d_enc;
  hipLaunchKernelGGL(k_obj_bwd_latent_implicit, dim3((unsigned)((cfg->B + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_kl_std_normal(const DmvObjCfg* cfg, const float* enc, float* kl, void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!enc || !kl) return DMV_E_BADARG;
  KlArgs a; a.B = cfg->B; a.G = cfg->G; a.Zd = cfg->Zd; a.q = qaddr(cfg); a.enc = enc; a.kl = kl;
  const int64_t BG = (int64_t)cfg->B * cfg->G;
  hipLaunchKernelGGL(k_kl, dim3((unsigned)((BG + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

static int backward_dec(const DmvObjCfg* cfg, const float* dec, int64_t ld_dec, const float* X, const uint8_t* M,
                        const float* wgt, const float* wgt_f, const float* gb, const float* gscale, float base_m,
                        float base_f, float* d_dec, int64_t ld_ddec, void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!dec || !X || !M || !d_dec) return DMV_E_BADARG;
  const bool normal = cfg->gen_dist <= DMV_GEN_NORMAL_EPS;
  const int P = normal ? 2 * cfg->D : cfg->D;
  if (ld_dec < P || ld_ddec < P) return DMV_E_BADARG;
  if ((int64_t)cfg->B * cfg->G > 0x7fffffff || ld_ddec > 0x7fffffff) return DMV_E_UNSUPPORTED;
  BwdDecArgs a;
  a.N = nrows(cfg); a.B = cfg->B; a.G = cfg->G; a.D = cfg->D; a.S = cfg->nz * cfg->I;
  a.gen_dist = cfg->gen_dist; a.x_per_component = cfg->x_per_component;
  a.dec = dec; a.ld_dec = ld_dec; a.X = X; a.M = M; a.wgt = wgt; a.wgt_f = wgt_f; a.gb = gb; a.gscale = gscale;
  a.base_m = base_m; a.base_f = base_f; a.d_dec = d_dec; a.ld_ddec = ld_ddec;
  const bool aligned = (cfg->D & 3) == 0 && (ld_dec & 3) == 0 && (ld_ddec & 3) == 0 &&
                       ((((uintptr_t)dec | (uintptr_t)X | (uintptr_t)d_dec) & 15) == 0) && (((uintptr_t)M & 3) == 0);
  const int BG = cfg->B * cfg->G;
  dim3 grid, block;
  if (!normal && aligned && cfg->D >= 64) {
    const int W4 = (int)(ld_ddec / 4), bx = W4 > 64 ? 256 : 64;
    grid = dim3((unsigned)((W4 + bx - 1) / bx), (unsigned)(BG > 65535 ? 65535 : BG));
    hipLaunchKernelGGL(k_obj_bwd_dec_bern4, grid, dim3(bx), 0, (hipStream_t)stream, a);
  } else if (normal) {
    row_geometry(cfg->D, BG, a.S, grid, block);
    hipLaunchKernelGGL(k_obj_bwd_dec_normal, grid, block, 0, (hipStream_t)stream, a);
  } else {
    row_geometry((int)ld_ddec, BG, a.S, grid, block);
    hipLaunchKernelGGL(k_obj_bwd_dec, grid, block, 0, (hipStream_t)stream, a);
  }
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_obj_backward_dec(const DmvObjCfg* cfg, const float* dec, int64_t ld_dec, const float* X,
                                    const uint8_t* M, const float* wgt, const float* gb, const float* gscale, float base_m,
                                    float base_f, float* d_dec, int64_t ld_ddec, void* stream) {
  return backward_dec(cfg, dec, ld_dec, X, M, wgt, nullptr, gb, gscale, base_m, base_f, d_dec, ld_ddec, stream);
}

extern "C" int dmv_obj_backward_dec_terms(const DmvObjCfg* cfg, const float* dec, int64_t ld_dec, const float* X,
                                          const uint8_t* M, const float* g_masked, const float* g_full, float* d_dec,
                                          int64_t ld_ddec, void* stream) {
  if (!g_masked || !g_full) return DMV_E_BADARG;
  return backward_dec(cfg, dec, ld_dec, X, M, g_masked, g_full, nullptr, nullptr, 1.f, 1.f, d_dec, ld_ddec, stream);
}

static int backward_latent(const DmvObjCfg* cfg, const DmvLatentBwd* p, const float* enc, const float* Z,
                           const float* eps, int64_t ldz, const float* dZ, int64_t ld_dz, const float* wgt,
                           const float* wgt_qown, const float* wgt_qmix, const float* gb, const float* zscale,
                           const float* comp_lp, const float* qmix, const float* dlogit, const float* kl_wgt,
                           const float* gscale, float* d_enc, void* stream) {
  int rc = check_cfg(cfg);
  if (rc) return rc;
  if (!p || !enc || !Z || !eps || !d_enc || ldz < cfg->Zd) return DMV_E_BADARG;
  if (dZ && ld_dz < cfg->Zd) return DMV_E_BADARG;
  if (p->base_qmix != 0.f && (!comp_lp || !qmix)) return DMV_E_BADARG;
  BwdLatArgs a;
  a.B = cfg->B; a.G = cfg->G; a.S = cfg->nz * cfg->I; a.Zd = cfg->Zd; a.N = nrows(cfg);
  a.q = qaddr(cfg); a.mix_kind = p->mix_kind;
  a.base_prior = p->base_prior; a.base_qown = p->base_qown; a.base_qmix = p->base_qmix; a.base_w = p->base_w;
  a.own_live = p->own_live; a.mix_live = p->mix_live; a.use_zscale = p->use_zscale; a.kl_coef = p->kl_coef;
  a.enc = enc; a.Z = Z; a.eps = eps; a.ldz = ldz; a.dZ = dZ; a.ld_dz = ld_dz;
  a.wgt = wgt; a.wgt_qown = wgt_qown; a.wgt_qmix = wgt_qmix;
  a.zscale = zscale; a.comp_lp = comp_lp; a.qmix = qmix; a.dlogit = dlogit; a.kl_wgt = kl_wgt; a.gscale = gscale;
  a.d_enc = d_enc; a.gb = gb;
  const int64_t work = (int64_t)cfg->B * cfg->G * (cfg->Zd + 1);
  if (work > 0x7fffffff || cfg->enc_sb > 0x7fffffff) return DMV_E_UNSUPPORTED;
  // which entries of a datapoint's slice of d_enc carry gradients
  if (cfg->enc_sg >= cfg->raw_off + cfg->Zd) {      // one [mu | raw sigma | padding] block per component (imputations)
    a.zf_nseg = cfg->G; a.zf_len = (int)cfg->enc_sg; a.zf_used = (int)cfg->raw_off + cfg->Zd;
  } else {                                          // one block per datapoint: [mu.. | raw.. | (logits) | padding]
    a.zf_nseg = 1; a.zf_len = (int)cfg->enc_sb;
    a.zf_used = p->mix_kind == DMV_Q_STRATIFIED ? (int)cfg->logit_off + cfg->G : (int)cfg->raw_off + cfg->G * cfg->Zd;
  }
  if (a.zf_used > a.zf_len) return DMV_E_BADARG;
  const int per_b = cfg->G * cfg->Zd;
  if (p->base_qmix != 0.f && per_b <= 1024) {
    // datapoints per workgroup: about 256 threads, at most 48 KB of LDS (one datapoint needs at most 29 KB)
    const size_t dp_bytes = (size_t)(3 * per_b + 2 * cfg->G + cfg->G * cfg->G) * sizeof(float);
    int nb = per_b >= 256 ? 1 : 256 / per_b;
    if ((size_t)nb * dp_bytes > 49152) nb = (int)(49152 / dp_bytes);
    if (nb < 1) nb = 1;
    if (nb > cfg->B) nb = cfg->B;
    const unsigned threads = (unsigned)((nb * per_b + 63) / 64 * 64);
    const unsigned blocks = (unsigned)((cfg->B + nb - 1) / nb);
    if (p->mix_kind == DMV_Q_STRATIFIED)
      hipLaunchKernelGGL(k_obj_bwd_latent_mix<true>, dim3(blocks), dim3(threads), nb * dp_bytes, (hipStream_t)stream, a, nb, per_b);
    else
      hipLaunchKernelGGL(k_obj_bwd_latent_mix<false>, dim3(blocks), dim3(threads), nb * dp_bytes, (hipStream_t)stream, a, nb, per_b);
    DMV_HIP_CHECK_LAUNCH();
    return DMV_OK;
  }
  hipLaunchKernelGGL(k_obj_bwd_latent, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_obj_backward_latent(const DmvObjCfg* cfg, const DmvLatentBwd* p, const float* enc, const float* Z,
                                       const float* eps, int64_t ldz, const float* dZ, int64_t ld_dz, const float* wgt,
                                       const float* gb, const float* zscale, const float* comp_lp, const float* qmix,
                                       const float* dlogit, const float* kl_wgt, const float* gscale, float* d_enc,
                                       void* stream) {
  return backward_latent(cfg, p, enc, Z, eps, ldz, dZ, ld_dz, wgt, nullptr, nullptr, gb, zscale, comp_lp, qmix, dlogit, kl_wgt,
                         gscale, d_enc, stream);
}

extern "C" int dmv_obj_backward_latent_terms(const DmvObjCfg* cfg, int32_t mix_kind, int32_t q_live, const float* enc,
                                             const float* Z, const float* eps, int64_t ldz, const float* dZ, int64_t ld_dz,
                                             const float* g_prior, const float* g_qown, const float* g_qmix,
                                             const float* zscale, const float* comp_lp, const float* qmix, float* d_enc,
                                             void* stream) {
  if (!g_prior || !g_qown || !g_qmix || !comp_lp || !qmix) return DMV_E_BADARG;
  if (mix_kind != DMV_Q_IMPUTATION && mix_kind != DMV_Q_STRATIFIED) return DMV_E_BADARG;
  DmvLatentBwd p;
  p.base_prior = 1.f; p.base_qown = 1.f; p.base_qmix = 1.f; p.base_w = 0.f; p.kl_coef = 0.f;
  p.own_live = q_live; p.mix_live = q_live; p.use_zscale = zscale ? 1 : 0; p.mix_kind = mix_kind;
  return backward_latent(cfg, &p, enc, Z, eps, ldz, dZ, ld_dz, g_prior, g_qown, g_qmix, nullptr, zscale, comp_lp, qmix, nullptr,
                         nullptr, nullptr, d_enc, stream);
}

extern "C" int dmv_lse_update(int32_t B, int64_t Kc, const float* terms, int32_t q_term, float* run_max, float* run_sum,
                              void* stream) {
  if (B < 1 || Kc < 1 || !terms || !run_max || !run_sum || (q_term != 3 && q_term != 4)) return DMV_E_BADARG;
  LseArgs a;
  const int64_t N = Kc * B;
  a.B = B; a.Kc = Kc; a.gm = terms; a.pr = terms + 2 * N; a.q = terms + (int64_t)q_term * N;
  a.run_max = run_max; a.run_sum = run_sum;
  hipLaunchKernelGGL(k_lse_update, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" size_t dmv_grid_marginal_ws_floats(int64_t Nz, int32_t D) {
  return (Nz < 0 || D < 1) ? 0 : (size_t)Nz * D * 3;
}

extern "C" int dmv_grid_marginal(int32_t gen_dist, int32_t D, int64_t Nz, const float* dec, int64_t ld_dec,
                                 const float* logw, int32_t B, const float* X, const uint8_t* M, float* ws,
                                 float* out_obs, float* out_full, void* stream) {
  if (gen_dist < DMV_GEN_NORMAL || gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
  if (D < 1 || Nz < 1 || B < 0 || !dec || !logw || !X || !M || !ws || !out_obs) return DMV_E_BADARG;
  if (ld_dec < (gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * D : D)) return DMV_E_BADARG;
  if (B == 0) return DMV_OK;
  GridArgs a;
  a.Nz = Nz; a.D = D; a.B = B; a.gen_dist = gen_dist; a.dec = dec; a.ld_dec = ld_dec; a.logw = logw; a.X = X; a.M = M;
  a.ws = ws; a.out_obs = out_obs; a.out_full = out_full;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_grid_prep, dim3(grid_for(Nz * D, 256)), dim3(256), 0, st, a);
  DMV_HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_grid_marginal, dim3((unsigned)B), dim3(256), 0, st, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_grid_divergences(int32_t gen_dist, int32_t D, int32_t Zd, int64_t Nz, const float* ws,
                                    const float* log_prior, const float* Zgrid, float dz, int32_t R, const float* X,
                                    const uint8_t* M, const float* log_marg, int32_t Kq, const float* q_mu,
                                    const float* q_sigma, float* out, void* stream) {
  if (gen_dist < DMV_GEN_NORMAL || gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
  if (D < 1 || Zd < 1 || Nz < 1 || R < 0 || Kq < 1 || !(dz > 0.f)) return DMV_E_BADARG;
  if (!ws || !log_prior || !Zgrid || !X || !log_marg || !q_mu || !q_sigma || !out) return DMV_E_BADARG;
  if (R == 0) return DMV_OK;
  GridDivArgs a;
  a.Nz = Nz; a.D = D; a.Zd = Zd; a.R = R; a.Kq = Kq; a.gen_dist = gen_dist; a.ws = ws; a.log_prior = log_prior;
  a.Zg = Zgrid; a.dz = dz; a.X = X; a.M = M; a.log_marg = log_marg; a.q_mu = q_mu; a.q_sigma = q_sigma; a.out = out;
  hipLaunchKernelGGL(k_grid_divs, dim3((unsigned)R), dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}
