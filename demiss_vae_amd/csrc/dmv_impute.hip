// DeMissVAE imputation update: device-resident imputation buffer (gather / scatter by imputation
// index), the pseudo-Gibbs / LAIR sampling tail (draw x_m ~ p(.|z), clamp, non-finite fallback, blend
// with the observed values, write back) and the per-datapoint categorical resampling of LAIR, with
// on-device Philox noise or caller-provided noise.  All HBM-bound row kernels.
#include "dmv_common.h"
#include "dmv_rng.h"
#include "dmv_tail.h"

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

// 16-byte variants (D % 4 == 0, aligned bases): blockIdx.y walks the rows, threads walk a row's float4 columns,
// so no per-element division and every access is a full-width coalesced vector
template <bool GATHER>
__global__ void k_rows_move4(float* __restrict__ compact, const int64_t* __restrict__ idx, int64_t n, int D4,
                             float* __restrict__ table) {
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    const int64_t t = idx[r];
    f32x4* c = (f32x4*)compact + r * D4;
    f32x4* g = (f32x4*)table + t * D4;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < D4; q += gridDim.x * blockDim.x) {
      if (GATHER) c[q] = g[q];
      else g[q] = c[q];
    }
  }
}

// ------------------------------------------------------------------------------------------
// sampling tail (vgiwae/shared/vae_imputation.py:38-52 pseudo-Gibbs; :526-561, :657-666 LAIR)
// ------------------------------------------------------------------------------------------
__global__ void k_impute_tail(TailArgs a) {
  const int64_t total = a.N * a.D;
  const uint64_t off = a.offset + (a.offset_base ? a.offset_base[0] : 0ull);
  const bool normal = a.gen_dist <= DMV_GEN_NORMAL_EPS;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / a.D;
    const int d = (int)(i - n * a.D);
    const int64_t r = a.row_map ? a.row_map[n] : n;
    const float* dec = a.dec + r * a.ld_dec;
    const float noise = a.noise ? a.noise[i]
                                : normal ? dmv_philox_normal(a.seed, off + (uint64_t)i)
                                         : dmv_philox_uniform(a.seed, off + (uint64_t)i);
    const float out = tail_element(a, dec, d, noise, a.X_old[i], a.M[i] != 0);
    a.X_out[i] = out;
    if (a.buf) a.buf[a.idx_imp[n] * a.D + d] = out;
  }
}

// Four consecutive columns per thread (D % 4 == 0, 16-byte aligned rows, offset % 4 == 0): one Philox call, one
// 16-byte access per array; blockIdx.y walks the rows.  Same values as the scalar kernel.
__global__ void k_impute_tail4(TailArgs a) {
  const bool normal = a.gen_dist <= DMV_GEN_NORMAL_EPS;
  const uint64_t off = a.offset + (a.offset_base ? a.offset_base[0] : 0ull);
  const int D4 = a.D >> 2;
  for (int64_t n = blockIdx.y; n < a.N; n += gridDim.y) {
    const int64_t r = a.row_map ? a.row_map[n] : n;
    const float* dec = a.dec + r * a.ld_dec;
    float* brow = a.buf ? a.buf + a.idx_imp[n] * a.D : nullptr;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < D4; q += gridDim.x * blockDim.x) {
      const int64_t i = n * a.D + 4 * q;
      const f32x4 xo = *(const f32x4*)(a.X_old + i);
      const uint32_t m4 = *(const uint32_t*)(a.M + i);
      dmv_f4 nz;
      if (a.noise) {
        const f32x4 t = *(const f32x4*)(a.noise + i);
        nz = dmv_f4{{t[0], t[1], t[2], t[3]}};
      } else {
        const uint64_t c = (off + (uint64_t)i) >> 2;
        nz = normal ? dmv_philox_normal4(a.seed, c) : dmv_philox_uniform4(a.seed, c);
      }
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = tail_element(a, dec, 4 * q + e, nz.v[e], xo[e], ((m4 >> (8 * e)) & 0xffu) != 0);
      *(f32x4*)(a.X_out + i) = o;
      if (brow) *(f32x4*)(brow + 4 * q) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------
// categorical draws per datapoint (vgiwae/shared/vae_imputation.py:181-186, :679-680):
// idx[b][k] = smallest p with cumsum(softmax(logits[b]))[p] > u[b][k]; one wavefront per datapoint
// ------------------------------------------------------------------------------------------
struct DrawArgs {
  int B, P, K;
  int inner;            // logit p of datapoint b sits at logits[(p / inner) * group_stride + b * inner + p % inner]:
  int64_t group_stride; //   inner == P: plain [B][P] rows; inner < P: a [P / inner][B][inner] array (proposals (t i) b k)
  int64_t* rows;        // nullable [B][rows_keep]: that position itself for the first rows_keep draws of a datapoint
  int rows_keep;        //   (= the row of the picked proposal in arrays laid out like the logits)
  const float* logits;
  const float* u;       // nullable [B][K] in [0,1)
  uint64_t seed, offset;
  const uint64_t* offset_base;  // nullable device counter added to offset
  int64_t* idx;         // [B][K]
};

// The unnormalised probabilities exp(logit - max) are computed once per datapoint and staged in LDS (up to kDrawStage
// of them; longer rows fall back to recomputing from global memory); every draw then walks the same running sum in the
// same order as before, from LDS.
constexpr int kDrawStage = 1024;

__global__ void k_categorical_draw(DrawArgs a) {
  __shared__ float stage[4][kDrawStage];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x * (blockDim.x >> 6) + wv;
  if (b >= a.B) return;
  const float* lw0 = a.logits + (int64_t)b * a.inner;
  auto pos = [&](int p) { const int grp = p / a.inner; return (int64_t)grp * a.group_stride + (p - grp * a.inner); };
  const bool staged = a.P <= kDrawStage;
  float mx = -INFINITY;
  for (int p = lane; p < a.P; p += 64) mx = fmaxf(mx, lw0[pos(p)]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float se = 0.f;
  for (int p = lane; p < a.P; p += 64) {
    const float e = expf(lw0[pos(p)] - mx);
    if (staged) stage[wv][p] = e;
    se += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
  __builtin_amdgcn_wave_barrier();      // one wave writes and reads its own stage row: LDS order within the wave
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the stores have landed before other lanes read them
  for (int k = lane; k < a.K; k += 64) {
    const float u = a.u ? a.u[(int64_t)b * a.K + k]
                        : dmv_philox_uniform(a.seed, a.offset + (a.offset_base ? a.offset_base[0] : 0ull) + (uint64_t)b * a.K + k);
    const float target = u * se;
    float c = 0.f;
    int pick = a.P - 1;
    if (staged) {
      for (int p = 0; p < a.P; ++p) {
        c += stage[wv][p];
        if (c > target) { pick = p; break; }
      }
    } else {
      for (int p = 0; p < a.P; ++p) {
        c += expf(lw0[pos(p)] - mx);
        if (c > target) { pick = p; break; }
      }
    }
    if (a.idx) a.idx[(int64_t)b * a.K + k] = pick;
    if (a.rows && k < a.rows_keep) a.rows[(int64_t)b * a.rows_keep + k] = (int64_t)b * a.inner + pos(pick);
  }
}

// LAIR's resampling and sampling tail in one launch (vgiwae/shared/vae_imputation.py:181-198, 521-561): a wavefront per
// datapoint draws its K + r proposal numbers exactly as k_categorical_draw does (same uniforms, same running sum), keeps
// the rows of the first K in LDS, and then produces the datapoint's K new imputations exactly as k_impute_tail does from
// those rows (same noise counters, same arithmetic): what the two launches compute, without the second launch and the
// round trip of the row map.
__global__ void k_lair_resample_tail(DrawArgs a, TailArgs t, int Kkeep) {
  __shared__ float stage[4][kDrawStage];
  __shared__ int64_t srow[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x * (blockDim.x >> 6) + wv;
  if (b >= a.B) return;
  const float* lw0 = a.logits + (int64_t)b * a.inner;
  auto pos = [&](int p) { const int grp = p / a.inner; return (int64_t)grp * a.group_stride + (p - grp * a.inner); };
  float mx = -INFINITY;
  for (int p = lane; p < a.P; p += 64) mx = fmaxf(mx, lw0[pos(p)]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float se = 0.f;
  for (int p = lane; p < a.P; p += 64) {
    const float e = expf(lw0[pos(p)] - mx);
    stage[wv][p] = e;
    se += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane < a.K) {
    const int k = lane;
    const float u = a.u ? a.u[(int64_t)b * a.K + k]
                        : dmv_philox_uniform(a.seed, a.offset + (a.offset_base ? a.offset_base[0] : 0ull) + (uint64_t)b * a.K + k);
    const float target = u * se;
    float c = 0.f;
    int pick = a.P - 1;
    for (int p = 0; p < a.P; ++p) {
      c += stage[wv][p];
      if (c > target) { pick = p; break; }
    }
    if (a.idx) a.idx[(int64_t)b * a.K + k] = pick;
    if (k < Kkeep) {
      const int64_t row = (int64_t)b * a.inner + pos(pick);
      srow[wv][k] = row;
      if (a.rows) a.rows[(int64_t)b * Kkeep + k] = row;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
  // ---- the datapoint's K rows of new imputations
  const uint64_t off = t.offset + (t.offset_base ? t.offset_base[0] : 0ull);
  const bool normal = t.gen_dist <= DMV_GEN_NORMAL_EPS;
  for (int e = lane; e < Kkeep * t.D; e += 64) {
    const int k = e / t.D, d = e - k * t.D;
    const int64_t n = (int64_t)b * Kkeep + k;
    const int64_t i = n * t.D + d;
    const float* dec = t.dec + srow[wv][k] * t.ld_dec;
    const float noise = t.noise ? t.noise[i]
                                : normal ? dmv_philox_normal(t.seed, off + (uint64_t)i)
                                         : dmv_philox_uniform(t.seed, off + (uint64_t)i);
    const float out = tail_element(t, dec, d, noise, t.X_old[i], t.M[i] != 0);
    t.X_out[i] = out;
    if (t.buf) t.buf[t.idx_imp[n] * t.D + d] = out;
  }
}

// Bernoulli(1 - p) keep masks for the residual blocks' dropout (vgiwae/shared/neural_nets.py:65,70)
__global__ void k_dropout_masks(uint8_t* __restrict__ keep, int64_t n, float p, uint64_t seed, uint64_t offset,
                                const uint64_t* offset_base) {
  if (offset_base) offset += offset_base[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    keep[i] = dmv_philox_uniform(seed, offset + (uint64_t)i) >= p ? 1 : 0;
}
// four masks per thread and Philox call (n % 4 == 0, offset % 4 == 0, 4-byte aligned base); same values
__global__ void k_dropout_masks4(uint32_t* __restrict__ keep4, int64_t n4, float p, uint64_t seed, uint64_t offset,
                                 const uint64_t* offset_base) {
  if (offset_base) offset += offset_base[0];
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (int64_t)gridDim.x * blockDim.x) {
    const dmv_f4 u = dmv_philox_uniform4(seed, (offset >> 2) + (uint64_t)q);
    keep4[q] = (u.v[0] >= p ? 1u : 0u) | (u.v[1] >= p ? 0x100u : 0u) | (u.v[2] >= p ? 0x10000u : 0u) | (u.v[3] >= p ? 0x1000000u : 0u);
  }
}

// ------------------------------------------------------------------------------------------
static inline unsigned grid_for(int64_t work, int block, unsigned cap = 8192) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}

// Rows `sel[i]` of up to four arrays at once (the arrays of a data split: data, mask, targets, indices), each array
// with its own row length in bytes: what a batch of the device loader is.  One thread moves one 4-byte word (rows
// whose length and addresses are multiples of 4) or one byte.
struct GatherMulti {
  int n_arrays;
  const uint8_t* src[4];
  uint8_t* dst[4];
  int64_t row_bytes[4];
  int64_t first[5];       // prefix sums of the arrays' units (words or bytes) per row
  int word[4];            // 1: the array moves as 4-byte words
};

__global__ void k_rows_gather_multi(GatherMulti g, const int64_t* __restrict__ sel, int64_t n_rows) {
  const int64_t per_row = g.first[g.n_arrays];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows * per_row; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / per_row, u = i - row * per_row;
    int k = 0;
    while (k + 1 < g.n_arrays && u >= g.first[k + 1]) ++k;
    const int64_t off = u - g.first[k], from = sel[row];
    if (g.word[k])
      ((uint32_t*)(g.dst[k] + row * g.row_bytes[k]))[off] = ((const uint32_t*)(g.src[k] + from * g.row_bytes[k]))[off];
    else
      g.dst[k][row * g.row_bytes[k] + off] = g.src[k][from * g.row_bytes[k] + off];
  }
}

extern "C" int dmv_rows_gather_multi(int32_t n_arrays, const void* const* src, void* const* dst, const int64_t* row_bytes,
                                     const int64_t* sel, int64_t n_rows, void* stream) {
  if (n_arrays < 1 || n_arrays > 4 || !src || !dst || !row_bytes || !sel || n_rows < 0) return DMV_E_BADARG;
  if (n_rows == 0) return DMV_OK;
  GatherMulti g;
  g.n_arrays = n_arrays;
  g.first[0] = 0;
  for (int k = 0; k < 4; ++k) {
    const bool live = k < n_arrays;
    if (live && (!src[k] || !dst[k] || row_bytes[k] < 1)) return DMV_E_BADARG;
    g.src[k] = live ? (const uint8_t*)src[k] : nullptr;
    g.dst[k] = live ? (uint8_t*)dst[k] : nullptr;
    g.row_bytes[k] = live ? row_bytes[k] : 0;
    g.word[k] = live && (row_bytes[k] & 3) == 0 && (((uintptr_t)src[k] | (uintptr_t)dst[k]) & 3) == 0;
    g.first[k + 1] = g.first[k] + (live ? (g.word[k] ? row_bytes[k] / 4 : row_bytes[k]) : 0);
  }
  hipLaunchKernelGGL(k_rows_gather_multi, dim3(grid_for(n_rows * g.first[n_arrays], 256)), dim3(256), 0, (hipStream_t)stream, g, sel,
                     n_rows);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_rows_gather(const float* src, const int64_t* idx, int64_t n, int32_t D, float* dst, void* stream) {
  if (!src || !idx || !dst || n < 0 || D < 1) return DMV_E_BADARG;
  if (n == 0) return DMV_OK;
  if ((D & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const int D4 = D / 4, bx = D4 > 64 ? 256 : (D4 > 32 ? 64 : 32);   // threads per row
    dim3 grid((unsigned)((D4 + bx - 1) / bx), (unsigned)(n > 65535 ? 65535 : n));
    hipLaunchKernelGGL(k_rows_move4<true>, grid, dim3(bx), 0, (hipStream_t)stream, dst, idx, n, D4, const_cast<float*>(src));
  } else {
    hipLaunchKernelGGL(k_rows_gather, dim3(grid_for(n * D, 256)), dim3(256), 0, (hipStream_t)stream, src, idx, n, (int)D, dst);
  }
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_rows_scatter(float* dst, const int64_t* idx, int64_t n, int32_t D, const float* src, void* stream) {
  if (!src || !idx || !dst || n < 0 || D < 1) return DMV_E_BADARG;
  if (n == 0) return DMV_OK;
  if ((D & 3) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const int D4 = D / 4, bx = D4 > 64 ? 256 : (D4 > 32 ? 64 : 32);
    dim3 grid((unsigned)((D4 + bx - 1) / bx), (unsigned)(n > 65535 ? 65535 : n));
    hipLaunchKernelGGL(k_rows_move4<false>, grid, dim3(bx), 0, (hipStream_t)stream, const_cast<float*>(src), idx, n, D4, dst);
  } else {
    hipLaunchKernelGGL(k_rows_scatter, dim3(grid_for(n * D, 256)), dim3(256), 0, (hipStream_t)stream, dst, idx, n, (int)D, src);
  }
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_impute_tail(int64_t N, int32_t D, int32_t gen_dist, const float* dec, int64_t ld_dec,
                               const int64_t* row_map, const float* X_old, const uint8_t* M, const float* noise,
                               uint64_t seed, uint64_t offset, const uint64_t* offset_base, const float* clip_min,
                               const float* clip_max, int32_t clip_after_blend, float* X_out, float* buf,
                               const int64_t* idx_imp, void* stream) {
  if (N < 0 || D < 1 || !dec || !X_old || !M || !X_out) return DMV_E_BADARG;
  if (gen_dist < DMV_GEN_NORMAL || gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
  const int P = gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * D : D;
  if (ld_dec < P || (buf && !idx_imp)) return DMV_E_BADARG;
  if (N == 0) return DMV_OK;
  TailArgs a;
  a.N = N; a.D = D; a.gen_dist = gen_dist; a.dec = dec; a.ld_dec = ld_dec; a.row_map = row_map;
  a.X_old = X_old; a.M = M; a.noise = noise; a.seed = seed; a.offset = offset; a.offset_base = offset_base;
  a.clip_min = clip_min; a.clip_max = clip_max; a.clip_after_blend = clip_after_blend;
  a.X_out = X_out; a.buf = buf; a.idx_imp = idx_imp;
  const uintptr_t al = (uintptr_t)dec | (uintptr_t)X_old | (uintptr_t)X_out | (uintptr_t)noise | (uintptr_t)buf;
  if (D >= 64 && (D & 3) == 0 && (ld_dec & 3) == 0 && (al & 15) == 0 && ((uintptr_t)M & 3) == 0 && (offset & 3) == 0) {
    const int D4 = D / 4, bx = D4 > 64 ? 256 : 64;
    dim3 grid((unsigned)((D4 + bx - 1) / bx), (unsigned)(N > 65535 ? 65535 : N));
    hipLaunchKernelGGL(k_impute_tail4, grid, dim3(bx), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(k_impute_tail, dim3(grid_for(N * D, 256)), dim3(256), 0, (hipStream_t)stream, a);
  }
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_categorical_draw(int32_t B, int32_t P, int32_t K, const float* logits, const float* u, uint64_t seed,
                                    uint64_t offset, const uint64_t* offset_base, int64_t* idx, void* stream) {
  if (B < 1 || P < 1 || K < 1 || !logits || !idx) return DMV_E_BADARG;
  DrawArgs a;
  a.B = B; a.P = P; a.K = K; a.logits = logits; a.u = u; a.seed = seed; a.offset = offset; a.offset_base = offset_base; a.idx = idx;
  a.inner = P; a.group_stride = 0; a.rows = nullptr; a.rows_keep = 0;
  hipLaunchKernelGGL(k_categorical_draw, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_categorical_draw_grouped(int32_t B, int32_t groups, int32_t inner, int32_t K, const float* logits,
                                            const float* u, uint64_t seed, uint64_t offset, const uint64_t* offset_base,
                                            int64_t* idx, int64_t* rows, int32_t rows_keep, void* stream) {
  if (B < 1 || groups < 1 || inner < 1 || K < 1 || !logits || (!idx && !rows)) return DMV_E_BADARG;
  if (rows && (rows_keep < 1 || rows_keep > K)) return DMV_E_BADARG;
  if ((int64_t)groups * inner > 0x7fffffff) return DMV_E_UNSUPPORTED;
  DrawArgs a;
  a.B = B; a.P = groups * inner; a.K = K; a.logits = logits; a.u = u; a.seed = seed; a.offset = offset;
  a.offset_base = offset_base; a.idx = idx;
  a.inner = inner; a.group_stride = (int64_t)B * inner; a.rows = rows; a.rows_keep = rows ? rows_keep : 0;
  hipLaunchKernelGGL(k_categorical_draw, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_lair_resample_tail(int32_t B, int32_t groups, int32_t inner, int32_t K_draws, int32_t K, const float* logits,
                                      const float* u, uint64_t draw_seed, uint64_t draw_offset, int32_t D, int32_t gen_dist,
                                      const float* dec, int64_t ld_dec, const float* X_old, const uint8_t* M, const float* noise,
                                      uint64_t tail_seed, uint64_t tail_offset, const uint64_t* offset_base,
                                      const float* clip_min, const float* clip_max, float* X_out, float* buf,
                                      const int64_t* idx_imp, int64_t* rows, void* stream) {
  if (B < 1 || groups < 1 || inner < 1 || K < 1 || K_draws < K || D < 1 || !logits || !dec || !X_old || !M || !X_out)
    return DMV_E_BADARG;
  if (gen_dist < DMV_GEN_NORMAL || gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
  // the proposals' probabilities are staged in LDS and the draws sit on the lanes of one wavefront
  if ((int64_t)groups * inner > kDrawStage || K_draws > 64) return DMV_E_UNSUPPORTED;
  const int P = gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * D : D;
  if (ld_dec < P || (buf && !idx_imp)) return DMV_E_BADARG;
  DrawArgs a;
  a.B = B; a.P = groups * inner; a.K = K_draws; a.logits = logits; a.u = u; a.seed = draw_seed; a.offset = draw_offset;
  a.offset_base = offset_base; a.idx = nullptr;
  a.inner = inner; a.group_stride = (int64_t)B * inner; a.rows = rows; a.rows_keep = K;
  TailArgs t;
  t.N = (int64_t)B * K; t.D = D; t.gen_dist = gen_dist; t.dec = dec; t.ld_dec = ld_dec; t.row_map = nullptr;
  t.X_old = X_old; t.M = M; t.noise = noise; t.seed = tail_seed; t.offset = tail_offset; t.offset_base = offset_base;
  t.clip_min = clip_min; t.clip_max = clip_max; t.clip_after_blend = 1;
  t.X_out = X_out; t.buf = buf; t.idx_imp = idx_imp;
  hipLaunchKernelGGL(k_lair_resample_tail, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, t, K);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

// rows of the proposal table picked by the resampling draws: proposal v = i Kp + k of datapoint b sits in row
// (i B + b) Kp + k of the [(i b k)] decoder outputs; the first K of the Kp draws are kept
__global__ void k_lair_row_map(int32_t B, int32_t K, int32_t Kp, const int64_t* __restrict__ draws, int64_t* __restrict__ out) {
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < (int64_t)B * K; n += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = n / K, j = n - b * K;
    const int64_t v = draws[b * Kp + j];
    out[n] = ((v / Kp) * B + b) * Kp + v % Kp;
  }
}

extern "C" int dmv_lair_row_map(int32_t B, int32_t K, int32_t Kp, const int64_t* draws, int64_t* row_map, void* stream) {
  if (B < 1 || K < 1 || Kp < K || !draws || !row_map) return DMV_E_BADARG;
  hipLaunchKernelGGL(k_lair_row_map, dim3(grid_for((int64_t)B * K, 256)), dim3(256), 0, (hipStream_t)stream, B, K, Kp, draws,
                     row_map);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_dropout_masks(uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset,
                                 const uint64_t* offset_base, void* stream) {
  if (!keep || n < 0 || !(p >= 0.f && p < 1.f)) return DMV_E_BADARG;
  if (n == 0) return DMV_OK;
  if ((n & 3) == 0 && (offset & 3) == 0 && ((uintptr_t)keep & 3) == 0)
    hipLaunchKernelGGL(k_dropout_masks4, dim3(grid_for(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, (uint32_t*)keep, n / 4, p,
                       seed, offset, offset_base);
  else
    hipLaunchKernelGGL(k_dropout_masks, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, keep, n, p, seed, offset,
                       offset_base);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}
