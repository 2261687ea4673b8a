// Residual fully-connected network on fp32 MFMA (gfx950): weight packing, fused forward,
// fused input-gradient chain, weight-gradient GEMMs.
//
// Replaces vgiwae/shared/neural_nets.py:67-72 (ResidualBlock.forward) and :127-133
// (ResidualFCNetwork.forward) plus their autograd backward.
//
// Execution model.  One wavefront owns 32 batch rows for the WHOLE network.  A layer is computed
// transposed, out^T[n][m] = sum_k W[n][k] act[m][k], with v_mfma_f32_32x32x2_f32: the weight
// fragment is the A operand (lane = output feature), the activation is the B operand
// (lane = batch row).  The accumulator layout of that product (lane (m, h = lane>>5), register r
// <-> feature 32 t + 8 (r>>2) + 4 h + (r&3)) is exactly the B-operand layout the next layer needs
// when the reduction index is visited in the order k = 8 s + 4 h + e (group s, step e): register
// 4 (s&3) + e of tile s>>2.  So activations never leave the register file between layers: no LDS,
// no barriers, weights stream from L2 in fragment order (dmv_mlp_pack), 1 KiB per wave load.
// fp32 MFMA runs at the fp32 vector rate (64 FLOP/clk/SIMD), so this path is bound by MFMA issue;
// one wave per SIMD with two 32*NT-register activation sets keeps the matrix pipe busy.
#include "dmv_common.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------
struct PackArgs {
  DmvLayout L;
  const float* w[DMV_MAX_LAYERS];
  const float* b[DMV_MAX_LAYERS];
  float* packed;
};

__global__ void k_pack(PackArgs a) {
  const int l = blockIdx.y;
  const int N = a.L.N[l], K = a.L.K[l];
  const int nt = (N + 31) / 32, S = a.L.S[l];
  const int kt = (K + 31) / 32, Sn = a.L.Sn[l];
  const int64_t nfwd = (int64_t)nt * S * 256, ntr = (int64_t)kt * Sn * 256, nb = (int64_t)nt * 32;
  const float* W = a.w[l];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nfwd + ntr + nb;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nfwd) {
      const int e = i & 3, lane = (i >> 2) & 63;
      const int64_t g = i >> 8;
      const int s = g % S, t = g / S;
      const int n = 32 * t + (lane & 31), k = 8 * s + 4 * (lane >> 5) + e;
      a.packed[a.L.w[l] + i] = (n < N && k < K) ? W[(int64_t)n * K + k] : 0.f;
    } else if (i < nfwd + ntr) {
      const int64_t j = i - nfwd;
      const int e = j & 3, lane = (j >> 2) & 63;
      const int64_t g = j >> 8;
      const int s = g % Sn, t = g / Sn;
      const int k = 32 * t + (lane & 31), n = 8 * s + 4 * (lane >> 5) + e;
      a.packed[a.L.wt[l] + j] = (n < N && k < K) ? W[(int64_t)n * K + k] : 0.f;
    } else {
      const int64_t j = i - nfwd - ntr;
      a.packed[a.L.b[l] + j] = (j < N) ? a.b[l][j] : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------
// device building blocks
// ------------------------------------------------------------------------------------------
// acc[t] += sum over the 32*NT inputs held in `in`  (Wimg: fragment image with S = 4*NT groups).
// Weight fragments are fetched PF groups ahead of their use (each group = NT coalesced 1 KiB loads
// feeding 4*NT MFMAs), so an L2 / Infinity-Cache round trip hides under the wave's own MFMAs.
//
// Two optional side streams ride on the same software pipeline, one 16-byte access per lane and group,
// so that no burst of memory traffic (and no wait for it) sits between two GEMMs:
//   ST: `in` is written to st[8 s .. 8 s + 3] in group s (st = this lane's row + 4 h): group s multiplies
//       exactly the four registers that hold those features.  Rows are padded to a multiple of 32 in the
//       saved buffers, so the store needs no predicate.
//   MK: the sign bits of a saved activation row (mk_src = that row + 4 h) are collected into mk: bit
//       16 (t & 1) + r of word t >> 1 is set when element r of tile t is > 0 (relu' of the layer whose
//       gradient `acc` is).
constexpr int mask_words(int NT) { return (NT + 1) / 2; }

// wave-uniform base + 32-bit lane offset: selects the scalar-base addressing mode (no 64-bit vector adds,
// no address registers per fragment)
__device__ __forceinline__ f32x4 frag_load(const f32x4* __restrict__ base, int frag, uint32_t lane_bytes) {
  return *(const f32x4*)((const char*)(base + (size_t)frag * 64) + lane_bytes);
}

template <int NT, int PF, bool ST, bool MK>
__device__ __forceinline__ void gemm_hh(const f32x4* __restrict__ Wimg, int lane, const f32x16 (&in)[NT],
                                        f32x16 (&acc)[NT], float* __restrict__ st, const float* __restrict__ mk_src,
                                        uint32_t (&mk)[mask_words(NT)]) {
  constexpr int S = 4 * NT;
  const uint32_t lb = (uint32_t)lane * 16u;
  f32x4 w[PF + 1][NT];
  f32x4 yv[PF + 1];
  if (MK) {
#pragma unroll
    for (int i = 0; i < mask_words(NT); ++i) mk[i] = 0u;
  }
#pragma unroll
  for (int p = 0; p < PF; ++p) {
#pragma unroll
    for (int t = 0; t < NT; ++t) w[p][t] = frag_load(Wimg, t * S + p, lb);
    if (MK) yv[p] = *(const f32x4*)(mk_src + 8 * p);
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (s + PF < S) {
#pragma unroll
      for (int t = 0; t < NT; ++t) w[(s + PF) % (PF + 1)][t] = frag_load(Wimg, t * S + s + PF, lb);
      if (MK) yv[(s + PF) % (PF + 1)] = *(const f32x4*)(mk_src + 8 * (s + PF));
    }
    if (ST) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = in[s >> 2][4 * (s & 3) + e];
      *(f32x4*)(st + 8 * s) = o;
    }
    // keep the memory operations of this group ahead of its MFMAs (the scheduler otherwise sinks them next to
    // their first use, which exposes the whole L2 latency)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float b = in[s >> 2][4 * (s & 3) + e];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = MFMA(w[s % (PF + 1)][t][e], b, acc[t]);
    }
    if (MK) {
      const f32x4 y = yv[s % (PF + 1)];
      uint32_t bits = 0u;
#pragma unroll
      for (int e = 0; e < 4; ++e) bits |= (y[e] > 0.f ? 1u : 0u) << e;
      mk[s >> 3] |= bits << (16 * ((s >> 2) & 1) + 4 * (s & 3));
    }
  }
}

// one 32-wide output tile: acc += W_tile in, optionally storing `in` on the way (see gemm_hh)
template <int NT, bool ST>
__device__ __forceinline__ void gemm_h1(const f32x4* __restrict__ Wtile, int lane, const f32x16 (&in)[NT], f32x16& acc,
                                        float* __restrict__ st) {
  constexpr int S = 4 * NT;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const f32x4 w = Wtile[s * 64 + lane];
    if (ST) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = in[s >> 2][4 * (s & 3) + e];
      *(f32x4*)(st + 8 * s) = o;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = MFMA(w[e], in[s >> 2][4 * (s & 3) + e], acc);
  }
}

// sign bits of a whole saved row, gemm_hh's MK layout
template <int NT>
__device__ __forceinline__ void load_mask(const float* __restrict__ src, uint32_t (&mk)[mask_words(NT)]) {
#pragma unroll
  for (int i = 0; i < mask_words(NT); ++i) mk[i] = 0u;
#pragma unroll
  for (int s = 0; s < 4 * NT; ++s) {
    const f32x4 y = *(const f32x4*)(src + 8 * s);
    uint32_t bits = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) bits |= (y[e] > 0.f ? 1u : 0u) << e;
    mk[s >> 3] |= bits << (16 * ((s >> 2) & 1) + 4 * (s & 3));
  }
}

// v = bit ? scale * v : 0
template <int NT>
__device__ __forceinline__ void apply_mask(f32x16 (&v)[NT], const uint32_t (&mk)[mask_words(NT)], float scale) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const bool on = (mk[t >> 1] >> (16 * (t & 1) + r)) & 1u;
      v[t][r] = on ? v[t][r] * scale : 0.f;
    }
}

template <int NT>
__device__ __forceinline__ void set_bias(f32x16 (&v)[NT], const float* __restrict__ bias, int h) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(const f32x4*)(bias + 32 * t + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[t][4 * q + e] = b[e];
    }
}

template <int NT>
__device__ __forceinline__ void add_bias(f32x16 (&v)[NT], const float* __restrict__ bias, int h) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *(const f32x4*)(bias + 32 * t + 8 * q + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[t][4 * q + e] += b[e];
    }
}

template <int NT>
__device__ __forceinline__ void zero(f32x16 (&v)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[t][r] = 0.f;
}

// store a register tile set to a row-major [rows][32*NT] buffer (this lane's row)
template <int NT>
__device__ __forceinline__ void store_rows(const f32x16 (&v)[NT], float* __restrict__ dst_row, int h) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[t][4 * q + e];
      *(f32x4*)(dst_row + 32 * t + 8 * q + 4 * h) = o;
    }
}

// relu (every shipped config) gets a branch-free path; the other activations share a generic one
template <int NT>
__device__ __forceinline__ void apply_act(f32x16 (&v)[NT], int act) {
  if (act == DMV_ACT_RELU) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[t][r] = fmaxf(v[t][r], 0.f);
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll 1
      for (int r = 0; r < 16; ++r) v[t][r] = dmv_act(v[t][r], act);
  }
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
struct FwdArgs {
  DmvLayout L;
  const float* packed;
  int nb, in_dim, hidden, act;
  float drop_scale;  // 1/(1-p), used only with drop_keep
  int in_mode, D;
  const float* x;
  int64_t ldx;
  const uint8_t* m;
  int64_t ldm;
  int64_t rows;
  float* y;
  int64_t ldy;
  float* xs;     // saved first-layer input [rows][32*T_in] (nullable)
  float* acts;   // saved activations [(2nb+1)][rows][32*NT] (nullable)
  const uint8_t* drop_keep;  // [nb][rows][hidden] (nullable)
};

__device__ __forceinline__ float first_layer_input(const FwdArgs& a, int64_t r, int k) {
  if (k >= a.in_dim) return 0.f;
  switch (a.in_mode) {
    case DMV_IN_PLAIN: return a.x[r * a.ldx + k];
    case DMV_IN_MASKED: return a.x[r * a.ldx + k] * (float)a.m[r * a.ldm + k];
    case DMV_IN_CONCAT:
      return k < a.D ? a.x[r * a.ldx + k] : (float)a.m[r * a.ldm + (k - a.D)];
    default:
      return k < a.D ? a.x[r * a.ldx + k] * (float)a.m[r * a.ldm + k] : (float)a.m[r * a.ldm + (k - a.D)];
  }
}

// SAVE: training forward (stores the first-layer input and every hidden activation for the backward;
// the saved buffers hold rows rounded up to a multiple of 32, so whole-wave stores need no predicate).
// A hidden activation is written while the next GEMM consumes it (gemm_hh / gemm_h1, ST).
template <int NT, bool SAVE>
__global__ __launch_bounds__(256, 1) void k_mlp_fwd(FwdArgs a) {
  constexpr int PF = 2;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
  if (row0 >= a.rows) return;
  const int m = lane & 31, h = lane >> 5;
  const int64_t row = row0 + m;
  const bool live = row < a.rows;
  const int64_t rl = live ? row : a.rows - 1;
  constexpr int Hp = 32 * NT;
  const int64_t act_stride = ((a.rows + 31) & ~(int64_t)31) * Hp;
  float* st = SAVE ? a.acts + row * Hp + 4 * h : nullptr;   // this lane's slot in the saved activations, layer 0
  uint32_t nomask[mask_words(NT)];

  f32x16 A[NT], T[NT];

  // ---- initial layer
  set_bias<NT>(A, a.packed + a.L.b[0], h);
  {
    const f32x4* W = (const f32x4*)(a.packed + a.L.w[0]);
    const int S = a.L.S_in;
    for (int s = 0; s < S; ++s) {
      float b[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = first_layer_input(a, rl, 8 * s + 4 * h + e);
      if (SAVE) {
        f32x4 o = {b[0], b[1], b[2], b[3]};
        *(f32x4*)(a.xs + row * (32 * a.L.T_in) + 8 * s + 4 * h) = o;
      }
      f32x4 w[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) w[t] = W[(t * S + s) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) A[t] = MFMA(w[t][e], b[e], A[t]);
    }
  }
  apply_act<NT>(A, a.act);

  // ---- residual blocks
  for (int blk = 0; blk < a.nb; ++blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    set_bias<NT>(T, a.packed + a.L.b[l1], h);
    gemm_hh<NT, PF, SAVE, false>((const f32x4*)(a.packed + a.L.w[l1]), lane, A, T, st + (l1 - 1) * act_stride, nullptr,
                                 nomask);
    apply_act<NT>(T, a.act);
    if (a.drop_keep) {
      const uint8_t* keep = a.drop_keep + ((int64_t)blk * a.rows + rl) * a.hidden;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
          const float kf = (n < a.hidden && keep[n]) ? a.drop_scale : 0.f;
          T[t][r] *= kf;
        }
    }
    add_bias<NT>(A, a.packed + a.L.b[l2], h);
    gemm_hh<NT, PF, SAVE, false>((const f32x4*)(a.packed + a.L.w[l2]), lane, T, A, st + l1 * act_stride, nullptr,
                                 nomask);
    apply_act<NT>(A, a.act);
  }

  // ---- final layer: one 32-wide output tile at a time
  {
    const int lf = a.L.nl - 1;
    const f32x4* W = (const f32x4*)(a.packed + a.L.w[lf]);
    const float* bias = a.packed + a.L.b[lf];
    constexpr int S = 4 * NT;
    for (int ot = 0; ot < a.L.OT; ++ot) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 b = *(const f32x4*)(bias + 32 * ot + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[4 * q + e] = b[e];
      }
      if (SAVE && ot == 0)
        gemm_h1<NT, true>(W, lane, A, acc, st + (lf - 1) * act_stride);
      else
        gemm_h1<NT, false>(W + (int64_t)ot * S * 64, lane, A, acc, nullptr);
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[4 * q + e];
          *(f32x4*)(a.y + row * a.ldy + 32 * ot + 8 * q + 4 * h) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward: input-gradient chain (writes the per-layer output gradients the weight-gradient
// GEMMs consume)
// ------------------------------------------------------------------------------------------
struct BwdArgs {
  DmvLayout L;
  const float* packed;
  int nb, hidden, act;
  float drop_scale;
  const float* acts;         // [(2nb+1)][rows][Hp]
  const uint8_t* drop_keep;  // nullable
  const float* gy;
  int64_t ldgy;
  int64_t rows;
  float* dacts;  // [(2nb+1)][rows][Hp]: gradient w.r.t. each hidden layer's pre-activation output
  float* dx;     // nullable, [rows][lddx]
  int64_t lddx;
};

// v *= act'(saved output)
template <int NT>
__device__ __forceinline__ void mul_act_grad(f32x16 (&v)[NT], const float* __restrict__ saved_row, int h, int act) {
  if (act == DMV_ACT_RELU) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 y = *(const f32x4*)(saved_row + 32 * t + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[t][4 * q + e] = y[e] > 0.f ? v[t][4 * q + e] : 0.f;
      }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 y = *(const f32x4*)(saved_row + 32 * t + 8 * q + 4 * h);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[t][4 * q + e] *= dmv_act_grad(y[e], act);
      }
  }
}

// RELU: the activation derivative is the sign bit of the saved output (post-dropout outputs included: a
// dropped or inactive unit saved 0).  Those bits are prefetched while the preceding GEMM runs (gemm_hh, MK),
// and every layer gradient is written while the following GEMM consumes it (ST), so the only memory waits
// left between GEMMs are the ones at the two ends of the chain.
template <int NT, bool RELU>
__global__ __launch_bounds__(256, 1) void k_mlp_bwd(BwdArgs a) {
  constexpr int PF = RELU ? 2 : 1;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 32;
  if (row0 >= a.rows) return;
  const int m = lane & 31, h = lane >> 5;
  const int64_t row = row0 + m;
  const bool live = row < a.rows;
  const int64_t rl = live ? row : a.rows - 1;
  constexpr int Hp = 32 * NT;
  const int64_t act_stride = ((a.rows + 31) & ~(int64_t)31) * Hp;
  const int lf = a.L.nl - 1;
  float* st = a.dacts + row * Hp + 4 * h;             // this lane's slot in the layer-gradient buffer, layer 0
  const float* sv = a.acts + rl * Hp + 4 * h;         // ... and in the saved activations
  const float dscale = a.drop_keep ? a.drop_scale : 1.f;
  uint32_t mkA[mask_words(NT)], mkT[mask_words(NT)];

  f32x16 dA[NT], dT[NT];
  if (RELU) load_mask<NT>(sv + (lf - 1) * act_stride, mkA);
  zero<NT>(dA);
  // dA = Wf^T gy
  {
    const f32x4* W = (const f32x4*)(a.packed + a.L.wt[lf]);
    const int S = a.L.S_out;
    for (int s = 0; s < S; ++s) {
      f32x4 b = *(const f32x4*)(a.gy + rl * a.ldgy + 8 * s + 4 * h);
      if (!live) b = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 w[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) w[t] = W[(t * S + s) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < NT; ++t) dA[t] = MFMA(w[t][e], b[e], dA[t]);
    }
  }
  for (int blk = a.nb - 1; blk >= 0; --blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    // through the block's output activation: dA becomes the gradient of (a_in + W2 t + b2)
    if (RELU) apply_mask<NT>(dA, mkA, 1.f);
    else mul_act_grad<NT>(dA, a.acts + l2 * act_stride + rl * Hp, h, a.act);
    zero<NT>(dT);
    gemm_hh<NT, PF, true, RELU>((const f32x4*)(a.packed + a.L.wt[l2]), lane, dA, dT, st + l2 * act_stride,
                                sv + l1 * act_stride, mkT);
    // through dropout and the inner activation
    if (RELU) {
      apply_mask<NT>(dT, mkT, dscale);
    } else if (!a.drop_keep) {
      mul_act_grad<NT>(dT, a.acts + l1 * act_stride + rl * Hp, h, a.act);
    } else {
      const float* trow = a.acts + l1 * act_stride + rl * Hp;
      const uint8_t* keep = a.drop_keep + ((int64_t)blk * a.rows + rl) * a.hidden;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 y = *(const f32x4*)(trow + 32 * t + 8 * q + 4 * h);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = 32 * t + 8 * q + 4 * h + e;
            const bool kp = n < a.hidden && keep[n];
            // the saved value is post-dropout: undo the 1/(1-p) scaling to get the activation output
            const float f = kp ? a.drop_scale * dmv_act_grad(y[e] / a.drop_scale, a.act) : 0.f;
            dT[t][4 * q + e] *= f;
          }
        }
    }
    // dA (skip path) += W1^T dT
    gemm_hh<NT, PF, true, RELU>((const f32x4*)(a.packed + a.L.wt[l1]), lane, dT, dA, st + l1 * act_stride,
                                sv + (l1 - 1) * act_stride, mkA);
  }
  if (RELU) apply_mask<NT>(dA, mkA, 1.f);
  else mul_act_grad<NT>(dA, a.acts + rl * Hp, h, a.act);

  if (a.dx) {
    const f32x4* W = (const f32x4*)(a.packed + a.L.wt[0]);
    constexpr int S = 4 * NT;
    for (int it = 0; it < a.L.T_in; ++it) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if (it == 0)
        gemm_h1<NT, true>(W, lane, dA, acc, st);
      else
        gemm_h1<NT, false>(W + (int64_t)it * S * 64, lane, dA, acc, nullptr);
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[4 * q + e];
          *(f32x4*)(a.dx + row * a.lddx + 32 * it + 8 * q + 4 * h) = o;
        }
      }
    }
  } else {
    store_rows<NT>(dA, a.dacts + row * Hp, h);
  }
}

// ------------------------------------------------------------------------------------------
// weight gradients: dW_l[n][k] = sum_r dY_l[r][n] X_l[r][k], db_l[n] = sum_r dY_l[r][n]
// One wave = a block of up to 4x4 32x32 tiles of one layer over one slice of the rows; operands go
// straight from row-major global memory into MFMA registers (lane = feature, reduction over rows).
// Partial sums go to a slab [split][layer image] and a second kernel folds the splits
// (fixed order: bitwise reproducible, no float atomics).
// ------------------------------------------------------------------------------------------
struct WgradLayer {
  const float* X;
  const float* dY;
  int64_t ldx, ldy;
  int nt, kt;          // 32-wide tiles of N and K
  int nblk, kblk;      // 4-tile blocks
  int splits;          // row slices of this layer (fewer for layers whose blocks hold fewer tiles)
  int64_t rows_per_split;
  int wave0;           // first wave of this layer; a layer owns nblk*kblk*splits consecutive waves
  int64_t slab_w, slab_b;  // offsets (floats) of this layer's split-0 image inside the slab
  int64_t slab_stride;     // floats per split of this layer
};

struct WgradArgs {
  WgradLayer ly[DMV_MAX_LAYERS];
  int nl;
  int total_waves;
  int64_t rows;
  float* slab;
};

// One block of TN x TK tiles over rows [r_begin, r_end).  Compile-time tile counts keep the loop free of
// branches; loads are unconditional with the row clamped into the slice, so no exec-masked region (and
// no vmcnt(0)) separates a load from its use.  A ring of 4 reduction steps (2 rows each) is kept in
// flight: a slot is refilled with the step 4 ahead right after its MFMAs issue (one wave per SIMD, so
// latency has to be covered by the wave's own MFMAs: 3 steps = 48 MFMAs at 4x4 tiles).
template <int TN, int TK>
__device__ __forceinline__ void wgrad_block(const WgradLayer& Ly, int nb0, int kb0, int64_t r_begin, int64_t r_end,
                                            float* __restrict__ slab, int lane) {
  const int j = lane & 31, h = lane >> 5;
  f32x16 acc[TN][TK];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][k][r] = 0.f;
  float bsum[TN];
#pragma unroll
  for (int i = 0; i < TN; ++i) bsum[i] = 0.f;

  const float* Yp = Ly.dY + 32 * nb0 + j;
  const float* Xp = Ly.X + 32 * kb0 + j;
  const int64_t r_last = r_end - 1;
  // small blocks issue few MFMAs per step, so they keep more steps in flight to cover the same latency
  constexpr int RING = (TN * TK >= 12) ? 4 : 8;
  float ya[RING][TN], xb[RING][TK];
  auto load_step = [&](int64_t rr0, float (&yy)[TN], float (&xx)[TK]) {
    const int64_t rr = rr0 + h;
    const int64_t rc = rr < r_end ? rr : r_last;
#pragma unroll
    for (int i = 0; i < TN; ++i) yy[i] = Yp[rc * Ly.ldy + 32 * i];
#pragma unroll
    for (int k = 0; k < TK; ++k) xx[k] = Xp[rc * Ly.ldx + 32 * k];
  };
  auto fma_step = [&](const float (&yy)[TN], const float (&xx)[TK]) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      bsum[i] += yy[i];
#pragma unroll
      for (int k = 0; k < TK; ++k) acc[i][k] = MFMA(yy[i], xx[k], acc[i][k]);
    }
  };
  const int64_t n_rows = r_end - r_begin;
  const int64_t n_steps = n_rows >> 1;       // steps with both rows inside the slice
  const int64_t n_groups = n_steps / RING;
#pragma unroll
  for (int u = 0; u < RING; ++u) load_step(r_begin + 2 * u, ya[u], xb[u]);
  int64_t r = r_begin;
  for (int64_t g = 0; g < n_groups; ++g, r += 2 * RING) {
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      fma_step(ya[u], xb[u]);
      load_step(r + 2 * RING + 2 * u, ya[u], xb[u]);
      // pin the program order: without it the scheduler sinks every refill to the end of the iteration
      // and the first slot of the next iteration is waited for right after it was issued
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // remainder: the ring already holds the next RING steps (rows clamped)
  const int rem = (int)(n_steps % RING);
#pragma unroll
  for (int u = 0; u < RING - 1; ++u)
    if (u < rem) fma_step(ya[u], xb[u]);
  if (n_rows & 1) {
    // a last single row: only the h = 0 half is inside the slice
    float yy[TN], xx[TK];
#pragma unroll
    for (int u = 0; u < RING; ++u)
      if (u == rem) {
#pragma unroll
        for (int i = 0; i < TN; ++i) yy[i] = h == 0 ? ya[u][i] : 0.f;
#pragma unroll
        for (int k = 0; k < TK; ++k) xx[k] = h == 0 ? xb[u][k] : 0.f;
      }
    fma_step(yy, xx);
  }

  const int Kp = 32 * Ly.kt;
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int k = 0; k < TK; ++k)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int n = 32 * (nb0 + i) + (rr & 3) + 8 * (rr >> 2) + 4 * h;
        slab[Ly.slab_w + (int64_t)n * Kp + 32 * (kb0 + k) + j] = acc[i][k][rr];
      }
  if (kb0 == 0) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const float s = bsum[i] + __shfl_xor(bsum[i], 32);
      if (h == 0) slab[Ly.slab_b + 32 * (nb0 + i) + j] = s;
    }
  }
}

template <int TN>
__device__ __forceinline__ void wgrad_dispatch_k(int tk, const WgradLayer& Ly, int nb0, int kb0, int64_t r0, int64_t r1,
                                                 float* slab, int lane) {
  switch (tk) {
    case 1: wgrad_block<TN, 1>(Ly, nb0, kb0, r0, r1, slab, lane); break;
    case 2: wgrad_block<TN, 2>(Ly, nb0, kb0, r0, r1, slab, lane); break;
    case 3: wgrad_block<TN, 3>(Ly, nb0, kb0, r0, r1, slab, lane); break;
    default: wgrad_block<TN, 4>(Ly, nb0, kb0, r0, r1, slab, lane); break;
  }
}

__global__ __launch_bounds__(256, 1) void k_mlp_wgrad(WgradArgs a) {
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; readfirstlane lets the compiler keep the whole work decomposition scalar
  const int wid = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wid >= a.total_waves) return;
  int l = 0;
  while (l + 1 < a.nl && a.ly[l + 1].wave0 <= wid) ++l;
  const WgradLayer& Ly = a.ly[l];
  const int nblocks = Ly.nblk * Ly.kblk;
  const int rel = wid - Ly.wave0;
  const int split = rel / nblocks, bi = rel % nblocks;   // the blocks of one row slice are neighbouring waves
  const int nb0 = (bi / Ly.kblk) * 4, kb0 = (bi % Ly.kblk) * 4;
  const int tn = min(4, Ly.nt - nb0), tk = min(4, Ly.kt - kb0);
  const int64_t r_begin = (int64_t)split * Ly.rows_per_split;
  const int64_t r_end = min(a.rows, r_begin + Ly.rows_per_split);
  float* slab = a.slab + (int64_t)split * Ly.slab_stride;
  switch (tn) {
    case 1: wgrad_dispatch_k<1>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane); break;
    case 2: wgrad_dispatch_k<2>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane); break;
    case 3: wgrad_dispatch_k<3>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane); break;
    default: wgrad_dispatch_k<4>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane); break;
  }
}

struct ReduceArgs {
  int nl;
  int N[DMV_MAX_LAYERS], K[DMV_MAX_LAYERS], Kp[DMV_MAX_LAYERS];
  int64_t slab_w[DMV_MAX_LAYERS], slab_b[DMV_MAX_LAYERS];
  int64_t first[DMV_MAX_LAYERS + 1];  // prefix sums of N*K + N
  float* gw[DMV_MAX_LAYERS];
  float* gb[DMV_MAX_LAYERS];
  int splits[DMV_MAX_LAYERS];
  int64_t slab_stride[DMV_MAX_LAYERS];
  const float* slab;
  float beta;
};

__global__ void k_wgrad_reduce(ReduceArgs a) {
  const int64_t total = a.first[a.nl];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int l = 0;
    while (i >= a.first[l + 1]) ++l;
    const int64_t e = i - a.first[l];
    const int64_t nk = (int64_t)a.N[l] * a.K[l];
    int64_t src;
    float* dst;
    if (e < nk) {
      const int n = (int)(e / a.K[l]), k = (int)(e % a.K[l]);
      src = a.slab_w[l] + (int64_t)n * a.Kp[l] + k;
      dst = a.gw[l] + e;
    } else {
      src = a.slab_b[l] + (e - nk);
      dst = a.gb[l] + (e - nk);
    }
    float s = 0.f;
    for (int p = 0; p < a.splits[l]; ++p) s += a.slab[(int64_t)p * a.slab_stride[l] + src];
    *dst = (a.beta != 0.f) ? a.beta * (*dst) + s : s;
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
extern "C" int32_t dmv_pad32(int32_t n) { return 32 * ((n + 31) / 32); }

extern "C" size_t dmv_mlp_packed_floats(const DmvNet* net) {
  if (dmv_check_net(net) != DMV_OK) return 0;
  return (size_t)dmv_layout(net).total;
}

extern "C" int dmv_mlp_pack(const DmvNet* net, const float* const* weights, const float* const* biases, float* packed,
                            void* stream) {
  int rc = dmv_check_net(net);
  if (rc) return rc;
  if (!weights || !biases || !packed) return DMV_E_BADARG;
  PackArgs a;
  a.L = dmv_layout(net);
  for (int l = 0; l < a.L.nl; ++l) {
    if (!weights[l] || !biases[l]) return DMV_E_BADARG;
    a.w[l] = weights[l];
    a.b[l] = biases[l];
  }
  a.packed = packed;
  dim3 grid(64, a.L.nl);
  hipLaunchKernelGGL(k_pack, grid, dim3(256), 0, (hipStream_t)stream, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" size_t dmv_mlp_saved_floats(const DmvNet* net, int64_t rows) {
  if (dmv_check_net(net) != DMV_OK || rows < 0) return 0;
  const int64_t Hp = dmv_pad32(net->hidden), Ip = dmv_pad32(net->in_dim);
  const int64_t rp = (rows + 31) & ~(int64_t)31;   // whole 32-row wave tiles: the kernels store without predicates
  return (size_t)(rp * Ip + (2 * net->nb + 1) * rp * Hp);
}

template <int NT>
static int launch_fwd(const FwdArgs& a, hipStream_t st) {
  const int64_t waves = (a.rows + 31) / 32;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  if (a.acts)
    hipLaunchKernelGGL((k_mlp_fwd<NT, true>), dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_mlp_fwd<NT, false>), dim3(grid), dim3(256), 0, st, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

template <int NT>
static int launch_bwd(const BwdArgs& a, hipStream_t st) {
  const int64_t waves = (a.rows + 31) / 32;
  const unsigned grid = (unsigned)((waves + 3) / 4);
  if (a.act == DMV_ACT_RELU)
    hipLaunchKernelGGL((k_mlp_bwd<NT, true>), dim3(grid), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_mlp_bwd<NT, false>), dim3(grid), dim3(256), 0, st, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

#define DMV_DISPATCH_NT(NTV, FN, ...)             \
  switch (NTV) {                                  \
    case 1: return FN<1>(__VA_ARGS__);            \
    case 2: return FN<2>(__VA_ARGS__);            \
    case 3: return FN<3>(__VA_ARGS__);            \
    case 4: return FN<4>(__VA_ARGS__);            \
    case 5: return FN<5>(__VA_ARGS__);            \
    case 6: return FN<6>(__VA_ARGS__);            \
    case 7: return FN<7>(__VA_ARGS__);            \
    case 8: return FN<8>(__VA_ARGS__);            \
    default: return DMV_E_UNSUPPORTED;            \
  }

extern "C" int dmv_mlp_forward(const DmvNet* net, const float* packed, int32_t in_mode, int32_t D, const float* x,
                               int64_t ldx, const uint8_t* m, int64_t ldm, int64_t rows, float* y, int64_t ldy,
                               float* saved, const uint8_t* drop_keep, void* stream) {
  int rc = dmv_check_net(net);
  if (rc) return rc;
  if (!packed || !x || !y || rows < 0) return DMV_E_BADARG;
  if (rows == 0) return DMV_OK;
  if (in_mode < DMV_IN_PLAIN || in_mode > DMV_IN_CONCAT_MASKED) return DMV_E_UNSUPPORTED;
  if (in_mode != DMV_IN_PLAIN && !m) return DMV_E_BADARG;
  const int want_in = (in_mode >= DMV_IN_CONCAT) ? 2 * D : D;
  if (want_in != net->in_dim || ldx < D || (m && ldm < D)) return DMV_E_BADARG;
  if (ldy < dmv_pad32(net->out_dim) || (ldy & 3) || ((uintptr_t)y & 15)) return DMV_E_BADARG;
  if (net->dropout_p > 0.f && !drop_keep && saved) return DMV_E_BADARG;  // training-mode dropout needs its masks
  FwdArgs a;
  a.L = dmv_layout(net);
  a.packed = packed;
  a.nb = net->nb; a.in_dim = net->in_dim; a.hidden = net->hidden; a.act = net->act;
  a.drop_scale = 1.f / (1.f - net->dropout_p);
  a.in_mode = in_mode; a.D = D;
  a.x = x; a.ldx = ldx; a.m = m; a.ldm = ldm; a.rows = rows;
  a.y = y; a.ldy = ldy;
  a.xs = saved;
  a.acts = saved ? saved + ((rows + 31) & ~(int64_t)31) * dmv_pad32(net->in_dim) : nullptr;
  a.drop_keep = drop_keep;
  DMV_DISPATCH_NT(a.L.NT, launch_fwd, a, (hipStream_t)stream);
}

// ---- backward host side
// Work decomposition of the weight-gradient GEMMs.  A wave owns one block (up to 4x4 tiles) of one layer over
// one row slice; slices are sized so that every wave issues about the same number of MFMAs (a layer whose
// blocks hold c of 16 tiles gets slices 16/c times longer), and the wave count aims at one full round of the
// 1024 wave slots (256 CUs x 4 SIMDs, one wave per SIMD).
static int64_t wgrad_plan(const DmvNet* net, int64_t rows, WgradArgs* wa, ReduceArgs* ra) {
  DmvLayout L = dmv_layout(net);
  double weight = 0.0;   // waves per unit of slices, in units of a full 16-tile block
  int cost[DMV_MAX_LAYERS];
  for (int l = 0; l < L.nl; ++l) {
    WgradLayer& y = wa->ly[l];
    y.nt = (L.N[l] + 31) / 32;
    y.kt = (L.K[l] + 31) / 32;
    y.nblk = (y.nt + 3) / 4;
    y.kblk = (y.kt + 3) / 4;
    cost[l] = (y.nt < 4 ? y.nt : 4) * (y.kt < 4 ? y.kt : 4);
    if (cost[l] < 8) cost[l] = 8;   // a small block is bound by load latency, not MFMA issue: at most 2x longer slices
    weight += (double)y.nblk * y.kblk * cost[l] / 16.0;
  }
  int64_t base_splits = (int64_t)(1024.0 / weight);
  if (base_splits < 1) base_splits = 1;
  const int64_t max_splits = (rows + 63) / 64;
  if (base_splits > max_splits) base_splits = max_splits;
  int64_t base_rps = (rows + base_splits - 1) / base_splits;
  int wave0 = 0;
  int64_t slab = 0;
  for (int l = 0; l < L.nl; ++l) {
    WgradLayer& y = wa->ly[l];
    int64_t rps = base_rps * 16 / cost[l];
    rps = (rps + 15) / 16 * 16;
    y.rows_per_split = rps;
    y.splits = (int)((rows + rps - 1) / rps);
    y.wave0 = wave0;
    wave0 += y.nblk * y.kblk * y.splits;
    y.slab_stride = (int64_t)y.nt * 32 * y.kt * 32 + (int64_t)y.nt * 32;
    y.slab_w = slab;
    y.slab_b = slab + (int64_t)y.nt * 32 * y.kt * 32;
    slab += y.slab_stride * y.splits;
  }
  wa->nl = L.nl;
  wa->total_waves = wave0;
  wa->rows = rows;
  if (ra) {
    ra->nl = L.nl;
    int64_t first = 0;
    for (int l = 0; l < L.nl; ++l) {
      ra->N[l] = L.N[l]; ra->K[l] = L.K[l]; ra->Kp[l] = wa->ly[l].kt * 32;
      ra->slab_w[l] = wa->ly[l].slab_w; ra->slab_b[l] = wa->ly[l].slab_b;
      ra->splits[l] = wa->ly[l].splits; ra->slab_stride[l] = wa->ly[l].slab_stride;
      ra->first[l] = first;
      first += (int64_t)L.N[l] * L.K[l] + L.N[l];
    }
    ra->first[L.nl] = first;
  }
  return slab;
}

extern "C" size_t dmv_mlp_backward_ws_bytes(const DmvNet* net, int64_t rows) {
  if (dmv_check_net(net) != DMV_OK || rows < 0) return 0;
  if (rows == 0) return 16;
  WgradArgs wa;
  const int64_t slab_floats = wgrad_plan(net, rows, &wa, nullptr);
  const int64_t Hp = dmv_pad32(net->hidden);
  const int64_t dacts = (2 * net->nb + 1) * ((rows + 31) & ~(int64_t)31) * Hp;
  return (size_t)(dacts + slab_floats) * sizeof(float);
}

extern "C" int dmv_mlp_backward(const DmvNet* net, const float* packed, const float* saved, const uint8_t* drop_keep,
                                const float* gy, int64_t ldgy, int64_t rows, float* dx, int64_t lddx,
                                float* const* grad_w, float* const* grad_b, float beta, void* ws, size_t ws_bytes,
                                void* stream) {
  int rc = dmv_check_net(net);
  if (rc) return rc;
  if (!packed || !saved || !gy || rows < 0 || !ws) return DMV_E_BADARG;
  if ((grad_w == nullptr) != (grad_b == nullptr)) return DMV_E_BADARG;
  if (rows == 0) return DMV_OK;
  if (ldgy < dmv_pad32(net->out_dim) || (ldgy & 3) || ((uintptr_t)gy & 15)) return DMV_E_BADARG;
  if (dx && (lddx < dmv_pad32(net->in_dim) || (lddx & 3) || ((uintptr_t)dx & 15))) return DMV_E_BADARG;
  if (net->dropout_p > 0.f && !drop_keep) return DMV_E_BADARG;
  if (ws_bytes < dmv_mlp_backward_ws_bytes(net, rows)) return DMV_E_WORKSPACE;
  if ((uintptr_t)ws & 15) return DMV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t Hp = dmv_pad32(net->hidden), Ip = dmv_pad32(net->in_dim);
  const int64_t rp = (rows + 31) & ~(int64_t)31;
  const float* xs = saved;
  const float* acts = saved + rp * Ip;
  float* dacts = (float*)ws;
  float* slab = dacts + (2 * net->nb + 1) * rp * Hp;

  BwdArgs b;
  b.L = dmv_layout(net);
  b.packed = packed;
  b.nb = net->nb; b.hidden = net->hidden; b.act = net->act;
  b.drop_scale = 1.f / (1.f - net->dropout_p);
  b.acts = acts; b.drop_keep = drop_keep;
  b.gy = gy; b.ldgy = ldgy; b.rows = rows;
  b.dacts = dacts; b.dx = dx; b.lddx = lddx;
  {
    int r2;
    switch (b.L.NT) {
      case 1: r2 = launch_bwd<1>(b, st); break;
      case 2: r2 = launch_bwd<2>(b, st); break;
      case 3: r2 = launch_bwd<3>(b, st); break;
      case 4: r2 = launch_bwd<4>(b, st); break;
      case 5: r2 = launch_bwd<5>(b, st); break;
      case 6: r2 = launch_bwd<6>(b, st); break;
      case 7: r2 = launch_bwd<7>(b, st); break;
      case 8: r2 = launch_bwd<8>(b, st); break;
      default: r2 = DMV_E_UNSUPPORTED;
    }
    if (r2) return r2;
  }
  if (!grad_w) return DMV_OK;

  WgradArgs wa;
  ReduceArgs ra;
  wgrad_plan(net, rows, &wa, &ra);
  const int nl = wa.nl;
  for (int l = 0; l < nl; ++l) {
    if (!grad_w[l] || !grad_b[l]) return DMV_E_BADARG;
    WgradLayer& y = wa.ly[l];
    if (l == 0) { y.X = xs; y.ldx = Ip; } else { y.X = acts + (int64_t)(l - 1) * rp * Hp; y.ldx = Hp; }
    if (l == nl - 1) { y.dY = gy; y.ldy = ldgy; } else { y.dY = dacts + (int64_t)l * rp * Hp; y.ldy = Hp; }
    ra.gw[l] = grad_w[l];
    ra.gb[l] = grad_b[l];
  }
  wa.slab = slab;
  ra.slab = slab;
  ra.beta = beta;
  hipLaunchKernelGGL(k_mlp_wgrad, dim3((unsigned)((wa.total_waves + 3) / 4)), dim3(256), 0, st, wa);
  DMV_HIP_CHECK_LAUNCH();
  const int64_t total = ra.first[nl];
  unsigned rgrid = (unsigned)((total + 255) / 256);
  if (rgrid > 4096) rgrid = 4096;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(rgrid), dim3(256), 0, st, ra);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}
