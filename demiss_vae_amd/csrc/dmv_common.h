// Shared device / host helpers for libdmv (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dmv.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DMV_MAX_LAYERS 18  // 2*nb+2 with nb <= 8
#define DMV_EPSILON 1e-5f  // vgiwae/shared/vae_enums.py:3

static inline int dmv_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Where each layer's fragment-ordered weights live inside the packed buffer (offsets in floats).
// Forward image of a layer y = W x (W [N][K]):   [ceil(N/32)][ceil(K/8)][64 lanes][4]
//   element (nt, s, lane, e) = W[32 nt + (lane & 31)][8 s + 4 (lane >> 5) + e]   (0 outside W)
// Transposed image (for dx = W^T dy):             [ceil(K/32)][ceil(N/8)][64][4]
//   element (kt, s, lane, e) = W[8 s + 4 (lane >> 5) + e][32 kt + (lane & 31)]
// Bias image: [ceil(N/32)][64 lanes] floats, an MFMA A-operand fragment per tile: lane n < 32 carries
//   bias[32 nt + n], lanes 32..63 carry zero (bias enters a layer as a rank-1 MFMA against a vector of ones).
struct DmvLayout {
  int nl;                           // layers = 2 nb + 2
  int NT;                           // hidden tiles = ceil(hidden/32)
  int S_in;                         // ceil(in_dim/8)
  int T_in;                         // ceil(in_dim/32)
  int OT;                           // ceil(out_dim/32)
  int S_out;                        // ceil(out_dim/8)
  int64_t w[DMV_MAX_LAYERS];        // forward images
  int64_t wt[DMV_MAX_LAYERS];       // transposed images
  int64_t b[DMV_MAX_LAYERS];        // bias images
  int N[DMV_MAX_LAYERS], K[DMV_MAX_LAYERS];
  int S[DMV_MAX_LAYERS];            // reduction groups of the forward image (K padded) / 8
  int Sn[DMV_MAX_LAYERS];           // reduction groups of the transposed image (N padded) / 8
  int64_t total;
};

static inline int dmv_check_net(const DmvNet* n) {
  if (!n) return DMV_E_BADARG;
  if (n->in_dim < 1 || n->out_dim < 1 || n->hidden < 1 || n->nb < 0) return DMV_E_BADARG;
  if (n->hidden > 512 || n->nb > 8 || n->in_dim > 4096 || n->out_dim > 4096) return DMV_E_UNSUPPORTED;
  // hidden widths 257..512 run on the four-wave and 16-row kernels only: relu
  if (n->hidden > 256 && n->act != DMV_ACT_RELU) return DMV_E_UNSUPPORTED;
  if (n->act < DMV_ACT_RELU || n->act > DMV_ACT_SIGMOID) return DMV_E_UNSUPPORTED;
  if (!(n->dropout_p >= 0.f && n->dropout_p < 1.f)) return DMV_E_BADARG;
  return DMV_OK;
}

// A dimension that is the hidden width is padded to 32*NT (activations live in 32-wide register
// tiles); the network's input and output widths are padded to a multiple of 8 (one fragment group).
static inline DmvLayout dmv_layout(const DmvNet* n) {
  DmvLayout L;
  L.nl = 2 * n->nb + 2;
  L.NT = dmv_ceil_div(n->hidden, 32);
  L.S_in = dmv_ceil_div(n->in_dim, 8);
  L.T_in = dmv_ceil_div(n->in_dim, 32);
  L.OT = dmv_ceil_div(n->out_dim, 32);
  L.S_out = dmv_ceil_div(n->out_dim, 8);
  const int Hp = 32 * L.NT;
  int64_t off = 0;
  for (int l = 0; l < L.nl; ++l) {
    L.N[l] = (l == L.nl - 1) ? n->out_dim : n->hidden;
    L.K[l] = (l == 0) ? n->in_dim : n->hidden;
    L.S[l] = (l == 0) ? L.S_in : Hp / 8;
    L.Sn[l] = (l == L.nl - 1) ? L.S_out : Hp / 8;
    const int64_t nt = dmv_ceil_div(L.N[l], 32), kt = dmv_ceil_div(L.K[l], 32);
    L.w[l] = off;  off += nt * L.S[l] * 256;
    L.wt[l] = off; off += kt * L.Sn[l] * 256;
    L.b[l] = off;  off += nt * 64;
  }
  L.total = off;
  return L;
}

#define DMV_HIP_CHECK_LAUNCH()                      \
  do {                                              \
    hipError_t e_ = hipGetLastError();              \
    if (e_ != hipSuccess) return (int)e_;           \
  } while (0)

__device__ __forceinline__ float dmv_act(float v, int act) {
  switch (act) {
    case DMV_ACT_RELU: return fmaxf(v, 0.f);
    case DMV_ACT_LEAKY_RELU: return v > 0.f ? v : 0.01f * v;
    case DMV_ACT_TANH: return tanhf(v);
    default: return 1.f / (1.f + __expf(-v));
  }
}

// derivative of the activation expressed through its OUTPUT y (all four are invertible enough for that)
__device__ __forceinline__ float dmv_act_grad(float y, int act) {
  switch (act) {
    case DMV_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case DMV_ACT_LEAKY_RELU: return y > 0.f ? 1.f : 0.01f;
    case DMV_ACT_TANH: return 1.f - y * y;
    default: return y * (1.f - y);
  }
}

// Transcendentals of the HBM-bound row kernels go through the hardware exp2 / log2 units (about 1 ulp; the
// argument scaling adds |x| * 6e-8 relative): a libm-grade expf + log1pf pair per element made those kernels
// ALU-bound.
__device__ __forceinline__ float dmv_exp(float x) { return __expf(x); }
__device__ __forceinline__ float dmv_log(float x) { return __logf(x); }
// log1p for t >= 0, relative accuracy of a few ulp over the whole range (softplus of a strongly negative argument, i.e.
// a standard deviation of 1e-3, must not lose digits: 1 + t rounds t to 6e-8 absolute).  Small t: the series (relative
// truncation error t^5 / 6 < 2e-10); otherwise log(u) t / (u - 1) with u = fl(1 + t), in which the rounding of u cancels.
__device__ __forceinline__ float dmv_log1p_pos(float t) {
  if (t < 0x1p-6f) return t * (1.f - t * (0.5f - t * (1.f / 3.f - t * (0.25f - t * 0.2f))));
  const float u = 1.f + t;
  return __logf(u) * (t * __frcp_rn(u - 1.f));
}
// torch.nn.functional.softplus, beta = 1, threshold = 20
__device__ __forceinline__ float dmv_softplus(float x) { return x > 20.f ? x : dmv_log1p_pos(dmv_exp(x)); }
__device__ __forceinline__ float dmv_sigmoid(float x) { return __frcp_rn(1.f + dmv_exp(-x)); }
