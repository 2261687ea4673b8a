// Optimiser step of the training loop: global gradient-norm clipping + Adam / AMSGrad over all parameter tensors of
// both networks in two launches (vgiwae/models/vae.py:349-390: torch.optim.Adam with two parameter groups, then
// clip_grad_norm_ over all parameters; torch needs about a dozen launches for the same work, and its multi-tensor
// Adam runs a 547 k-parameter model on ten workgroups).
//
//   k_opt_sumsq   partial sums of g^2 over fixed 1024-element quarters of 4096-element chunks (one workgroup per
//                 quarter: the summation order does not depend on the launch geometry); its first workgroup also
//                 advances the groups' step counters.  Without clipping k_opt_tick (one thread) advances them.
//   k_opt_apply   one workgroup per quarter chunk (a 547 k-parameter model fills the chip only at this grain;
//                 16-byte accesses where the tensors allow).  Every workgroup first sums the partials in the same
//                 order (a few hundred floats from L2) -> total norm -> clip coefficient min(1, max_norm / (norm +
//                 1e-6)) (torch's clip_grad_norm_) and forms its group's bias corrections from the step counter, then
//                 g' = coef g;  m = m + (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;  [vmax = max(vmax, v)];
//                 p -= (lr / (1 - b1^t)) m / (sqrt(v or vmax) / sqrt(1 - b2^t) + eps).
//                 With a pack map it also writes each new parameter value to its places in the owning network's MFMA
//                 fragment image (forward fragment, transposed fragment or bias fragment: the layout of k_pack in
//                 dmv_mlp.hip), so the next forward pass needs no packing launch.
// Tensor tables travel by value in the kernel arguments (like torch's multi-tensor kernels), so a recorded step
// replays against the same gradient addresses without host copies.  Learning rates and step counters are read from
// device memory: schedulers and replays change them without re-recording.
#include "dmv_common.h"
#include <mutex>

namespace {

constexpr int kChunk = 4096;
constexpr int kSub = 4;                 // workgroups per chunk
constexpr int kQuarter = kChunk / kSub;
constexpr int kMaxT = DMV_OPT_MAX_TENSORS;
constexpr int kMaxG = DMV_OPT_MAX_GROUPS;

struct OptTable {
  int n;
  int first_chunk[kMaxT + 1];   // prefix sums of ceil(numel / kChunk)
  float* p[kMaxT];
  const float* g[kMaxT];
  float* m[kMaxT];
  float* v[kMaxT];
  float* vmax[kMaxT];           // null: plain Adam
  int numel[kMaxT];
  signed char group[kMaxT];
};

struct OptGroups {
  int n;
  const float* lr[kMaxG];       // device scalars
  float* step[kMaxG];           // device scalars (float counters, like torch's capturable optimisers)
  float beta1[kMaxG], beta2[kMaxG], eps[kMaxG];
  float omb1[kMaxG], omb2[kMaxG];  // 1 - beta, rounded from double
  double beta1d[kMaxG], beta2d[kMaxG];
};

__device__ inline int find_tensor(const OptTable& t, int chunk) {
  int lo = 0, hi = t.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t.first_chunk[mid] <= chunk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ inline float block_sum(float x, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = x;
  __syncthreads();
  float s = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;   // valid in thread 0
}

__device__ inline void tick(const OptGroups& gr) {
  for (int g = 0; g < gr.n; ++g) gr.step[g][0] = gr.step[g][0] + 1.f;
}

__global__ void k_opt_sumsq(OptTable t, OptGroups gr, float* __restrict__ partial) {
  __shared__ float red[4];
  const int chunk = blockIdx.x / kSub;
  const int ti = find_tensor(t, chunk);
  const int base = (chunk - t.first_chunk[ti]) * kChunk + (blockIdx.x % kSub) * kQuarter;
  const int end = min(base + kQuarter, t.numel[ti]);
  const float* g = t.g[ti];
  float acc = 0.f;
  for (int i = base + threadIdx.x; i < end; i += blockDim.x) { const float x = g[i]; acc += x * x; }
  const float s = block_sum(acc, red);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = s;
    if (blockIdx.x == 0) tick(gr);
  }
}

__global__ void k_opt_tick(OptGroups gr) { tick(gr); }

struct AdamCoef {
  float coef, b2, eps, omb1, omb2, step_size, bc2s;
};

__device__ __forceinline__ void adam_one(const AdamCoef& c, float g, float& p, float& m, float& v, float* vmax) {
  const float gs = g * c.coef;
  m = m + c.omb1 * (gs - m);
  v = c.b2 * v + c.omb2 * gs * gs;
  float d = v;
  if (vmax) { d = fmaxf(*vmax, v); *vmax = d; }
  p = p - c.step_size * (m / (sqrtf(d) / c.bc2s + c.eps));
}

// new value of element i of a tensor -> its places in the network's fragment image
__device__ __forceinline__ void image_store(const DmvPackMap& pm, int i, float val) {
  if (pm.is_bias) { pm.image[pm.fwd_off + (int64_t)(i >> 5) * 64 + (i & 31)] = val; return; }
  const int n = i / pm.K, k = i - n * pm.K;
  pm.image[pm.fwd_off + 4 * ((int64_t)((n >> 5) * pm.S + (k >> 3)) * 64 + (n & 31) + 32 * ((k >> 2) & 1)) + (k & 3)] = val;
  pm.image[pm.tr_off + 4 * ((int64_t)((k >> 5) * pm.Sn + (n >> 3)) * 64 + (k & 31) + 32 * ((n >> 2) & 1)) + (n & 3)] = val;
}

__global__ void k_opt_apply(OptTable t, OptGroups gr, int n_partials, float max_norm, const float* __restrict__ partial,
                            float* norm_out, const DmvPackMap* __restrict__ pack_map, uint64_t* counter, uint64_t counter_add) {
  __shared__ float red[4];
  __shared__ float sh[3];      // clip coefficient, lr / (1 - b1^t), sqrt(1 - b2^t)
  __shared__ __attribute__((aligned(16))) float turn[kQuarter];
  const int chunk = blockIdx.x / kSub;
  const int ti = find_tensor(t, chunk);
  const int base = (chunk - t.first_chunk[ti]) * kChunk + (blockIdx.x % kSub) * kQuarter;
  const int end = min(base + kQuarter, t.numel[ti]);
  const int gi = t.group[ti];
  {
    float acc = 0.f;
    if (max_norm > 0.f)
      for (int i = threadIdx.x; i < n_partials; i += blockDim.x) acc += partial[i];
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) {
      float coef = 1.f;
      if (max_norm > 0.f) {
        const float norm = sqrtf(tot);
        if (norm_out && blockIdx.x == 0) norm_out[0] = norm;
        coef = fminf(max_norm / (norm + 1e-6f), 1.f);
      }
      if (counter && blockIdx.x == 0) counter[0] += counter_add;   // nothing else in this launch reads it
      const double sd = (double)gr.step[gi][0];                         // already advanced for this step
      sh[0] = coef;
      sh[1] = gr.lr[gi][0] / (float)(1.0 - pow(gr.beta1d[gi], sd));     // double, like torch's host-side scalars
      sh[2] = (float)sqrt(1.0 - pow(gr.beta2d[gi], sd));
    }
    __syncthreads();
  }
  if (base >= end) return;
  AdamCoef c;
  c.coef = sh[0]; c.b2 = gr.beta2[gi]; c.eps = gr.eps[gi]; c.omb1 = gr.omb1[gi]; c.omb2 = gr.omb2[gi];
  c.step_size = sh[1]; c.bc2s = sh[2];
  float* p = t.p[ti];
  const float* g = t.g[ti];
  float* m = t.m[ti];
  float* v = t.v[ti];
  float* vm = t.vmax[ti];
  DmvPackMap pm;
  pm.image = nullptr;
  if (pack_map) pm = pack_map[ti];
  // a quarter starts at a multiple of 1024 elements: 16-byte accesses need only aligned tensor bases
  const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vm) & 15) == 0;
  const int i4 = base + 4 * (int)threadIdx.x;       // blockDim.x * 4 == kQuarter
  const bool whole = vec && end - base == kQuarter;  // (uniform over the workgroup: every thread takes the 16-byte path)
  if (vec && i4 + 4 <= end) {
    f32x4 P = *(f32x4*)(p + i4), M = *(f32x4*)(m + i4), V = *(f32x4*)(v + i4), X;
    const f32x4 G = *(const f32x4*)(g + i4);
    if (vm) X = *(f32x4*)(vm + i4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = P[e], me = M[e], ve = V[e], xe = vm ? X[e] : 0.f;
      adam_one(c, G[e], pe, me, ve, vm ? &xe : nullptr);
      P[e] = pe; M[e] = me; V[e] = ve; X[e] = xe;
    }
    *(f32x4*)(p + i4) = P; *(f32x4*)(m + i4) = M; *(f32x4*)(v + i4) = V;
    if (vm) *(f32x4*)(vm + i4) = X;
    if (pm.image) {
      if (!pm.is_bias && (pm.K & 3) == 0) {
        // four consecutive inputs of one output row: one 16-byte store into the forward fragment
        const int n = i4 / pm.K, k = i4 - n * pm.K;
        *(f32x4*)(pm.image + pm.fwd_off + 4 * ((int64_t)((n >> 5) * pm.S + (k >> 3)) * 64 + (n & 31) + 32 * ((k >> 2) & 1))) = P;
        // The transposed fragment wants four consecutive OUTPUT rows of one input as 16 bytes.  A whole quarter that
        // holds complete rows in multiples of four (K divides 256: every hidden-to-hidden layer) is turned through
        // LDS, one coalesced 16-byte store per thread; anything else writes four single words.
        const int R = kQuarter / pm.K;
        if (whole && R * pm.K == kQuarter && (R & 3) == 0) {
          *(f32x4*)(turn + 4 * threadIdx.x) = P;
          __syncthreads();
          const int kk = threadIdx.x % pm.K, q = threadIdx.x / pm.K;
          const int n0 = base / pm.K + 4 * q;
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = turn[(4 * q + e) * pm.K + kk];
          *(f32x4*)(pm.image + pm.tr_off + 4 * ((int64_t)((kk >> 5) * pm.Sn + (n0 >> 3)) * 64 + (kk & 31) + 32 * ((n0 >> 2) & 1))) = o;
        } else {
          float* tr = pm.image + pm.tr_off + 4 * ((int64_t)((k >> 5) * pm.Sn + (n >> 3)) * 64 + (k & 31) + 32 * ((n >> 2) & 1)) + (n & 3);
#pragma unroll
          for (int e = 0; e < 4; ++e) tr[4 * e] = P[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) image_store(pm, i4 + e, P[e]);
      }
    }
    return;
  }
  for (int i = i4; i < min(i4 + 4, end); ++i) {
    adam_one(c, g[i], p[i], m[i], v[i], vm ? vm + i : nullptr);
    if (pm.image) image_store(pm, i, p[i]);
  }
}

}  // namespace

extern "C" size_t dmv_opt_ws_floats(int32_t n, const int64_t* numel) {
  if (n < 1 || n > kMaxT || !numel) return 0;
  size_t chunks = 0;
  for (int i = 0; i < n; ++i) {
    if (numel[i] < 0 || numel[i] > 0x7fffffff) return 0;
    chunks += (size_t)((numel[i] + kChunk - 1) / kChunk);
  }
  return kSub * chunks + 1;
}

// Both entry points: argument checks, the tables, the two launches.  ws: a device address with room for
// dmv_opt_ws_floats(n, numel) floats (the caller's workspace, or this GPU's copy of the static array below).
static int adam_step(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg,
                     float* const* exp_avg_sq, float* const* max_exp_avg_sq, const int64_t* numel,
                     const int32_t* group_of, int32_t n_groups, const DmvAdamGroup* groups, float max_norm,
                     float* ws, float* norm_out, const DmvPackMap* pack_map, uint64_t* counter, uint64_t counter_add,
                     void* stream) {
  OptTable t;
  t.n = n;
  int chunks = 0;
  for (int i = 0; i < n; ++i) {
    if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i] || group_of[i] < 0 || group_of[i] >= n_groups) return DMV_E_BADARG;
    t.first_chunk[i] = chunks;
    chunks += (int)((numel[i] + kChunk - 1) / kChunk);
    t.p[i] = params[i]; t.g[i] = grads[i]; t.m[i] = exp_avg[i]; t.v[i] = exp_avg_sq[i];
    t.vmax[i] = max_exp_avg_sq ? max_exp_avg_sq[i] : nullptr;
    t.numel[i] = (int)numel[i];
    t.group[i] = (signed char)group_of[i];
  }
  t.first_chunk[n] = chunks;
  for (int i = n; i < kMaxT; ++i) { t.first_chunk[i + 1] = chunks; t.numel[i] = 0; t.group[i] = 0; t.p[i] = nullptr; t.g[i] = nullptr; t.m[i] = nullptr; t.v[i] = nullptr; t.vmax[i] = nullptr; }
  OptGroups gr;
  gr.n = n_groups;
  for (int g = 0; g < kMaxG; ++g) {
    const bool live = g < n_groups;
    if (live && (!groups[g].lr || !groups[g].step)) return DMV_E_BADARG;
    gr.lr[g] = live ? groups[g].lr : nullptr;
    gr.step[g] = live ? groups[g].step : nullptr;
    gr.beta1[g] = live ? (float)groups[g].beta1 : 0.f;
    gr.beta2[g] = live ? (float)groups[g].beta2 : 0.f;
    gr.eps[g] = live ? (float)groups[g].eps : 0.f;
    gr.beta1d[g] = live ? groups[g].beta1 : 0.0;
    gr.beta2d[g] = live ? groups[g].beta2 : 0.0;
    gr.omb1[g] = live ? (float)(1.0 - groups[g].beta1) : 0.f;
    gr.omb2[g] = live ? (float)(1.0 - groups[g].beta2) : 0.f;
  }
  if (chunks == 0) return DMV_OK;
  float* partial = ws;
  hipStream_t s = (hipStream_t)stream;
  if (max_norm > 0.f) hipLaunchKernelGGL(k_opt_sumsq, dim3((unsigned)(kSub * chunks)), dim3(256), 0, s, t, gr, partial);
  else hipLaunchKernelGGL(k_opt_tick, dim3(1), dim3(1), 0, s, gr);
  DMV_HIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_opt_apply, dim3((unsigned)(kSub * chunks)), dim3(kQuarter / 4), 0, s, t, gr, kSub * chunks, max_norm,
                     partial, norm_out, pack_map, counter, counter_add);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

static int check_tables(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, const int32_t* group_of, int32_t n_groups,
                        const DmvAdamGroup* groups) {
  if (n < 1 || n > kMaxT || n_groups < 1 || n_groups > kMaxG) return DMV_E_UNSUPPORTED;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !group_of || !groups) return DMV_E_BADARG;
  return DMV_OK;
}

extern "C" int dmv_opt_adam_step(int32_t n, float* const* params, const float* const* grads, float* const* exp_avg,
                                 float* const* exp_avg_sq, float* const* max_exp_avg_sq, const int64_t* numel,
                                 const int32_t* group_of, int32_t n_groups, const DmvAdamGroup* groups, float max_norm,
                                 float* ws, size_t ws_floats, float* norm_out, const DmvPackMap* pack_map, uint64_t* counter,
                                 uint64_t counter_add, void* stream) {
  const int rc = check_tables(n, params, grads, exp_avg, exp_avg_sq, numel, group_of, n_groups, groups);
  if (rc) return rc;
  if (!ws) return DMV_E_BADARG;
  if (ws_floats < dmv_opt_ws_floats(n, numel) || dmv_opt_ws_floats(n, numel) == 0) return DMV_E_WORKSPACE;
  return adam_step(n, params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq, numel, group_of, n_groups, groups, max_norm,
                   ws, norm_out, pack_map, counter, counter_add, stream);
}

// ------------------------------------------------------------------------------------------
// The same step for callers that cannot allocate: the workspace (the partial sums of squares) lives in a file-scope
// device array instead of a caller buffer.  Every GPU that runs the library holds its
// own copy of the array at its own address until the process ends; `scratch` below is only the host-side handle of
// the symbol, never a device address: the address on a GPU comes from hipGetSymbolAddress, looked up once per device
// (with that device current) and kept.
// ------------------------------------------------------------------------------------------
static __device__ __align__(256) float scratch[DMV_OPT_STATIC_WS_FLOATS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_lock;
float* g_scratch_ptr[kMaxDevices];   // per device: where that GPU keeps its copy of `scratch` (null: not looked up yet)

// Device address of the current device's copy of `scratch`; > 0: a hipError_t, < 0: DMV_E_*.
int scratch_address(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return DMV_E_UNSUPPORTED;
  std::lock_guard<std::mutex> hold(g_scratch_lock);
  if (!g_scratch_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (!p) return DMV_E_BADARG;
    g_scratch_ptr[dev] = (float*)p;
  }
  *out = g_scratch_ptr[dev];
  return DMV_OK;
}
}  // namespace

extern "C" int dmv_opt_adam_step_static(int32_t n, float* const* params, const float* const* grads,
                                        float* const* exp_avg, float* const* exp_avg_sq, float* const* max_exp_avg_sq,
                                        const int64_t* numel, const int32_t* group_of, int32_t n_groups,
                                        const DmvAdamGroup* groups, float max_norm, float* norm_out,
                                        const DmvPackMap* pack_map, uint64_t* counter, uint64_t counter_add, void* stream) {
  const int rc = check_tables(n, params, grads, exp_avg, exp_avg_sq, numel, group_of, n_groups, groups);
  if (rc) return rc;
  const size_t need = dmv_opt_ws_floats(n, numel);
  // a size table dmv_opt_ws_floats cannot size (negative or > 2^31 elements) gets the code dmv_opt_adam_step gives it
  if (need == 0 || need > (size_t)DMV_OPT_STATIC_WS_FLOATS) return DMV_E_WORKSPACE;   // more parameters than the static array serves
  float* ws = nullptr;
  const int rs = scratch_address(&ws);
  if (rs) return rs;
  // nothing in the array has to start at zero: k_opt_sumsq writes every partial sum k_opt_apply reads, on the same
  // stream
  return adam_step(n, params, grads, exp_avg, exp_avg_sq, max_exp_avg_sq, numel, group_of, n_groups, groups, max_norm,
                   ws, norm_out, pack_map, counter, counter_add, stream);
}
