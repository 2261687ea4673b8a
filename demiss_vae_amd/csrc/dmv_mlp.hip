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
#include "dmv_rng.h"
#include "dmv_objmath.h"
#include "dmv_tail.h"

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// The build compiles this file three times (-DDMV_MLP_PART=1, 2, 3: three objects, in parallel), each time with the
// entry points of one group of kernels; a kernel template is only instantiated in the object whose entry points reach
// it.  1: the C entry points, the kernels of hidden widths up to 256 and the weight-gradient kernels; 2: hidden widths
// 257 - 512; 3: the one-launch encoder-decoder passes and sampler updates.  Without the macro (the tuning build) everything
// lands in one object.
#ifndef DMV_MLP_PART
#define DMV_MLP_PART 0
#endif
#define DMV_PART(p) (DMV_MLP_PART == 0 || DMV_MLP_PART == (p))

// Phase timestamps for kernel tuning (debug builds only: make TRACE=1): wave w writes the shader clock at
// phase i to dmv_trace_buf[16 w + i]; dmv_debug_trace copies the buffer out.
#ifdef DMV_TRACE
__device__ long long dmv_trace_buf[16 * 8192];
#define TRACE(i)                                                                                   \
  do {                                                                                             \
    const long long t_ = __builtin_readcyclecounter();                                             \
    const int w_ = blockIdx.x * 4 + (threadIdx.x >> 6);                                            \
    if ((threadIdx.x & 63) == 0 && w_ < 8192) dmv_trace_buf[16 * w_ + (i)] = t_;                   \
  } while (0)
#define TRACE_REAL(i)                                                                              \
  do {                                                                                             \
    const long long t_ = (long long)__builtin_amdgcn_s_memrealtime();                              \
    const int w_ = blockIdx.x * 4 + (threadIdx.x >> 6);                                            \
    if ((threadIdx.x & 63) == 0 && w_ < 8192) dmv_trace_buf[16 * w_ + (i)] = t_;                   \
  } while (0)
extern "C" int dmv_debug_trace(void* dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(dmv_trace_buf), bytes);
}
#else
#define TRACE(i)
#define TRACE_REAL(i)
#endif

// ------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------
struct PackArgs {
  DmvLayout L;
  const float* w[DMV_MAX_LAYERS];
  const float* b[DMV_MAX_LAYERS];
  float* packed;
};

// one thread = one lane slot of one fragment (16 bytes of an image) or one bias-fragment lane
#if DMV_PART(1)
__global__ void k_pack(PackArgs a) {
  const int l = blockIdx.y;
  const int N = a.L.N[l], K = a.L.K[l];
  const int nt = (N + 31) / 32, S = a.L.S[l];
  const int kt = (K + 31) / 32, Sn = a.L.Sn[l];
  const int64_t nfwd = (int64_t)nt * S * 64, ntr = (int64_t)kt * Sn * 64, nb = (int64_t)nt * 64;
  const float* W = a.w[l];
  const bool row4 = (K & 3) == 0 && ((uintptr_t)W & 15) == 0;      // four consecutive inputs of a row as one load
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nfwd + ntr + nb;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nfwd) {
      const int lane = i & 63;
      const int64_t g = i >> 6;
      const int s = g % S, t = g / S;
      const int n = 32 * t + (lane & 31), k0 = 8 * s + 4 * (lane >> 5);
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (n < N) {
        if (row4 && k0 + 4 <= K) o = *(const f32x4*)(W + (int64_t)n * K + k0);
        else
          for (int e = 0; e < 4; ++e) o[e] = (k0 + e < K) ? W[(int64_t)n * K + k0 + e] : 0.f;
      }
      *(f32x4*)(a.packed + a.L.w[l] + 4 * i) = o;
    } else if (i < nfwd + ntr) {
      const int64_t j = i - nfwd;
      const int lane = j & 63;
      const int64_t g = j >> 6;
      const int s = g % Sn, t = g / Sn;
      const int k = 32 * t + (lane & 31), n0 = 8 * s + 4 * (lane >> 5);
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (k < K)
        for (int e = 0; e < 4; ++e) o[e] = (n0 + e < N) ? W[(int64_t)(n0 + e) * K + k] : 0.f;
      *(f32x4*)(a.packed + a.L.wt[l] + 4 * j) = o;
    } else {
      // bias as an MFMA A-operand fragment: lane (n, h = 0) carries bias[32 t + n], the h = 1 half is zero
      const int64_t j = i - nfwd - ntr;
      const int lane = j & 63, n = 32 * (int)(j >> 6) + (lane & 31);
      a.packed[a.L.b[l] + j] = (lane < 32 && n < N) ? a.b[l][n] : 0.f;
    }
  }
}
#endif

// ------------------------------------------------------------------------------------------
// device building blocks
// ------------------------------------------------------------------------------------------
// Hidden-to-hidden GEMM: acc[t] (+)= sum over the 32*NT inputs held in `in`  (Wimg: fragment image with
// S = 4*NT groups).  Weight fragments are fetched PF groups ahead of their use (each group = NT coalesced
// 1 KiB loads feeding 4*NT MFMAs), so an L2 / Infinity-Cache round trip hides under the wave's own MFMAs;
// the first PF groups are requested by gemm_head, which the caller places ahead of the previous layer's
// epilogue.
//   FRESH: acc starts from zero (the first MFMA of each tile takes a zero C operand).
//   BIAS:  the layer's bias enters as one more rank-1 MFMA per tile (bias fragment x ones), so no bias
//          loads or adds sit between two GEMMs.
// An optional side stream rides on the same software pipeline, one 16-byte store per lane and group, so that
// no burst of memory traffic (and no wait for it) sits between two GEMMs:
//   ST:  `in` is written to st[8 s .. 8 s + 3] in group s (st = this lane's row + 4 h): group s multiplies
//        exactly the four registers that hold those features.  Rows are padded to a multiple of 32 in the
//        saved buffers, so the store needs no predicate.
//   MKW: (with ST) the sign bits of `in` are collected on the way and written to mk_dst when the GEMM ends:
//        bit 16 (t & 1) + r of word t >> 1 is set when element r of tile t is > 0.  The backward chain reads
//        these 4 NT bits per lane instead of the activations (relu' = sign bit of the output).
constexpr int mask_words(int NT) { return (NT + 1) / 2; }

// Fragment and activation-tile traffic goes through buffer instructions: a wave-uniform resource (base in
// scalar registers), one 32-bit per-lane offset that never changes, and the fragment / group offset as a scalar
// or immediate.  No per-access vector address arithmetic competes with the MFMAs for issue slots, and an access
// outside the resource is dropped by the hardware instead of faulting.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, off, 0));
}

__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, uint32_t lane_bytes, uint32_t off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, lane_bytes, off, 0);
}

// Where the keep decisions of the residual blocks' dropout layers come from: explicit bytes [nb][rows][hidden]
// (parity with recorded masks), or generated in the kernel from the Philox stream (training: no mask array is
// written or read at all).  Generated decisions depend only on (seed, first call, block, row, feature): every
// kernel family and dmv_mlp_dropout_keep produce the same ones.  A unit draws `bits` random bits (1 when the
// dropout probability is 1/2, 8 when it is a multiple of 1/256, else 16) and is kept when their value is at least
// `thresh` = round(p 2^bits).  The draws of one (block, row, half h = the lane's half of the 8-feature groups) are
// consecutive 128-bit generator calls: unit j = 4 q + e of 32-wide tile t (feature 32 t + 8 q + 4 h + e) owns bits
// [(16 t + j) bits, (16 t + j + 1) bits) of that run.
struct DropSrc {
  int32_t mode;            // 0 off, 1 explicit bytes, 2 generated
  int32_t bits;
  uint32_t thresh;
  const uint8_t* keep;
  uint64_t seed, q0;       // generated: Philox key, first call number
  const uint64_t* offset_base;   // generated, nullable: device counter (in elements of 32 bits) added to the start
};
enum { DROP_OFF = 0, DROP_BYTES = 1, DROP_GEN = 2 };
#define DMV_DROP_TAG 0x64726f70ull /* "drop" */

__host__ __device__ inline int drop_calls(int bits, int NT) { return bits == 1 ? (NT + 7) / 8 : bits == 8 ? NT : 2 * NT; }

__device__ __forceinline__ uint64_t drop_first_call(const DropSrc& d) {
  return d.q0 + (d.offset_base ? (*d.offset_base >> 2) : 0ull);
}

__device__ __forceinline__ uint32_t u4_word(const dmv_u4& r, int i) { return i == 0 ? r.x : i == 1 ? r.y : i == 2 ? r.z : r.w; }

// Keep bits of row rl, half h, block blk for the 32-wide tiles [tile_lo, tile_hi), in the layout of the sign-bit
// words: bit 16 (t & 1) + 4 q + e of kb[t >> 1].  Tiles outside the range may come back set or clear.
template <int NT>
__device__ __forceinline__ void drop_words(const DropSrc& d, uint64_t q0, int64_t rows, int hidden, int blk, int64_t rl, int h,
                                           int tile_lo, int tile_hi, uint32_t (&kb)[mask_words(NT)]) {
#pragma unroll
  for (int i = 0; i < mask_words(NT); ++i) kb[i] = 0u;
  if (d.mode == DROP_BYTES) {
    const uint8_t* k = d.keep + ((int64_t)blk * rows + rl) * hidden + 4 * h;
    const bool dwords = (hidden & 3) == 0 && ((uintptr_t)d.keep & 3) == 0;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t >= tile_lo && t < tile_hi) {
        uint32_t b16 = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = 32 * t + 8 * q + 4 * h;
          if (dwords) {
            const uint32_t w = n < hidden ? *(const uint32_t*)(k + 32 * t + 8 * q) : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) b16 |= (((w >> (8 * e)) & 0xffu) ? 1u : 0u) << (4 * q + e);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) b16 |= ((n + e < hidden && k[32 * t + 8 * q + e]) ? 1u : 0u) << (4 * q + e);
          }
        }
        kb[t >> 1] |= b16 << (16 * (t & 1));
      }
    return;
  }
  const uint64_t qb = q0 + (((uint64_t)blk * (uint64_t)rows + (uint64_t)rl) * 2u + (uint64_t)h) * (uint64_t)drop_calls(d.bits, NT);
  if (d.bits == 1) {
#pragma unroll
    for (int c = 0; c < (NT + 7) / 8; ++c) {
      const dmv_u4 r = dmv_philox4x32_10(d.seed, qb + c, DMV_DROP_TAG);
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (4 * c + w < mask_words(NT)) kb[4 * c + w] = u4_word(r, w);
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t >= tile_lo && t < tile_hi) {
      uint32_t b16 = 0u;
      if (d.bits == 8) {
        const dmv_u4 r = dmv_philox4x32_10(d.seed, qb + t, DMV_DROP_TAG);
#pragma unroll
        for (int j = 0; j < 16; ++j) b16 |= (((u4_word(r, j >> 2) >> (8 * (j & 3))) & 0xffu) >= d.thresh ? 1u : 0u) << j;
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const dmv_u4 r = dmv_philox4x32_10(d.seed, qb + 2 * t + c, DMV_DROP_TAG);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            b16 |= (((u4_word(r, j >> 1) >> (16 * (j & 1))) & 0xffffu) >= d.thresh ? 1u : 0u) << (8 * c + j);
        }
      }
      kb[t >> 1] |= b16 << (16 * (t & 1));
    }
}

// halfword `tile` of a keep-bit / sign-bit word array, for a tile number known only at run time (wave-uniform)
template <int MW>
__device__ __forceinline__ uint32_t half_of(const uint32_t (&kb)[MW], int tile) {
  uint32_t w = 0u;
#pragma unroll
  for (int i = 0; i < MW; ++i) w = (tile >> 1) == i ? kb[i] : w;
  return (w >> (16 * (tile & 1))) & 0xffffu;
}

// fragment `frag` (64 lanes x 16 bytes) of a weight image
__device__ __forceinline__ f32x4 frag_load(__amdgpu_buffer_rsrc_t img, int frag, uint32_t lane_bytes) {
  return buf_load4(img, lane_bytes, (uint32_t)frag * 1024u);
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

template <int NT, int PF, bool BIAS>
__device__ __forceinline__ void gemm_head(const f32x4* __restrict__ Wimg, const float* __restrict__ Bimg, int lane,
                                          f32x4 (&w)[PF + 1][NT], float (&bf)[NT]) {
  constexpr int S = 4 * NT;
  const uint32_t lb = (uint32_t)lane * 16u;
  const __amdgpu_buffer_rsrc_t img = make_rsrc(Wimg, NT * S * 1024u);
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int t = 0; t < NT; ++t) w[p][t] = frag_load(img, t * S + p, lb);
  if (BIAS) {
    const __amdgpu_buffer_rsrc_t bimg = make_rsrc(Bimg, NT * 256u);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      bf[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bimg, (uint32_t)lane * 4u + t * 256u, 0, 0));
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int NT>
__device__ __forceinline__ void store_mask(uint32_t* __restrict__ dst, const uint32_t (&mk)[mask_words(NT)]) {
  if (mask_words(NT) == 4) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    *(u32x4*)dst = u32x4{mk[0], mk[1], mk[2], mk[3]};
  } else {
#pragma unroll
    for (int i = 0; i < mask_words(NT); ++i) dst[i] = mk[i];
  }
}

template <int NT>
__device__ __forceinline__ void load_mask(const uint32_t* __restrict__ src, uint32_t (&mk)[mask_words(NT)]) {
  if (mask_words(NT) == 4) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(const u32x4*)src;
#pragma unroll
    for (int i = 0; i < 4; ++i) mk[i] = v[i];
  } else {
#pragma unroll
    for (int i = 0; i < mask_words(NT); ++i) mk[i] = src[i];
  }
}

// sign bits of group s of `in`, gemm_hh's MKW layout
template <int NT>
__device__ __forceinline__ void collect_mask(const f32x16 (&in)[NT], int s, uint32_t (&mk)[mask_words(NT)]) {
  uint32_t bits = 0u;
#pragma unroll
  for (int e = 0; e < 4; ++e) bits |= (in[s >> 2][4 * (s & 3) + e] > 0.f ? 1u : 0u) << (16 * ((s >> 2) & 1) + 4 * (s & 3) + e);
  if ((s & 7) == 0) mk[s >> 3] = bits;
  else mk[s >> 3] |= bits;
  // pin the few compare / select instructions to this group (they issue in the shadow of its MFMAs); left alone,
  // the optimiser gathers all 32 NT of them after the last group, where nothing hides them
  asm volatile("" : "+v"(mk[s >> 3]));
}

// What a GEMM does to its input registers on the way in (in place, four registers per group, in the shadow of
// the previous group's MFMAs), so that no elementwise pass over the whole state sits between two GEMMs:
//   IN_MASK:      v = sign bit set ? scale * v : 0           (relu backward, dropout scale folded in)
//   IN_RELU_DROP: v = keep bit set ? scale * max(v, 0) : 0  (relu then inverted dropout; the bits as drop_words lays them out)
enum { IN_ASIS = 0, IN_RELU = 1, IN_MASK = 2, IN_RELU_DROP = 3 };

template <int INOP, int NT>
__device__ __forceinline__ void in_op(f32x16 (&in)[NT], int s, const uint32_t (&mkin)[mask_words(NT)], float scale) {
  if (INOP == IN_ASIS) return;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v = in[s >> 2][4 * (s & 3) + e];
    if (INOP == IN_RELU) {
      v = __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff());   // max(v, 0) in one instruction (NaN -> 0, like fmaxf)
    } else if (INOP == IN_RELU_DROP) {
      const bool on = (mkin[s >> 3] >> (16 * ((s >> 2) & 1) + 4 * (s & 3) + e)) & 1u;
      v = on ? scale * __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : 0.f;
    } else {
      const bool on = (mkin[s >> 3] >> (16 * ((s >> 2) & 1) + 4 * (s & 3) + e)) & 1u;
      v = on ? v * scale : 0.f;
    }
    in[s >> 2][4 * (s & 3) + e] = v;
  }
}

template <int NT, int PF, bool FRESH, bool BIAS, bool ST, bool MKW, int INOP>
__device__ __forceinline__ void gemm_hh(const f32x4* __restrict__ Wimg, int lane, f32x16 (&in)[NT],
                                        f32x16 (&acc)[NT], f32x4 (&w)[PF + 1][NT], const float (&bf)[NT],
                                        float* __restrict__ st_tile, uint32_t st_lane, uint32_t* __restrict__ mk_dst,
                                        const uint32_t (&mkin)[mask_words(NT)], float scale) {
  constexpr int S = 4 * NT;
  const uint32_t lb = (uint32_t)lane * 16u;
  const __amdgpu_buffer_rsrc_t img = make_rsrc(Wimg, NT * S * 1024u);
  const __amdgpu_buffer_rsrc_t out = make_rsrc(st_tile, ST ? 32u * 32u * NT * 4u : 0u);
  uint32_t mk[mask_words(NT)];
  if (BIAS) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = MFMA(bf[t], 1.f, FRESH ? zero16() : acc[t]);
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (s + PF < S) {
#pragma unroll
      for (int t = 0; t < NT; ++t) w[(s + PF) % (PF + 1)][t] = frag_load(img, t * S + s + PF, lb);
    }
    in_op<INOP, NT>(in, s, mkin, scale);
    if (ST) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = in[s >> 2][4 * (s & 3) + e];
      buf_store4(out, st_lane, 32u * s, o);
    }
    // keep the memory operations of this group ahead of its MFMAs (the scheduler otherwise sinks them next to
    // their first use, which exposes the whole L2 latency)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float b = in[s >> 2][4 * (s & 3) + e];
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = MFMA(w[s % (PF + 1)][t][e], b, (FRESH && !BIAS && s == 0 && e == 0) ? zero16() : acc[t]);
    }
    if (MKW) collect_mask<NT>(in, s, mk);
  }
  if (MKW) store_mask<NT>(mk_dst, mk);
}

// One 32-wide output tile from the hidden state: acc += W_tile in, optionally storing `in` on the way (see
// gemm_hh).  One accumulator, so a step is only 4 MFMAs: fragments are requested 8 steps ahead.
template <int NT, bool ST, bool MKW, int INOP>
__device__ __forceinline__ void gemm_h1(const f32x4* __restrict__ Wtile, int lane, f32x16 (&in)[NT], f32x16& acc,
                                        float* __restrict__ st_tile, uint32_t st_lane, uint32_t* __restrict__ mk_dst,
                                        const uint32_t (&mkin)[mask_words(NT)], float scale) {
  constexpr int S = 4 * NT;
  constexpr int P1 = S < 8 ? S : 8;
  const uint32_t lb = (uint32_t)lane * 16u;
  const __amdgpu_buffer_rsrc_t img = make_rsrc(Wtile, S * 1024u);
  const __amdgpu_buffer_rsrc_t out = make_rsrc(st_tile, ST ? 32u * 32u * NT * 4u : 0u);
  uint32_t mk[mask_words(NT)];
  f32x4 w[P1 + 1];
#pragma unroll
  for (int p = 0; p < P1; ++p) w[p] = frag_load(img, p, lb);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (s + P1 < S) w[(s + P1) % (P1 + 1)] = frag_load(img, s + P1, lb);
    in_op<INOP, NT>(in, s, mkin, scale);
    if (ST) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = in[s >> 2][4 * (s & 3) + e];
      buf_store4(out, st_lane, 32u * s, o);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = MFMA(w[s % (P1 + 1)][e], in[s >> 2][4 * (s & 3) + e], acc);
    if (MKW) collect_mask<NT>(in, s, mk);
  }
  if (MKW) store_mask<NT>(mk_dst, mk);
}

// Network-input (or output-gradient) GEMM: acc[t] += W[:, 8 s ..] b(s) over a run-time number of groups, the B
// operand of group s coming from load_b(s) (4 values of this lane's row).  Next group's fragments and operand
// are requested before the current group's MFMAs (the index is clamped, so the last group is fetched twice
// rather than branching).
template <int NT, class LoadB>
__device__ __forceinline__ void gemm_in(const f32x4* __restrict__ W, int S, int lane, f32x16 (&acc)[NT], LoadB load_b) {
  const uint32_t lb = (uint32_t)lane * 16u;
  const __amdgpu_buffer_rsrc_t img = make_rsrc(W, (uint32_t)(NT * S) * 1024u);
  f32x4 wn[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wn[t] = frag_load(img, t * S, lb);
  f32x4 bn = load_b(0);
  for (int s = 0; s < S; ++s) {
    f32x4 w[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) w[t] = wn[t];
    const f32x4 b = bn;
    const int s1 = s + 1 < S ? s + 1 : S - 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) wn[t] = frag_load(img, t * S + s1, lb);
    bn = load_b(s1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = MFMA(w[t][e], b[e], acc[t]);
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

// relu (every shipped config) is compiled into its own kernels; the other activations share a generic,
// rolled path (act is a run-time value there)
template <int NT, bool RELU>
__device__ __forceinline__ void apply_act(f32x16 (&v)[NT], int act) {
  if (RELU) {
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
  float drop_scale;  // 1/(1-p), used only with dropout on
  int in_mode, D;
  const float* x;
  int64_t ldx;
  const uint8_t* m;
  int64_t ldm;
  int64_t rows;
  int64_t row_begin, row_end;   // this launch covers rows [row_begin, row_end) (row_begin a multiple of 32)
  float* y;
  int64_t ldy;
  float* xs;     // saved first-layer input [rows][32*T_in] (nullable)
  float* acts;   // saved activations [(2nb+1)][rows][32*NT] (nullable)
  uint32_t* masks;  // sign bits of the saved activations [(2nb+1)][rows][2][mask_words] (with acts)
  DropSrc drop;
};

// Four consecutive first-layer inputs (columns k0 .. k0+3) of row r, branch-free: loads are unconditional with
// the column clamped into the row, the input mode only selects between the loaded values.
__device__ __forceinline__ f32x4 first_layer_input4_of(const FwdArgs& a, const float* xr, const uint8_t* mr, int k0) {
  const bool use_mask = a.in_mode == DMV_IN_MASKED || a.in_mode == DMV_IN_CONCAT_MASKED;
  f32x4 out;
  if (a.in_mode == DMV_IN_PLAIN) {   // uniform
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + e;
      const float xv = xr[k < a.D ? k : a.D - 1];
      out[e] = k < a.in_dim ? xv : 0.f;
    }
    return out;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = k0 + e;
    int kc = k < a.D ? k : k - a.D;
    kc = kc < a.D ? kc : a.D - 1;
    const float xv = xr[kc], mv = (float)mr[kc];
    const float head = use_mask ? xv * mv : xv;
    out[e] = k < a.in_dim ? (k < a.D ? head : mv) : 0.f;
  }
  return out;
}

__device__ __forceinline__ f32x4 first_layer_input4(const FwdArgs& a, int64_t r, int k0) {
  return first_layer_input4_of(a, a.x + r * a.ldx, a.in_mode == DMV_IN_PLAIN ? nullptr : a.m + r * a.ldm, k0);
}

// SAVE: training forward (stores the first-layer input and every hidden activation for the backward;
// the saved buffers hold rows rounded up to a multiple of 32, so whole-wave stores need no predicate).
// A hidden activation is written while the next GEMM consumes it (gemm_hh / gemm_h1, ST).
// MODE: relu networks (every shipped configuration) run specialised kernels, without (FWD_RELU) or with
// (FWD_RELU_DROP) dropout inside the residual blocks; FWD_GENERIC takes the activation at run time.
enum { FWD_GENERIC = 0, FWD_RELU = 1, FWD_RELU_DROP = 2 };

template <int NT, bool SAVE, int MODE>
__global__ __launch_bounds__(256, 1) void k_mlp_fwd(FwdArgs a) {
  constexpr bool FAST = MODE != FWD_GENERIC;
  constexpr int PF = 1;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = a.row_begin + ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 32;
  if (row0 >= a.row_end) return;
  const int m = lane & 31, h = lane >> 5;
  const int64_t row = row0 + m;
  const bool live = row < a.rows;
  const int64_t rl = live ? row : a.rows - 1;
  constexpr int Hp = 32 * NT;
  const int64_t act_stride = ((a.rows + 31) & ~(int64_t)31) * Hp;
  constexpr int MW = mask_words(NT);
  const int64_t mask_stride = ((a.rows + 31) & ~(int64_t)31) * 2 * MW;
  float* st = SAVE ? a.acts + row0 * Hp : nullptr;          // this wave's 32-row tile of the saved activations, layer 0
  const uint32_t stl = (uint32_t)(m * Hp + 4 * h) * 4u;     // this lane's byte offset inside a tile
  uint32_t* mkd = SAVE ? a.masks + (row * 2 + h) * MW : nullptr;   // ... and in their sign bits

  f32x16 A[NT], T[NT];
  f32x4 w[PF + 1][NT];   // weight ring of the hidden GEMMs
  float bf[NT];          // bias fragments of the next hidden GEMM
  const uint64_t drop_q0 = (MODE == FWD_RELU_DROP && a.drop.mode == DROP_GEN) ? drop_first_call(a.drop) : 0ull;

  TRACE(0);
  TRACE_REAL(13);
  // ---- initial layer (the first block's weights are requested first: they arrive under the input loads)
  if (a.nb > 0) gemm_head<NT, PF, true>((const f32x4*)(a.packed + a.L.w[1]), a.packed + a.L.b[1], lane, w, bf);
  {
    const float* B0 = a.packed + a.L.b[0];
#pragma unroll
    for (int t = 0; t < NT; ++t) A[t] = MFMA(B0[t * 64 + lane], 1.f, zero16());
    float* xs_row = SAVE ? a.xs + row * (32 * a.L.T_in) + 4 * h : nullptr;
    gemm_in<NT>((const f32x4*)(a.packed + a.L.w[0]), a.L.S_in, lane, A, [&](int s) {
      const f32x4 b = first_layer_input4(a, rl, 8 * s + 4 * h);
      if (SAVE) *(f32x4*)(xs_row + 8 * s) = b;
      return b;
    });
  }
  TRACE(1);
  // FAST: relu is applied by the GEMM that consumes a state (IN_RELU), in place
  constexpr int INR = FAST ? IN_RELU : IN_ASIS;
  uint32_t nomask[mask_words(NT)];
  if (!FAST) apply_act<NT, false>(A, a.act);
  TRACE(2);

  // ---- residual blocks
  for (int blk = 0; blk < a.nb; ++blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    gemm_hh<NT, PF, true, true, SAVE, SAVE, INR>((const f32x4*)(a.packed + a.L.w[l1]), lane, A, T, w, bf,
                                                 st + (l1 - 1) * act_stride, stl, mkd + (l1 - 1) * mask_stride, nomask, 1.f);
    TRACE(3 + 4 * blk);
    gemm_head<NT, PF, true>((const f32x4*)(a.packed + a.L.w[l2]), a.packed + a.L.b[l2], lane, w, bf);
    uint32_t kb[mask_words(NT)];
    if (MODE == FWD_RELU_DROP) drop_words<NT>(a.drop, drop_q0, a.rows, a.hidden, blk, rl, h, 0, NT, kb);
    if (!FAST) apply_act<NT, false>(T, a.act);
    TRACE(4 + 4 * blk);
    if (!FAST && a.drop.mode == DROP_BYTES) {
      const uint8_t* keep = a.drop.keep + ((int64_t)blk * a.rows + rl) * a.hidden;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = 32 * t + 8 * (r >> 2) + 4 * h + (r & 3);
          const float kf = (n < a.hidden && keep[n]) ? a.drop_scale : 0.f;
          T[t][r] *= kf;
        }
    }
    if (MODE == FWD_RELU_DROP)
      gemm_hh<NT, PF, false, true, SAVE, SAVE, IN_RELU_DROP>((const f32x4*)(a.packed + a.L.w[l2]), lane, T, A, w, bf,
                                                             st + l1 * act_stride, stl, mkd + l1 * mask_stride, kb, a.drop_scale);
    else
      gemm_hh<NT, PF, false, true, SAVE, SAVE, INR>((const f32x4*)(a.packed + a.L.w[l2]), lane, T, A, w, bf,
                                                    st + l1 * act_stride, stl, mkd + l1 * mask_stride, nomask, 1.f);
    TRACE(5 + 4 * blk);
    if (blk + 1 < a.nb)
      gemm_head<NT, PF, true>((const f32x4*)(a.packed + a.L.w[l2 + 1]), a.packed + a.L.b[l2 + 1], lane, w, bf);
    if (!FAST) apply_act<NT, false>(A, a.act);
    TRACE(6 + 4 * blk);
  }

  // ---- final layer: one 32-wide output tile at a time
  {
    const int lf = a.L.nl - 1;
    const f32x4* W = (const f32x4*)(a.packed + a.L.w[lf]);
    const float* Bf = a.packed + a.L.b[lf];
    constexpr int S = 4 * NT;
    for (int ot = 0; ot < a.L.OT; ++ot) {
      f32x16 acc = MFMA(Bf[ot * 64 + lane], 1.f, zero16());
      if (SAVE && ot == 0)
        gemm_h1<NT, true, true, INR>(W, lane, A, acc, st + (lf - 1) * act_stride, stl, mkd + (lf - 1) * mask_stride,
                                     nomask, 1.f);
      else   // (relu is idempotent: a second pass over A leaves it as it is)
        gemm_h1<NT, false, false, INR>(W + (int64_t)ot * S * 64, lane, A, acc, nullptr, 0u, nullptr, nomask, 1.f);
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
  TRACE(15);
  TRACE_REAL(14);
}

// ------------------------------------------------------------------------------------------
// Small batches: one workgroup (4 waves) per 32-row tile.
//
// With one wave per tile a network pass is a serial chain of 2 nb + 2 GEMMs (about 130 us at hidden 256) and a
// batch of a few thousand rows leaves most SIMDs idle.  Here the four waves of a workgroup share a tile: every
// wave holds the whole input state of a layer (B operands) but computes only TW = ceil(NT / 4) of its NT output
// tiles, so a layer issues a quarter of the MFMAs per wave; the output slices are exchanged through LDS
// ([tile][register quad][lane] float4: conflict-free) with one barrier pair per layer.  Same accumulation order per
// output as the one-wave kernels (bit-identical results), same saved-activation / sign-bit layout.
// relu networks only (with or without dropout); other activations take the one-wave kernels.
// ------------------------------------------------------------------------------------------
// (plan_row_segments decides which rows of a batch these kernels take)
template <int NT>
struct Coop {
  static constexpr int TW = (NT + 3) / 4;
  static constexpr int LDS_BYTES = NT * 4 * 64 * 16;
};

// acc[i] (+)= W[tile t0 + i] in, i < TW (tiles past NT are skipped); fragments one group ahead
template <int NT, bool FRESH, bool BIAS>
__device__ __forceinline__ void coop_gemm(const f32x4* __restrict__ Wimg, const float* __restrict__ Bimg, int lane, int t0,
                                          const f32x16 (&in)[NT], f32x16 (&acc)[Coop<NT>::TW]) {
  constexpr int S = 4 * NT, TW = Coop<NT>::TW;
  const uint32_t lb = (uint32_t)lane * 16u;
  const __amdgpu_buffer_rsrc_t img = make_rsrc(Wimg, NT * S * 1024u);
  f32x4 w[2][TW];
  const uint32_t tile_bytes = (uint32_t)S * 1024u;
  // tiles past NT read outside the resource and come back as zeros: their MFMAs are harmless, never stored
#pragma unroll
  for (int i = 0; i < TW; ++i) w[0][i] = buf_load4(img, lb, (uint32_t)(t0 + i) * tile_bytes);
  if (BIAS) {
    const __amdgpu_buffer_rsrc_t bimg = make_rsrc(Bimg, NT * 256u);
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      const float bf = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bimg, (uint32_t)lane * 4u, (uint32_t)(t0 + i) * 256u, 0));
      acc[i] = MFMA(bf, 1.f, FRESH ? zero16() : acc[i]);
    }
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (s + 1 < S) {
#pragma unroll
      for (int i = 0; i < TW; ++i) w[(s + 1) & 1][i] = buf_load4(img, lb, (uint32_t)(t0 + i) * tile_bytes + (uint32_t)(s + 1) * 1024u);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float b = in[s >> 2][4 * (s & 3) + e];
#pragma unroll
      for (int i = 0; i < TW; ++i)
        acc[i] = MFMA(w[s & 1][i][e], b, (FRESH && !BIAS && s == 0 && e == 0) ? zero16() : acc[i]);
    }
  }
}

// all-gather of the waves' output slices: afterwards every wave holds all NT tiles in `full`
template <int NT>
__device__ __forceinline__ void coop_exchange(float* lds, int lane, int t0, const f32x16 (&slice)[Coop<NT>::TW], f32x16 (&full)[NT]) {
  constexpr int TW = Coop<NT>::TW;
  f32x4* L4 = (f32x4*)lds;
  __syncthreads();                                   // previous readers are done with the buffer
#pragma unroll
  for (int i = 0; i < TW; ++i)
    if (t0 + i < NT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = slice[i][4 * q + e];
        L4[((t0 + i) * 4 + q) * 64 + lane] = o;
      }
    }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = L4[(t * 4 + q) * 64 + lane];
#pragma unroll
      for (int e = 0; e < 4; ++e) full[t][4 * q + e] = v[e];
    }
}

// slice[i] = full[wv * TW + i]: the tile number is wave-uniform but not a compile-time constant, and register
// arrays cannot be indexed at run time, so the wave's case is picked from an unrolled loop over the four waves
template <int NT>
__device__ __forceinline__ void coop_own_slice(const f32x16 (&full)[NT], int wv, f32x16 (&slice)[Coop<NT>::TW]) {
  constexpr int TW = Coop<NT>::TW;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww)
    if (ww == wv) {
#pragma unroll
      for (int i = 0; i < TW; ++i)
        if (ww * TW + i < NT) slice[i] = full[ww * TW + i < NT ? ww * TW + i : 0];
    }
}

// relu on a slice (DROP: then inverted dropout, keep bits as drop_words lays them out); with SAVE also the slice's
// rows of the saved activation and its sign-bit word
template <int NT, bool SAVE, bool DROP = false>
__device__ __forceinline__ void coop_relu_save(f32x16 (&slice)[Coop<NT>::TW], int t0, float* act_row, uint32_t* mask_row,
                                               const uint32_t* kb = nullptr, float scale = 1.f) {
  constexpr int TW = Coop<NT>::TW;
#pragma unroll
  for (int i = 0; i < TW; ++i) {
    uint32_t bits = 0u;
    uint32_t keep16 = 0xffffu;
    if (DROP) {
      uint32_t kw[mask_words(NT)];
#pragma unroll
      for (int j = 0; j < mask_words(NT); ++j) kw[j] = kb[j];
      keep16 = half_of<mask_words(NT)>(kw, t0 + i < NT ? t0 + i : 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      slice[i][r] = __builtin_amdgcn_fmed3f(slice[i][r], 0.f, __builtin_inff());
      if (DROP) slice[i][r] = ((keep16 >> r) & 1u) ? scale * slice[i][r] : 0.f;
      bits |= (slice[i][r] > 0.f ? 1u : 0u) << r;
    }
    if (SAVE && t0 + i < NT) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = slice[i][4 * q + e];
        *(f32x4*)(act_row + 32 * (t0 + i) + 8 * q) = o;
      }
      // word (t >> 1) holds tiles 2 (t >> 1) and 2 (t >> 1) + 1 in its low / high half
      uint16_t* half = (uint16_t*)(mask_row + ((t0 + i) >> 1)) + ((t0 + i) & 1);
      *half = (uint16_t)bits;
    }
  }
}

template <int NT, bool SAVE, bool DROP>
__global__ __launch_bounds__(256, 1) void k_mlp_fwd_coop(FwdArgs a) {
  constexpr int TW = Coop<NT>::TW, Hp = 32 * NT, MW = mask_words(NT);
  __shared__ __attribute__((aligned(16))) float lds[Coop<NT>::LDS_BYTES / 4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t0 = wv * TW;
  const int64_t row0 = a.row_begin + (int64_t)blockIdx.x * 32;
  const int m = lane & 31, h = lane >> 5;
  const int64_t row = row0 + m;
  const bool live = row < a.rows;
  const int64_t rl = live ? row : a.rows - 1;
  const int64_t rp = (a.rows + 31) & ~(int64_t)31;
  const int64_t act_stride = rp * Hp, mask_stride = rp * 2 * MW;
  float* act_row = SAVE ? a.acts + row * Hp + 4 * h : nullptr;              // layer 0; + l * act_stride
  uint32_t* mask_row = SAVE ? a.masks + (row * 2 + h) * MW : nullptr;       // layer 0; + l * mask_stride

  // F: the layer input every wave holds in full; res: this wave's slice of the residual stream, which is all of the
  // stream a wave needs while F holds the block's inner activation (at hidden 512 two full states would not fit 512 registers)
  f32x16 F[NT], res[TW], acc[TW];
  const uint64_t drop_q0 = (DROP && a.drop.mode == DROP_GEN) ? drop_first_call(a.drop) : 0ull;

  // ---- initial layer: this wave's output tiles from the (tiny) network input
  {
    const float* B0 = a.packed + a.L.b[0];
    const f32x4* W = (const f32x4*)(a.packed + a.L.w[0]);
    const int S = a.L.S_in;
    const __amdgpu_buffer_rsrc_t img = make_rsrc(W, (uint32_t)(NT * S) * 1024u);
    const __amdgpu_buffer_rsrc_t bimg = make_rsrc(B0, NT * 256u);
#pragma unroll
    for (int i = 0; i < TW; ++i)
      acc[i] = MFMA(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bimg, (uint32_t)lane * 4u, (uint32_t)(t0 + i) * 256u, 0)),
                    1.f, zero16());
    float* xs_row = SAVE ? a.xs + row * (32 * a.L.T_in) + 4 * h : nullptr;
    for (int s = 0; s < S; ++s) {
      const f32x4 b = first_layer_input4(a, rl, 8 * s + 4 * h);
      if (SAVE && wv == 0) *(f32x4*)(xs_row + 8 * s) = b;
      f32x4 w[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) w[i] = buf_load4(img, (uint32_t)lane * 16u, (uint32_t)((t0 + i) * S + s) * 1024u);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TW; ++i) acc[i] = MFMA(w[i][e], b[e], acc[i]);
    }
  }
  coop_relu_save<NT, SAVE>(acc, t0, act_row, mask_row);
#pragma unroll
  for (int i = 0; i < TW; ++i) res[i] = acc[i];
  coop_exchange<NT>(lds, lane, t0, acc, F);

  // ---- residual blocks
  for (int blk = 0; blk < a.nb; ++blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    uint32_t kb[MW];
    if (DROP) drop_words<NT>(a.drop, drop_q0, a.rows, a.hidden, blk, rl, h, t0, t0 + TW, kb);   // (independent of the GEMM)
    coop_gemm<NT, true, true>((const f32x4*)(a.packed + a.L.w[l1]), a.packed + a.L.b[l1], lane, t0, F, acc);
    coop_relu_save<NT, SAVE, DROP>(acc, t0, act_row + l1 * act_stride, mask_row + l1 * mask_stride, kb, a.drop_scale);
    coop_exchange<NT>(lds, lane, t0, acc, F);
    // this wave's slice of the residual stream is the accumulator of the second GEMM
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = res[i];
    coop_gemm<NT, false, true>((const f32x4*)(a.packed + a.L.w[l2]), a.packed + a.L.b[l2], lane, t0, F, acc);
    coop_relu_save<NT, SAVE>(acc, t0, act_row + l2 * act_stride, mask_row + l2 * mask_stride);
#pragma unroll
    for (int i = 0; i < TW; ++i) res[i] = acc[i];
    coop_exchange<NT>(lds, lane, t0, acc, F);
  }

  // ---- final layer: output tiles round-robin over the waves
  {
    const int lf = a.L.nl - 1;
    const f32x4* W = (const f32x4*)(a.packed + a.L.w[lf]);
    const float* Bf = a.packed + a.L.b[lf];
    constexpr int S = 4 * NT;
    uint32_t nomask[MW];
    for (int ot = wv; ot < a.L.OT; ot += 4) {
      f32x16 o16 = MFMA(Bf[ot * 64 + lane], 1.f, zero16());
      gemm_h1<NT, false, false, IN_ASIS>(W + (int64_t)ot * S * 64, lane, F, o16, nullptr, 0u, nullptr, nomask, 1.f);
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = o16[4 * q + e];
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
  const uint32_t* masks;     // sign bits of acts, [(2nb+1)][rows][2][mask_words]
  const uint8_t* drop_keep;  // nullable
  const float* gy;
  int64_t ldgy;
  int64_t rows;
  int64_t row_begin, row_end;   // this launch covers rows [row_begin, row_end) (row_begin a multiple of 32)
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
// dropped or inactive unit saved 0).  The forward left those bits packed per lane (gemm_hh, MKW): one 16-byte
// load per layer, requested before the preceding GEMM.  Every layer gradient is written while the following
// GEMM consumes it (ST), so the only memory waits left between GEMMs are the ones at the two ends of the chain.
template <int NT, bool RELU>
__global__ __launch_bounds__(256, 1) void k_mlp_bwd(BwdArgs a) {
  constexpr int PF = 1;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = a.row_begin + ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 32;
  if (row0 >= a.row_end) return;
  const int m = lane & 31, h = lane >> 5;
  const int64_t row = row0 + m;
  const bool live = row < a.rows;
  const int64_t rl = live ? row : a.rows - 1;
  constexpr int Hp = 32 * NT;
  const int64_t act_stride = ((a.rows + 31) & ~(int64_t)31) * Hp;
  const int lf = a.L.nl - 1;
  float* st = a.dacts + row0 * Hp;                    // this wave's 32-row tile of the layer-gradient buffer, layer 0
  const uint32_t stl = (uint32_t)(m * Hp + 4 * h) * 4u;   // this lane's byte offset inside a tile
  constexpr int MW = mask_words(NT);
  const int64_t mask_stride = ((a.rows + 31) & ~(int64_t)31) * 2 * MW;
  const uint32_t* mks = a.masks + (row * 2 + h) * MW;  // ... and in the sign bits of the saved activations
  const float dscale = a.drop_keep ? a.drop_scale : 1.f;
  uint32_t mkA[mask_words(NT)], mkT[mask_words(NT)];
  constexpr int INM = RELU ? IN_MASK : IN_ASIS;   // relu backward is applied by the GEMM that consumes a gradient

  TRACE(0);
  TRACE_REAL(13);
  f32x16 dA[NT], dT[NT];
  f32x4 w[PF + 1][NT];
  float nobias[NT];
  if (a.nb > 0) gemm_head<NT, PF, false>((const f32x4*)(a.packed + a.L.wt[2 * a.nb]), nullptr, lane, w, nobias);
  if (RELU) load_mask<NT>(mks + (lf - 1) * mask_stride, mkA);
  // dA = Wf^T gy
  zero<NT>(dA);
  {
    const float* gyr = a.gy + rl * a.ldgy + 4 * h;
    gemm_in<NT>((const f32x4*)(a.packed + a.L.wt[lf]), a.L.S_out, lane, dA, [&](int s) {
      const f32x4 b = *(const f32x4*)(gyr + 8 * s);
      return live ? b : f32x4{0.f, 0.f, 0.f, 0.f};
    });
  }
  TRACE(1);
  for (int blk = a.nb - 1; blk >= 0; --blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    const int tb_ = 4 * (a.nb - 1 - blk); (void)tb_;
    // through the block's output activation: dA becomes the gradient of (a_in + W2 t + b2)
    if (!RELU) mul_act_grad<NT>(dA, a.acts + l2 * act_stride + rl * Hp, h, a.act);
    TRACE(2 + tb_);
    if (RELU) load_mask<NT>(mks + l1 * mask_stride, mkT);   // requested here, needed after the GEMM
    gemm_hh<NT, PF, true, false, true, false, INM>((const f32x4*)(a.packed + a.L.wt[l2]), lane, dA, dT, w, nobias,
                                                   st + l2 * act_stride, stl, nullptr, mkA, 1.f);
    TRACE(3 + tb_);
    gemm_head<NT, PF, false>((const f32x4*)(a.packed + a.L.wt[l1]), nullptr, lane, w, nobias);
    // through dropout and the inner activation
    if (RELU) {
      // applied by the next GEMM on its way in
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
    TRACE(4 + tb_);
    // dA (skip path) += W1^T dT
    if (RELU) load_mask<NT>(mks + (l1 - 1) * mask_stride, mkA);
    gemm_hh<NT, PF, false, false, true, false, INM>((const f32x4*)(a.packed + a.L.wt[l1]), lane, dT, dA, w, nobias,
                                                    st + l1 * act_stride, stl, nullptr, mkT, dscale);
    TRACE(5 + tb_);
    if (blk > 0) gemm_head<NT, PF, false>((const f32x4*)(a.packed + a.L.wt[l1 - 1]), nullptr, lane, w, nobias);
  }
  if (!RELU) mul_act_grad<NT>(dA, a.acts + rl * Hp, h, a.act);

  TRACE(10);
  if (a.dx) {
    const f32x4* W = (const f32x4*)(a.packed + a.L.wt[0]);
    constexpr int S = 4 * NT;
    for (int it = 0; it < a.L.T_in; ++it) {
      f32x16 acc = zero16();
      if (it == 0)
        gemm_h1<NT, true, false, INM>(W, lane, dA, acc, st, stl, nullptr, mkA, 1.f);
      else   // dA is already masked in place by the first pass
        gemm_h1<NT, false, false, IN_ASIS>(W + (int64_t)it * S * 64, lane, dA, acc, nullptr, 0u, nullptr, mkA, 1.f);
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
    if (RELU) apply_mask<NT>(dA, mkA, 1.f);
    store_rows<NT>(dA, a.dacts + row * Hp, h);
  }
  TRACE(15);
  TRACE_REAL(14);
}

// relu backward on a slice (sign bits of the saved activation of that layer), then its rows of the layer-gradient
// buffer
template <int NT>
__device__ __forceinline__ void coop_mask_store(f32x16 (&slice)[Coop<NT>::TW], int t0, const uint32_t* mask_row, float scale,
                                                float* dact_row) {
  constexpr int TW = Coop<NT>::TW;
#pragma unroll
  for (int i = 0; i < TW; ++i)
    if (t0 + i < NT) {
      const uint32_t bits = ((const uint16_t*)mask_row)[t0 + i];
#pragma unroll
      for (int r = 0; r < 16; ++r) slice[i][r] = ((bits >> r) & 1u) ? slice[i][r] * scale : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = slice[i][4 * q + e];
        *(f32x4*)(dact_row + 32 * (t0 + i) + 8 * q) = o;
      }
    }
}

template <int NT>
__global__ __launch_bounds__(256, 1) void k_mlp_bwd_coop(BwdArgs a) {
  constexpr int TW = Coop<NT>::TW, Hp = 32 * NT, MW = mask_words(NT);
  __shared__ __attribute__((aligned(16))) float lds[Coop<NT>::LDS_BYTES / 4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t0 = wv * TW;
  const int64_t row0 = a.row_begin + (int64_t)blockIdx.x * 32;
  const int m = lane & 31, h = lane >> 5;
  const int64_t row = row0 + m;
  const bool live = row < a.rows;
  const int64_t rl = live ? row : a.rows - 1;
  const int64_t rp = (a.rows + 31) & ~(int64_t)31;
  const int64_t act_stride = rp * Hp, mask_stride = rp * 2 * MW;
  const int lf = a.L.nl - 1;
  float* dact_row = a.dacts + row * Hp + 4 * h;
  const uint32_t* mask_row = a.masks + (row * 2 + h) * MW;
  const float dscale = a.drop_keep ? a.drop_scale : 1.f;

  f32x16 F[NT], res[TW], acc[TW];   // as in the forward kernel: full layer input, own slice of the skip-path gradient
  // ---- this wave's tiles of Wf^T gy
  {
    const f32x4* W = (const f32x4*)(a.packed + a.L.wt[lf]);
    const int S = a.L.S_out;
    const __amdgpu_buffer_rsrc_t img = make_rsrc(W, (uint32_t)(NT * S) * 1024u);
    const float* gyr = a.gy + rl * a.ldgy + 4 * h;
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = zero16();
    for (int s = 0; s < S; ++s) {
      f32x4 b = *(const f32x4*)(gyr + 8 * s);
      if (!live) b = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 w[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) w[i] = buf_load4(img, (uint32_t)lane * 16u, (uint32_t)((t0 + i) * S + s) * 1024u);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TW; ++i) acc[i] = MFMA(w[i][e], b[e], acc[i]);
    }
  }
  coop_mask_store<NT>(acc, t0, mask_row + (lf - 1) * mask_stride, 1.f, dact_row + (lf - 1) * act_stride);
#pragma unroll
  for (int i = 0; i < TW; ++i) res[i] = acc[i];
  coop_exchange<NT>(lds, lane, t0, acc, F);

  for (int blk = a.nb - 1; blk >= 0; --blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    coop_gemm<NT, true, false>((const f32x4*)(a.packed + a.L.wt[l2]), nullptr, lane, t0, F, acc);
    coop_mask_store<NT>(acc, t0, mask_row + l1 * mask_stride, dscale, dact_row + l1 * act_stride);
    coop_exchange<NT>(lds, lane, t0, acc, F);
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = res[i];                           // skip path
    coop_gemm<NT, false, false>((const f32x4*)(a.packed + a.L.wt[l1]), nullptr, lane, t0, F, acc);
    coop_mask_store<NT>(acc, t0, mask_row + (l1 - 1) * mask_stride, 1.f, dact_row + (l1 - 1) * act_stride);
#pragma unroll
    for (int i = 0; i < TW; ++i) res[i] = acc[i];
    coop_exchange<NT>(lds, lane, t0, acc, F);
  }

  if (a.dx) {
    const f32x4* W = (const f32x4*)(a.packed + a.L.wt[0]);
    constexpr int S = 4 * NT;
    uint32_t nomask[MW];
    for (int it = wv; it < a.L.T_in; it += 4) {
      f32x16 o16 = zero16();
      gemm_h1<NT, false, false, IN_ASIS>(W + (int64_t)it * S * 64, lane, F, o16, nullptr, 0u, nullptr, nomask, 1.f);
      if (live) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = o16[4 * q + e];
          *(f32x4*)(a.dx + row * a.lddx + 32 * it + 8 * q + 4 * h) = o;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// 16-row tiles for the smallest batches (at most one round of 256 workgroups)
// ------------------------------------------------------------------------------------------
// The four-wave kernels above are bound by one wave's MFMA issue: 32 rows x Hp outputs x Hp inputs per layer
// over four SIMDs.  While the batch does not fill the chip anyway, halving the rows per workgroup halves
// that chain: v_mfma_f32_16x16x4_f32 (8 passes), weights as the A operand (lane (o, g = lane>>4): output o,
// inputs 4 g + i), activations as the B operand (lane (m, g): row m).  The accumulator (lane (m, g), register i
// <-> output 16 t + 4 g + i) is again the B operand of the next layer when the inputs are visited in the order
// 16 t + 4 g + i (tile t, step i), and the A operand of (output tile ot, input tile t) is a quarter of two
// consecutive fragments of the same packed image the 32-row kernels read: lane (o, g) takes the 16 bytes of
// lane slot 32 (g & 1) + 16 (ot & 1) + o in fragment 2 t + (g >> 1) of tile ot >> 1.  Saved activations and
// sign-bit words keep the layout of the other kernel families, which therefore mix freely with these.
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

static inline bool dmv_use_rows16(int64_t rows) { return (rows + 15) / 16 <= 256; }

// W waves share a tile: 4, one per SIMD.  (8, two per SIMD, was measured and gains nothing - 27 against 28 us a pass at
// 2 560 rows, hidden 256: a tile's layer is 16 x Hp x Hp MACs on ONE compute unit, whose four matrix pipes need 3.4 us
// for it at hidden 256 however many waves feed them.)
template <int NT, int W = 4>
struct Rows16 {
  static constexpr int NK = 2 * NT;                       // 16-wide tiles of the hidden width
  static constexpr int TW = 2 * ((NT + W - 1) / W);       // output tiles per wave (whole 32-wide tiles: sign-bit halfwords)
  static constexpr int PF = W == 8 ? 4 : 2;               // input tiles a weight fragment is requested ahead of its use
                                                          // (3 and 4 were measured in the one-launch sampler: slower)
  static constexpr int LDS_BYTES = NK * 1024;
};
template <int NT, int W> using R16Slice = f32x4[Rows16<NT, W>::TW];   // a wave's output tiles of a layer
template <int NT> using R16Full = f32x4[2 * NT];                      // all tiles of a layer

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// byte offset of this lane's 16 bytes inside the fragment pair of (output tile parity 0, input tile 0)
__device__ __forceinline__ uint32_t r16_lane_off(int lane) {
  const int m = lane & 15, g = lane >> 4;
  return (uint32_t)((g >> 1) * 1024 + ((g & 1) * 32 + m) * 16);
}

// scalar offset of output tile ot (16 wide) in an image with S groups per 32-wide tile
__device__ __forceinline__ uint32_t r16_tile_off(int ot, int S) {
  return (uint32_t)(ot >> 1) * (uint32_t)S * 1024u + (uint32_t)(ot & 1) * 256u;
}

// bias of output tile ot as an A operand: lanes g = 0 carry bias[16 ot + o], the others zero
__device__ __forceinline__ float r16_bias(__amdgpu_buffer_rsrc_t bimg, int lane, int ot) {
  const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
      bimg, (uint32_t)(lane & 15) * 4u, (uint32_t)(ot >> 1) * 256u + (uint32_t)(ot & 1) * 64u, 0));
  return lane < 16 ? v : 0.f;
}

// acc[i] (+)= W[tile t0 + i] in over the 32 NT hidden inputs; fragments PF input tiles ahead
template <int NT, int W, bool FRESH, bool BIAS>
__device__ __forceinline__ void r16_gemm(const f32x4* __restrict__ Wimg, const float* __restrict__ Bimg, int lane, int t0,
                                         const R16Full<NT>& in, R16Slice<NT, W>& acc) {
  constexpr int S = 4 * NT, TW = Rows16<NT, W>::TW, NK = Rows16<NT, W>::NK, PF = Rows16<NT, W>::PF;
  const uint32_t lo = r16_lane_off(lane);
  const __amdgpu_buffer_rsrc_t img = make_rsrc(Wimg, NT * S * 1024u);
  uint32_t base[TW];
#pragma unroll
  for (int i = 0; i < TW; ++i) base[i] = r16_tile_off(t0 + i, S);   // tiles past NK fall outside the resource: zeros
  f32x4 w[PF + 1][TW];
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int i = 0; i < TW; ++i) w[p][i] = (p < NK) ? buf_load4(img, lo, base[i] + (uint32_t)p * 2048u) : zero4();
  if (BIAS) {
    const __amdgpu_buffer_rsrc_t bimg = make_rsrc(Bimg, NT * 256u);
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = MFMA16(r16_bias(bimg, lane, t0 + i), 1.f, FRESH ? zero4() : acc[i]);
  }
#pragma unroll
  for (int t = 0; t < NK; ++t) {
    if (t + PF < NK) {
#pragma unroll
      for (int i = 0; i < TW; ++i) w[(t + PF) % (PF + 1)][i] = buf_load4(img, lo, base[i] + (uint32_t)(t + PF) * 2048u);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float b = in[t][e];
#pragma unroll
      for (int i = 0; i < TW; ++i)
        acc[i] = MFMA16(w[t % (PF + 1)][i][e], b, (FRESH && !BIAS && t == 0 && e == 0) ? zero4() : acc[i]);
    }
  }
}

// one output tile (fragment offset `toff`) from all hidden inputs, single accumulator
template <int NT>
__device__ __forceinline__ f32x4 r16_gemm1(__amdgpu_buffer_rsrc_t img, uint32_t toff, int lane, const R16Full<NT>& in,
                                           f32x4 acc) {
  constexpr int NK = Rows16<NT>::NK;
  const uint32_t lo = r16_lane_off(lane);
  f32x4 w[NK];
#pragma unroll
  for (int t = 0; t < NK; ++t) w[t] = buf_load4(img, lo, toff + (uint32_t)t * 2048u);
#pragma unroll
  for (int t = 0; t < NK; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = MFMA16(w[t][e], in[t][e], acc);
  return acc;
}

// all-gather of the waves' output slices through LDS
template <int NT, int W>
__device__ __forceinline__ void r16_exchange(float* lds, int lane, int t0, const R16Slice<NT, W>& slice,
                                             R16Full<NT>& full) {
  constexpr int TW = Rows16<NT, W>::TW, NK = Rows16<NT, W>::NK;
  f32x4* L4 = (f32x4*)lds;
  __syncthreads();                                   // previous readers are done with the buffer
#pragma unroll
  for (int i = 0; i < TW; ++i)
    if (t0 + i < NK) L4[(t0 + i) * 64 + lane] = slice[i];
  __syncthreads();
#pragma unroll
  for (int t = 0; t < NK; ++t) full[t] = L4[t * 64 + lane];
}

template <int NT, int W>
__device__ __forceinline__ void r16_own_slice(const R16Full<NT>& full, int wv, R16Slice<NT, W>& slice) {
  constexpr int TW = Rows16<NT, W>::TW, NK = Rows16<NT, W>::NK;
#pragma unroll
  for (int ww = 0; ww < W; ++ww)
    if (ww == wv) {
#pragma unroll
      for (int i = 0; i < TW; ++i) slice[i] = full[ww * TW + i < NK ? ww * TW + i : NK - 1];   // past NK: never stored
    }
}

// relu on a slice; with SAVE also its columns of the saved activation and its sign bits.  Halfword t of a
// (row, h) mask line holds bit 4 q + e for feature 32 t + 8 q + 4 h + e: this lane owns q = 2 (ot & 1) + (g >> 1)
// of line h = g & 1, the lane 32 further the other q of the same line.
// DROP: inverted dropout after the relu; kb = the keep bits of this lane's (row, h = g & 1) line (drop_words).
// st: whether this lane's row stores what it saves (a tile of the fused encoder-decoder kernel holds an encoder row once
// per sample row that uses it; one of the copies stores).
template <int NT, int W, bool SAVE, bool DROP = false>
__device__ __forceinline__ void r16_relu_save(R16Slice<NT, W>& slice, int lane, int t0, float* act_row, uint32_t* mask_line,
                                              const uint32_t* kb = nullptr, float scale = 1.f, bool st = true) {
  constexpr int TW = Rows16<NT, W>::TW, NK = Rows16<NT, W>::NK;
  const int g = lane >> 4;
#pragma unroll
  for (int p = 0; p < TW / 2; ++p) {
    uint32_t part = 0u;
    uint32_t keep16 = 0xffffu;
    if (DROP) {
      uint32_t kw[mask_words(NT)];
#pragma unroll
      for (int j = 0; j < mask_words(NT); ++j) kw[j] = kb[j];
      const int t32 = (t0 >> 1) + p;                 // the 32-wide tile the pair (2 p, 2 p + 1) of this wave's tiles forms
      keep16 = half_of<mask_words(NT)>(kw, t32 < NT ? t32 : 0);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = 2 * p + j;
      uint32_t nib = 0u;
      const uint32_t knib = keep16 >> (8 * j + 4 * (g >> 1));     // q = 2 j + (g >> 1)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        slice[i][e] = __builtin_amdgcn_fmed3f(slice[i][e], 0.f, __builtin_inff());
        if (DROP) slice[i][e] = ((knib >> e) & 1u) ? scale * slice[i][e] : 0.f;
        nib |= (slice[i][e] > 0.f ? 1u : 0u) << e;
      }
      part |= nib << (8 * j + 4 * (g >> 1));
      if (SAVE && st && t0 + i < NK) *(f32x4*)(act_row + 16 * (t0 + i)) = slice[i];
    }
    if (SAVE) {
      part |= (uint32_t)__shfl_xor((int)part, 32);
      if (g < 2 && st && t0 + 2 * p < NK) ((uint16_t*)mask_line)[(t0 >> 1) + p] = (uint16_t)part;
    }
  }
}

// One 16-row tile through a whole network.  `row` addresses the saved buffers and `rl` (a valid row) the dropout
// stream; `st` predicates the stores of what is saved; in4(k0) returns columns k0 .. k0+3 of this lane's row of the
// network input, out4(ot, o) receives output columns 16 ot + 4 g .. + 3 of this lane's row (output tiles go round-robin
// over the waves).
template <int NT, bool SAVE, bool DROP, int W, class In4, class Out4>
__device__ __forceinline__ void r16_forward_chain(const FwdArgs& a, float* lds, int lane, int wv, int64_t row, int64_t rl,
                                                  bool st, In4 in4, Out4 out4) {
  constexpr int TW = Rows16<NT, W>::TW, NK = Rows16<NT, W>::NK, Hp = 32 * NT, MW = mask_words(NT);
  const int t0 = wv * TW;
  const int g = lane >> 4;
  const int64_t rp = (a.rows + 31) & ~(int64_t)31;
  const int64_t act_stride = rp * Hp, mask_stride = rp * 2 * MW;
  float* act_row = SAVE ? a.acts + row * Hp + 4 * g : nullptr;                    // layer 0; + l * act_stride
  uint32_t* mask_line = SAVE ? a.masks + (row * 2 + (g & 1)) * MW : nullptr;      // layer 0; + l * mask_stride
  const uint32_t lo = r16_lane_off(lane);

  f32x4 A[NK], T[NK], acc[TW];
  const uint64_t drop_q0 = (DROP && a.drop.mode == DROP_GEN) ? drop_first_call(a.drop) : 0ull;

  // ---- initial layer: this wave's output tiles from the network input
  {
    const int S = a.L.S_in;
    const __amdgpu_buffer_rsrc_t img = make_rsrc(a.packed + a.L.w[0], (uint32_t)(NT * S) * 1024u);
    const __amdgpu_buffer_rsrc_t bimg = make_rsrc(a.packed + a.L.b[0], NT * 256u);
    uint32_t base[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      base[i] = r16_tile_off(t0 + i, S);
      acc[i] = MFMA16(r16_bias(bimg, lane, t0 + i), 1.f, zero4());
    }
    float* xs_row = SAVE ? a.xs + row * (32 * a.L.T_in) + 4 * g : nullptr;
    for (int t = 0; 2 * t < S; ++t) {
      const bool in_image = 2 * t + (g >> 1) < S;      // an odd group count leaves the last half tile empty
      const f32x4 b = in4(16 * t + 4 * g);
      if (SAVE && st && wv == 0 && in_image) *(f32x4*)(xs_row + 16 * t) = b;
      f32x4 w[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        w[i] = buf_load4(img, lo, base[i] + (uint32_t)t * 2048u);
        if (!in_image) w[i] = zero4();
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TW; ++i) acc[i] = MFMA16(w[i][e], b[e], acc[i]);
    }
  }
  r16_relu_save<NT, W, SAVE>(acc, lane, t0, act_row, mask_line, nullptr, 1.f, st);
  r16_exchange<NT, W>(lds, lane, t0, acc, A);

  // ---- residual blocks
  for (int blk = 0; blk < a.nb; ++blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    uint32_t kb[MW];
    if (DROP) drop_words<NT>(a.drop, drop_q0, a.rows, a.hidden, blk, rl, g & 1, t0 >> 1, (t0 + TW) >> 1, kb);
    r16_gemm<NT, W, true, true>((const f32x4*)(a.packed + a.L.w[l1]), a.packed + a.L.b[l1], lane, t0, A, acc);
    r16_relu_save<NT, W, SAVE, DROP>(acc, lane, t0, act_row + l1 * act_stride, mask_line + l1 * mask_stride, kb, a.drop_scale, st);
    r16_exchange<NT, W>(lds, lane, t0, acc, T);
    r16_own_slice<NT, W>(A, wv, acc);                  // the residual stream is the accumulator of the second GEMM
    r16_gemm<NT, W, false, true>((const f32x4*)(a.packed + a.L.w[l2]), a.packed + a.L.b[l2], lane, t0, T, acc);
    r16_relu_save<NT, W, SAVE>(acc, lane, t0, act_row + l2 * act_stride, mask_line + l2 * mask_stride, nullptr, 1.f, st);
    r16_exchange<NT, W>(lds, lane, t0, acc, A);
  }

  // ---- final layer: output tiles round-robin over the waves
  {
    const int lf = a.L.nl - 1;
    constexpr int S = 4 * NT;
    const __amdgpu_buffer_rsrc_t img = make_rsrc(a.packed + a.L.w[lf], (uint32_t)(a.L.OT * S) * 1024u);
    const __amdgpu_buffer_rsrc_t bimg = make_rsrc(a.packed + a.L.b[lf], (uint32_t)a.L.OT * 256u);
    for (int ot = wv; ot < 2 * a.L.OT; ot += W)
      out4(ot, r16_gemm1<NT>(img, r16_tile_off(ot, S), lane, A, MFMA16(r16_bias(bimg, lane, ot), 1.f, zero4())));
  }
}

template <int NT, bool SAVE, bool DROP, int W>
__global__ __launch_bounds__(64 * W, 1) void k_mlp_fwd_rows16(FwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[Rows16<NT, W>::LDS_BYTES / 4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int64_t row = a.row_begin + (int64_t)blockIdx.x * 16 + m;
  const bool live = row < a.rows;
  const int64_t rl = live ? row : a.rows - 1;
  // (the saved buffers hold whole 32-row tiles, so rows past the batch store too)
  r16_forward_chain<NT, SAVE, DROP, W>(
      a, lds, lane, wv, row, rl, true, [&](int k0) { return first_layer_input4(a, rl, k0); },
      [&](int ot, f32x4 o) { if (live) *(f32x4*)(a.y + row * a.ldy + 16 * ot + 4 * g) = o; });
}

// ------------------------------------------------------------------------------------------
// Encoder -> latent sample -> decoder of a small batch in one launch (vgiwae/models/vae.py:285-318: vae_forward's
// predict_var_latent_params, rsample and generator_network calls).  A workgroup owns 16 consecutive sample rows
// n = (s B + b) G + g.  It first runs the ENCODER on the encoder rows those samples draw from (row b, or row b G + g when
// every imputation has an encoder row of its own; a row needed by several of the tile's samples is simply held several
// times), keeps the 16 output rows in LDS, forms z = mu + sigma eps there, and runs the decoder on it: no round trip of
// the encoder outputs or the latents through HBM and one launch instead of three.  The encoder is recomputed by every
// tile that needs a row (S times over the batch), on compute units a 512-row encoder pass would leave idle.  All
// buffers the three separate launches produce are written with the same values (encoder outputs and saved activations
// by the tile that holds a row's first sample), so everything downstream - objective kernels, backward passes - is
// unchanged.
// ------------------------------------------------------------------------------------------
struct VaeFwdArgs {
  FwdArgs enc, dec;
  int64_t N;                  // sample rows
  int B, G, Zd;
  int per_comp;               // 1: an encoder row per imputation (G - n_prior of them per datapoint), 0: one per datapoint
  int rows_per_b;             // encoder rows per datapoint
  int n_prior;                // the last n_prior of a datapoint's G components are standard-normal proposals without
  float prior_raw;            // an encoder row (mu = 0, raw sigma = prior_raw); only with one row per imputation
  QAddr q;
  const float* eps;           // [N][ldz] or null (Philox)
  float* eps_out;             // nullable
  float* Z;                   // [N][ldz]
  int64_t ldz;
  uint64_t seed, offset;
  const uint64_t* offset_base;
  int ZP;                     // latent width rounded up to 16: pitch of the latent tile in LDS
  // The IWAE / VAE bound in the same launch (one Gaussian per datapoint, G = 1): tiles are then datapoint-major - the
  // S = nz I samples of 16 / S datapoints - so the logsumexp over a datapoint's samples stays inside the workgroup.
  int obj_on, S, nz, I, D, gen_dist, dreg;
  float* log_w;               // nullable [N]
  float* bound;               // [B]
  float* wgt;                 // nullable [N]
  float* zscale;              // nullable [N]
  unsigned* ticket;           // nullable: with it the launch also forms the batch mean
  float* mean; float* neg_mean;
  // ... or a stratified mixture of mix_C = G Gaussians per datapoint, one sample row per component (MissSIWAE, the tight
  // bound: logsumexp over (i, g) of log w + log pi_g): tiles of 16 / (S G) datapoints
  int mix_C;                  // 0: one Gaussian per datapoint
  float* terms;               // nullable [5][N]: masked / full generator term, prior, own component, mixture
  float* comp_lp;             // nullable [N][mix_C]
  float* dlogit;              // nullable [B][G]
};

// MIX: the mixture bound's epilogue is compiled in (a.mix_C != 0); without it the component count is known to be one
template <int NT, bool SAVE, bool MIX>
__global__ __launch_bounds__(256, 1) void k_vae_fwd_rows16(VaeFwdArgs a) {
  constexpr int W = 4;
  __shared__ __attribute__((aligned(16))) float lds[Rows16<NT, W>::LDS_BYTES / 4];
  __shared__ int par_off[16];                                  // where a row's (mu | raw sigma) start inside its encoder row
  __shared__ int nrow[16];                                     // sample row of a tile row, -1: none
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [16][ldE] encoder outputs, then [16][ZP] latents
  const int ldE = (int)a.enc.ldy, ldD = (int)a.dec.ldy;
  float* et = tile;
  float* zt = tile + 16 * ldE;
  float* dt = zt + 16 * a.ZP;                                  // [16][ldD] decoder outputs (with the bound only)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, g = lane >> 4;
  int64_t n = (int64_t)blockIdx.x * 16 + m;
  bool live = n < a.N;
  if (a.obj_on) {                          // tile row (q, s, g): sample s, component g of the tile's q-th datapoint
    const int G1 = MIX ? a.G : 1;
    const int SG = a.S * G1, Q = 16 / SG, q = m / SG, bq = (int)blockIdx.x * Q + q;
    const int rem = m - q * SG, s = rem / G1;
    live = q < Q && bq < a.B;
    n = live ? ((int64_t)s * a.B + bq) * G1 + (rem - s * G1) : 0;
  }
  const int nl = (int)(live ? n : (a.obj_on ? 0 : a.N - 1));   // N <= 4096 rows here
  const int BG = a.B * a.G;
  const int bg = nl % BG;
  const int b = bg / a.G, comp = bg - b * a.G;
  // encoder row of this sample (a prior component has none: it rides along on a neighbour's row and ignores it), and
  // the row of the encoder-output array its parameters live in
  const bool prior = a.per_comp && comp >= a.rows_per_b;
  const int er = a.per_comp ? b * a.rows_per_b + (prior ? a.rows_per_b - 1 : comp) : b;
  const int orow = a.per_comp ? bg : b;
  // the first sample of an encoder row stores what the encoder pass saves and the row's outputs
  const bool owner = live && nl < BG && !prior && (a.per_comp || comp == 0);
  if (wv == 0 && g == 0) {
    par_off[m] = prior ? -1 : (int)((int64_t)b * a.q.sb + (int64_t)comp * a.q.sg - (int64_t)orow * ldE);
    nrow[m] = live ? nl : -1;
  }

  r16_forward_chain<NT, SAVE, false, W>(
      a.enc, lds, lane, wv, er, er, owner, [&](int k0) { return first_layer_input4(a.enc, er, k0); },
      [&](int ot, f32x4 o) {
        *(f32x4*)(et + m * ldE + 16 * ot + 4 * g) = o;
        if (owner) *(f32x4*)(a.enc.y + (int64_t)orow * ldE + 16 * ot + 4 * g) = o;
      });
  __syncthreads();

  // ---- z = mu + sigma eps for the tile's rows, into LDS (the decoder's input) and out to Z / eps_out
  {
    const uint64_t off = a.offset + (a.offset_base ? a.offset_base[0] : 0ull);
    const int JW = a.ZP > (int)a.ldz ? a.ZP : (int)a.ldz;
    for (int idx = threadIdx.x; idx < 16 * JW; idx += 256) {
      const int mm = idx / JW, j = idx - mm * JW;
      const int64_t n2 = nrow[mm] >= 0 ? nrow[mm] : a.N;
      float z = 0.f, e = 0.f;
      if (j < a.Zd) {
        const int po = par_off[mm];
        const float* pr = et + mm * ldE + (po < 0 ? 0 : po);
        const float mu = po < 0 ? 0.f : pr[j];
        const float sg = dmv_softplus(po < 0 ? a.prior_raw : pr[a.q.raw_off + j]) + a.q.eps;
        const int64_t i = (n2 < a.N ? n2 : a.N - 1) * a.ldz + j;
        e = a.eps ? a.eps[i] : dmv_philox_normal(a.seed, off + (uint64_t)i);
        z = mu + sg * e;
      }
      if (j < a.ZP) zt[mm * a.ZP + j] = z;
      if (n2 < a.N && j < (int)a.ldz) {
        a.Z[n2 * a.ldz + j] = z;
        if (a.eps_out) a.eps_out[n2 * a.ldz + j] = e;
      }
    }
  }
  __syncthreads();

  if (!a.obj_on) {
    r16_forward_chain<NT, SAVE, false, W>(
        a.dec, lds, lane, wv, n, nl, true, [&](int k0) { return *(const f32x4*)(zt + m * a.ZP + k0); },
        [&](int ot, f32x4 o) { if (live) *(f32x4*)(a.dec.y + n * a.dec.ldy + 16 * ot + 4 * g) = o; });
    return;
  }
  r16_forward_chain<NT, SAVE, false, W>(
      a.dec, lds, lane, wv, nl, nl, live, [&](int k0) { return *(const f32x4*)(zt + m * a.ZP + k0); },
      [&](int ot, f32x4 o) {
        *(f32x4*)(dt + m * ldD + 16 * ot + 4 * g) = o;
        if (live) *(f32x4*)(a.dec.y + n * a.dec.ldy + 16 * ot + 4 * g) = o;
      });
  __syncthreads();

  // ---- the bound (vgiwae/shared/iwae.py:9-55 log-weights, vgiwae/models/iwae.py:70-123 logsumexp and mean): a group of
  // 16 lanes per sample row shares out the dimensions of its three sums
  __shared__ float lw[16], lpi[16], bz[16], sh[32];
  __shared__ unsigned last;
  const int tid = threadIdx.x;
  const int G1 = MIX ? a.G : 1;
  {
    const int p = tid >> 4, l16 = tid & 15;
    const int np = nrow[p];
    if (np >= 0) {
      const int bb = (np / G1) % a.B;
      const float* z = zt + p * a.ZP;
      const float* pe = et + p * ldE + par_off[p];
      float pr = 0.f, qo = 0.f;
      for (int j = l16; j < a.Zd; j += 16) {
        const float zj = z[j];
        pr += -(zj * zj) * 0.5f - LOG_SQRT_2PI;
        const float sg = dmv_softplus(pe[a.q.raw_off + j]) + a.q.eps;
        const float rs = __builtin_amdgcn_rcpf(sg);
        qo += q_logp(QEntry{pe[j], 0.5f * rs * rs, dmv_log(sg) + LOG_SQRT_2PI, 0.f}, zj);
      }
      const float* x = a.enc.x + (int64_t)bb * a.enc.ldx;
      const uint8_t* mk = a.enc.m + (int64_t)bb * a.enc.ldm;
      const float* dec = dt + p * ldD;
      const float geps = a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f;
      float gm = 0.f, gf = 0.f;
      for (int d = l16; d < a.D; d += 16) {
        float lp;
        if (a.gen_dist <= DMV_GEN_NORMAL_EPS) lp = log_normal_r(x[d], dec[d], dmv_softplus(dec[a.D + d]) + geps);
        else if (a.gen_dist == DMV_GEN_BERN) lp = log_bern(x[d], dec[d]);
        else lp = log_bern((x[d] + 1.f) * 0.5f, dec[d]);
        if (MIX) gf += lp;
        gm += lp * (mk[d] ? 1.f : 0.f);          // (the product the reference forms: non-finite under a missing entry -> NaN)
      }
      gm = gsum<16>(gm); pr = gsum<16>(pr); qo = gsum<16>(qo);
      if (MIX) gf = gsum<16>(gf);
      float qm = qo;
      if (MIX) lpi[p] = 0.f;
      if (MIX && a.mix_C) {
        // log q of the whole mixture at z (vgiwae/models/multiple_iwae.py:124-199): the datapoint's encoder row holds
        // every component's parameters and the logits
        const float* row = et + p * ldE;
        const float* lg = row + a.q.logit_off;
        float lm = -INFINITY;
        for (int c = 0; c < a.mix_C; ++c) lm = fmaxf(lm, lg[c]);
        float ss = 0.f;
        for (int c = 0; c < a.mix_C; ++c) ss += dmv_exp(lg[c] - lm);
        const float lse_logit = lm + dmv_log(ss);
        float mx = -INFINITY, se = 0.f;
        for (int c = 0; c < a.mix_C; ++c) {
          const float* pc = row + (int64_t)c * a.q.sg;
          float lp = 0.f;
          for (int j = l16; j < a.Zd; j += 16)
            lp += log_normal_r(z[j], pc[j], dmv_softplus(pc[a.q.raw_off + j]) + a.q.eps);
          lp = gsum<16>(lp);
          if (a.comp_lp && l16 == 0) a.comp_lp[(int64_t)np * a.mix_C + c] = lp;
          const float t = lp + lg[c] - lse_logit;
          const float nm = fmaxf(mx, t);
          se = se * dmv_exp(mx - nm) + dmv_exp(t - nm);
          mx = nm;
        }
        qm = mx + dmv_log(se);
        lpi[p] = lg[np % G1] - lse_logit;
      }
      const float w = gm + pr - qm;
      if (l16 == 0) {
        lw[p] = w;
        if (a.log_w) a.log_w[np] = w;
        if (MIX && a.terms) {
          a.terms[np] = gm; a.terms[a.N + np] = gf; a.terms[2 * a.N + np] = pr; a.terms[3 * a.N + np] = qo;
          a.terms[4 * a.N + np] = qm;
        }
      }
    }
  }
  __syncthreads();
  const int IG = a.I * G1, Q = 16 / (a.S * G1);
  if (tid < Q * a.nz) {                           // logsumexp over the I G rows of (datapoint, outer sample)
    const int q = tid / a.nz, r0 = q * a.S * G1 + (tid - q * a.nz) * IG;
    if (nrow[r0] >= 0) {
      if (MIX)
        for (int i = 0; i < IG; ++i) lw[r0 + i] += lpi[r0 + i];
      float mx = -INFINITY;
      for (int i = 0; i < IG; ++i) mx = fmaxf(mx, lw[r0 + i]);
      float se = 0.f;
      for (int i = 0; i < IG; ++i) se += expf(lw[r0 + i] - mx);
      const float lse = mx + logf(se);
      bz[tid] = lse - logf((float)a.I);
      for (int i = 0; i < IG; ++i) lw[r0 + i] = expf(lw[r0 + i] - lse);
    }
  }
  __syncthreads();
  const float inv_nz = 1.f / (float)a.nz;
  if (tid < 16 && nrow[tid] >= 0) {
    const float pw = lw[tid];
    if (a.wgt) a.wgt[nrow[tid]] = pw * inv_nz;
    if (a.zscale) a.zscale[nrow[tid]] = a.dreg ? pw : 1.f;
  }
  if (MIX && a.dlogit && tid >= 32 && tid - 32 < Q * G1) {
    // d/d logit_c of sum_n p_n log pi_{g(n)} = (sum_{n: g(n) = c} p_n) - pi_c per outer sample
    const int q = (tid - 32) / G1, gg = (tid - 32) - q * G1, r0 = q * a.S * G1;
    if (nrow[r0] >= 0) {
      float acc = 0.f;
      for (int s = 0; s < a.S; ++s) acc += lw[r0 + s * G1 + gg];
      a.dlogit[(int64_t)((nrow[r0] / G1) % a.B) * G1 + gg] = (acc - (float)a.nz * expf(lpi[r0 + gg])) * inv_nz;
    }
  }
  // The batch mean in this launch.  The usual way (make the workgroup's results visible, take a ticket) would cost a
  // device-scope release here, i.e. a write-back of every dirty L2 line the forward pass has just produced (measured:
  // +17 us).  Instead the few bounds of a tile go out as device-scope atomic exchanges, which are performed at the level
  // all compute dies share; the ticket is taken only after their old values have come back (they are parked in LDS, so
  // the barrier waits for them), and the workgroup that draws the last ticket reads the bounds after an acquire.
  if (tid >= 64 && tid - 64 < Q && nrow[(tid - 64) * a.S * G1] >= 0) {
    const int q = tid - 64;
    float acc = 0.f;
    for (int zz = 0; zz < a.nz; ++zz) acc += bz[q * a.nz + zz];
    float* dst = a.bound + (nrow[q * a.S * G1] / G1) % a.B;
    if (a.ticket) bz[q * a.nz] = atomicExch(dst, acc * inv_nz);
    else *dst = acc * inv_nz;
  }
  if (!a.ticket) return;
  __syncthreads();
  if (tid == 0) last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
  __syncthreads();
  if (!last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  block_mean2(sh, a.bound, nullptr, (int64_t)a.B, 1.f / (float)a.B, a.mean, nullptr, a.neg_mean);
  if (tid == 0) *a.ticket = 0u;
}

// ------------------------------------------------------------------------------------------
// LAIR's whole imputation update in one launch (vgiwae/shared/vae_imputation.py:390-563: encoder on the current
// imputations, proposals z ~ q(. | x_k) plus standard-normal replenish proposals, decoder, DMIS log-weights against the
// uniform mixture of the datapoint's components (shared/iwae.py:57-100), global multinomial resampling (:181-198), new
// imputations from the chosen proposals (:521-561); and :671-694, the final resampling from the pooled proposals of all
// iterations with one more decoder pass).  Nothing in the update couples two datapoints, so a workgroup owns DP = 16 / Kp
// datapoints - their Kp = K + r proposals are the rows of one 16-row tile - and walks them through all T iterations:
// encoder outputs, latents, decoder outputs, log-weights and the proposals kept for the final resampling never leave
// LDS; only the imputations themselves are updated in global memory (in place, read back by the same workgroup after a
// barrier).  One importance sample per component (I = 1, what every shipped configuration uses).
// Noise: explicit arrays (parity with recorded draws) or the Philox ranges the separate launches would take, handed
// over per iteration, so both ways consume the same stream.
// ------------------------------------------------------------------------------------------
#define DMV_LAIR_MAX_T 32
struct LairArgs {
  FwdArgs enc, dec;           // enc.x = the imputations X [B K][D] (updated in place), enc.m = their masks; y unused
  int B, K, Kp, Zd, D, T, DP;
  int gen_dist, final_resample;
  int resample;               // 1: LAIR.  0: pseudo-Gibbs (:38-52) - every row its own datapoint (K = Kp = 1), the one
                              // proposal is taken as it is, and the sample is clamped before the blend
  float prior_raw, q_eps;
  int raw_off;                // where the raw sigma starts in an encoder output row
  int ZP, ldE, ldD;
  uint64_t seed;
  uint64_t off_eps[DMV_LAIR_MAX_T], off_draw[DMV_LAIR_MAX_T + 1], off_tail[DMV_LAIR_MAX_T + 1];
  const uint64_t* offset_base;
  const float* eps;           // nullable [T][B Kp][Zd]
  const int64_t* draws;       // nullable [T][B][Kp]: proposal numbers chosen by the resampling
  const int64_t* final_draws; // nullable [B][K]: pooled proposal numbers t Kp + c
  const float* noise;         // nullable [T (+1)][B K][D]
  const uint8_t* M;           // [B K][D]
  float* X;                   // [B K][D]
  const float* clip_min; const float* clip_max;
  float* buf; const int64_t* idx_imp;
  float* lw_out;              // nullable [T][B][Kp]: the log-weights of every iteration (checks)
};

template <int NT>
__global__ __launch_bounds__(256, 1) void k_lair_rows16(LairArgs a) {
  constexpr int W = 4;
  __shared__ __attribute__((aligned(16))) float lds[Rows16<NT, W>::LDS_BYTES / 4];
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int ldE = a.ldE, ldD = a.ldD, ZP = a.ZP, Zd = a.Zd, Kp = a.Kp, K = a.K, D = a.D;
  float* et = tile;                              // [16][ldE] encoder outputs
  float* zt = et + 16 * ldE;                     // [16][ZP] proposals (the decoder's input)
  float* dt = zt + 16 * ZP;                      // [16][ldD] decoder outputs
  QEntry* qt = (QEntry*)(dt + 16 * ldD);         // [16][Zd] the components' Gaussians
  float* lw = (float*)(qt + 16 * Zd);            // [16] log-weights of this iteration
  int* pick = (int*)(lw + 16);                   // [16] tile row (or pooled row) chosen for output row (dp, k)
  float* lwall = (float*)(pick + 16);            // [T][16]       final resampling only
  float* zall = lwall + 16 * a.T;                // [T][16][Zd]   final resampling only
  const int RX = a.DP * K;                       // imputation rows of a tile
  float* xs = zall + (a.final_resample ? 16 * a.T * Zd : 0);   // [RX][D] the tile's imputations, updated in place
  float* nzs = xs + RX * D;                      // [RX][D] this iteration's sampling noise
  float* us = nzs + RX * D;                      // [16] this iteration's resampling uniforms
  uint8_t* ms = (uint8_t*)(us + 16);             // [RX][D] the imputations' masks
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int b0 = blockIdx.x * a.DP;
  const int nb = min(a.DP, a.B - b0);
  const int live_rows = a.DP * Kp;               // tile rows that map to a proposal (those of datapoints past the batch
                                                 // repeat the tile's first datapoint; their results are never used)
  // this lane's tile row as a proposal, and the encoder row it draws from (a replenish proposal has none: it rides
  // along on a neighbour's and ignores the result)
  const int dpm = m / Kp, km = m - dpm * Kp;
  const int er = (dpm < nb ? dpm : 0) * K + (km < K ? km : K - 1);          // row of xs / ms
  const uint64_t base = a.offset_base ? a.offset_base[0] : 0ull;
  const int64_t NZ = (int64_t)a.B * Kp, NX = (int64_t)a.B * K;
  const bool normal = a.gen_dist <= DMV_GEN_NORMAL_EPS;
  const float geps = a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f;
  TailArgs ta;
  ta.D = D; ta.gen_dist = a.gen_dist; ta.clip_min = a.clip_min; ta.clip_max = a.clip_max; ta.clip_after_blend = a.resample;
  if (!a.resample && tid < 16) pick[tid] = tid;   // (ordered before its first reader by the barriers of the passes below)
  // the tile's imputations and masks: global memory sees them again when the update is over
  const int nx = nb * K * D;                     // (a tile's rows are consecutive in X and M)
  for (int e = tid; e < nx; e += 256) {
    xs[e] = a.X[(int64_t)b0 * K * D + e];
    ms[e] = a.M[(int64_t)b0 * K * D + e];
  }
  __syncthreads();

  // sampling noise of element e of the tile's imputations, noise set `tn`
  auto tail_noise = [&](int tn, int e) {
    const int64_t i = (int64_t)b0 * K * D + e;
    return a.noise ? a.noise[(int64_t)tn * NX * D + i]
                   : normal ? dmv_philox_normal(a.seed, a.off_tail[tn] + base + (uint64_t)i)
                            : dmv_philox_uniform(a.seed, a.off_tail[tn] + base + (uint64_t)i);
  };
  // The noise of an iteration does not depend on anything the iteration computes: the last threads of the workgroup
  // draw it (into nzs) while the first ones are busy with the log-weights, so the tail finds it ready.
  auto draw_tail_noise = [&](int tn) {
    for (int e = 255 - tid; e < nx; e += 256) nzs[e] = tail_noise(tn, e);
  };
  // new imputations of the tile's datapoints from the decoder rows pick[] names; noise from nzs, or set `tn` drawn here
  auto tail = [&](const TailArgs& tt, int tn, bool ready, bool write_out) {
    const int per_dp = K * D;
    for (int e = tid; e < nx; e += 256) {
      const int dp2 = e / per_dp, rem = e - dp2 * per_dp;
      const int k2 = rem / D, d = rem - k2 * D;
      const float* dec = dt + pick[dp2 * K + k2] * ldD;
      const float out = tail_element(tt, dec, d, ready ? nzs[e] : tail_noise(tn, e), xs[e], ms[e] != 0);
      xs[e] = out;
      if (write_out) {
        const int64_t n = (int64_t)(b0 + dp2) * K + k2;
        a.X[n * D + d] = out;
        if (a.buf) a.buf[a.idx_imp[n] * D + d] = out;
      }
    }
  };
  // inverse-CDF draw from softmax of P log-weights l(p), p = 0 .. P-1 (the order of dmv_categorical_draw)
  auto draw = [&](int P, float u, auto l) {
    float mx = -INFINITY;
    for (int p = 0; p < P; ++p) mx = fmaxf(mx, l(p));
    float se = 0.f;
    for (int p = 0; p < P; ++p) se += expf(l(p) - mx);
    const float target = u * se;
    float c = 0.f;
    for (int p = 0; p < P; ++p) {
      c += expf(l(p) - mx);
      if (c > target) return p;
    }
    return P - 1;
  };

  // Every network pass of the update - T encoder and T decoder passes, and the decoder pass of the final resampling -
  // goes through the ONE call of the chain below (one copy of its code in the kernel instead of three).
  const int n_pass = 2 * a.T + (a.final_resample ? 1 : 0);
  for (int pass = 0; pass < n_pass; ++pass) {
    const int t = pass >> 1;
    const bool fin = pass == 2 * a.T;              // the decoder pass of the final resampling
    const bool is_dec = (pass & 1) || fin;
    // (tuning build: phase stamps of the second iteration)
#define LTRACE(i) do { if (t == 1) TRACE(i); } while (0)
    if (fin) {
      // ---- K latents per datapoint from the pooled T Kp proposals, decoded once more; no clamping (:657-666)
      if (tid < nb * K) {
        const int dp2 = tid / K, k2 = tid - dp2 * K;
        const int64_t b2 = b0 + dp2;
        int v;
        if (a.final_draws) {
          v = (int)a.final_draws[b2 * K + k2];
        } else {
          const float u = dmv_philox_uniform(a.seed, a.off_draw[a.T] + base + (uint64_t)(b2 * K + k2));
          v = draw(a.T * Kp, u, [&](int p) { const int tt = p / Kp; return lwall[tt * 16 + dp2 * Kp + (p - tt * Kp)]; });
        }
        const int tt = v / Kp;
        pick[tid] = tt * 16 + dp2 * Kp + (v - tt * Kp);
      }
      __syncthreads();
      for (int idx = tid; idx < 16 * ZP; idx += 256) {
        const int mm = idx / ZP, j = idx - mm * ZP;
        zt[mm * ZP + j] = (mm < nb * K && j < Zd) ? zall[(size_t)pick[mm] * Zd + j] : 0.f;
      }
      __syncthreads();
      if (tid < 16) pick[tid] = tid;               // output row (dp, k) decodes in tile row dp K + k (the pass's barriers
    }                                              // order this after the gather and before the tail)
    if (!is_dec) LTRACE(0);
    // ---- the encoder on the current imputations, or the decoder on the proposals
    {
      const FwdArgs& net = is_dec ? a.dec : a.enc;
      float* orow = is_dec ? dt + m * ldD : et + m * ldE;
      r16_forward_chain<NT, false, false, W>(
          net, lds, lane, wv, 0, 0, false,
          [&](int k0) {
            return is_dec ? *(const f32x4*)(zt + m * ZP + k0) : first_layer_input4_of(a.enc, xs + er * D, ms + er * D, k0);
          },
          [&](int ot, f32x4 o) { *(f32x4*)(orow + 16 * ot + 4 * g) = o; });
      __syncthreads();
    }
    if (fin) {
      TailArgs tf = ta;
      tf.clip_min = nullptr; tf.clip_max = nullptr; tf.clip_after_blend = 0;
      tail(tf, a.T, false, true);
      break;
    }
    if (!is_dec) {
    LTRACE(1);
    // ---- proposals z = mu + sigma eps and the table of the components' Gaussians
    for (int idx = tid; idx < 16 * ZP; idx += 256) {
      const int mm = idx / ZP, j = idx - mm * ZP;
      float z = 0.f;
      if (j < Zd) {
        const int dp2 = mm / Kp, k2 = mm - dp2 * Kp;
        const bool pri = k2 >= K;
        const float mu = pri ? 0.f : et[mm * ldE + j];
        const float sg = dmv_softplus(pri ? a.prior_raw : et[mm * ldE + a.raw_off + j]) + a.q_eps;
        const int64_t i = ((int64_t)(b0 + (dp2 < nb ? dp2 : 0)) * Kp + k2) * Zd + j;
        const float e = a.eps ? a.eps[(int64_t)t * NZ * Zd + i] : dmv_philox_normal(a.seed, a.off_eps[t] + base + (uint64_t)i);
        z = mu + sg * e;
        if (a.resample) {
          const float rs = __builtin_amdgcn_rcpf(sg);
          qt[mm * Zd + j] = QEntry{mu, 0.5f * rs * rs, dmv_log(sg) + LOG_SQRT_2PI, 0.f};
        }
        if (a.final_resample) zall[((size_t)t * 16 + mm) * Zd + j] = z;
      }
      zt[mm * ZP + j] = z;
    }
    __syncthreads();
    LTRACE(2);
    continue;
    }
    LTRACE(3);
    if (!a.resample) {
      tail(ta, t, false, t == a.T - 1);
      __syncthreads();
      continue;
    }
    draw_tail_noise(t);
    // ---- DMIS log-weights: log p(x_o | z) + log p(z) - log (1 / Kp) sum_c q_c(z), the observed entries under the mask of
    // the datapoint's first imputation (:407) and that imputation's values.  A group of 16 lanes per proposal: the lanes
    // share out the dimensions of every sum, so a wavefront runs one short instruction stream for its four proposals.
    {
      const int p = tid >> 4, l16 = tid & 15;
      if (p < live_rows) {
        const int dp2 = p / Kp;
        const float* z = zt + p * ZP;
        float pr = 0.f;
        for (int j = l16; j < Zd; j += 16) { const float zj = z[j]; pr += -(zj * zj) * 0.5f - LOG_SQRT_2PI; }
        pr = gsum<16>(pr);
        const int r0 = (dp2 < nb ? dp2 : 0) * K * D;
        const float* x0 = xs + r0;
        const uint8_t* m0 = ms + r0;
        const float* dec = dt + p * ldD;
        float gm = 0.f;
        for (int d = l16; d < D; d += 16) {
          const float x = x0[d];
          float lp;
          if (normal) lp = log_normal_r(x, dec[d], dmv_softplus(dec[D + d]) + geps);
          else if (a.gen_dist == DMV_GEN_BERN) lp = log_bern(x, dec[d]);
          else lp = log_bern((x + 1.f) * 0.5f, dec[d]);
          gm += lp * (m0[d] ? 1.f : 0.f);          // (the product the reference forms: non-finite under a missing entry -> NaN)
        }
        gm = gsum<16>(gm);
        float mx = -INFINITY, se = 0.f;
        for (int c = 0; c < Kp; ++c) {
          const QEntry* q = qt + (dp2 * Kp + c) * Zd;
          float v = 0.f;
          for (int j = l16; j < Zd; j += 16) v += q_logp(q[j], z[j]);
          v = gsum<16>(v);
          const float nm = fmaxf(mx, v);
          se = se * dmv_exp(mx - nm) + dmv_exp(v - nm);
          mx = nm;
        }
        const float w = gm + pr - (mx + dmv_log(se) - dmv_log((float)Kp));
        if (l16 == 0) {
          lw[p] = w;
          if (a.final_resample) lwall[t * 16 + p] = w;
          if (a.lw_out && dp2 < nb) a.lw_out[((int64_t)t * a.B + b0 + dp2) * Kp + (p - dp2 * Kp)] = w;
        }
      }
      // the resampling uniforms, by the last wavefront (whose proposals are the padding rows of most tiles)
      if (tid >= 192 && tid - 192 < nb * K && !a.draws) {
        const int o = tid - 192, dp2 = o / K;
        us[o] = dmv_philox_uniform(a.seed, a.off_draw[t] + base + (uint64_t)((int64_t)(b0 + dp2) * Kp + (o - dp2 * K)));
      }
    }
    __syncthreads();
    LTRACE(4);
    LTRACE(5);
    // ---- resampling: of the Kp draws per datapoint the first K are kept (:521-524).  A group of 16 lanes per draw: the
    // lanes hold the proposals' probabilities, a prefix sum gives the distribution function
    {
      const int o = tid >> 4, l16 = tid & 15;
      if (o < nb * K) {
        const int dp2 = o / K, kd = o - dp2 * K;
        int pk;
        if (a.draws) {
          pk = (int)a.draws[((int64_t)t * a.B + b0 + dp2) * Kp + kd];
        } else {
          const float lv = l16 < Kp ? lw[dp2 * Kp + l16] : -INFINITY;
          const float mx = group_max<16>(lv);
          float c = l16 < Kp ? expf(lv - mx) : 0.f;
#pragma unroll
          for (int off = 1; off < 16; off <<= 1) {
            const float up = __shfl_up(c, off, 16);
            if (l16 >= off) c += up;
          }
          const float target = us[o] * __shfl(c, 15, 16);
          const unsigned long long hit = __ballot(l16 < Kp && c > target);
          const unsigned mine = (unsigned)(hit >> (16 * (lane >> 4))) & 0xffffu;
          pk = mine ? __ffs(mine) - 1 : Kp - 1;
        }
        if (l16 == 0) pick[o] = dp2 * Kp + pk;
      }
    }
    __syncthreads();
    LTRACE(6);
    tail(ta, t, true, t == a.T - 1 && !a.final_resample);
    __syncthreads();                              // the next encoder pass reads the new imputations
    LTRACE(7);
#undef LTRACE
  }
}

// relu backward on a slice (sign bits of that layer's saved activation), then its columns of the layer-gradient buffer
template <int NT, int W>
__device__ __forceinline__ void r16_mask_store(R16Slice<NT, W>& slice, int lane, int t0, const uint32_t* mask_line,
                                               float scale, float* dact_row, bool st = true) {
  constexpr int TW = Rows16<NT, W>::TW, NK = Rows16<NT, W>::NK;
  const int g = lane >> 4;
#pragma unroll
  for (int i = 0; i < TW; ++i) {
    // register work is unconditional (tiles past NK hold zeros and are never stored): only the store is predicated
    const int t32 = (t0 + i) >> 1;
    const uint32_t half = ((const uint16_t*)mask_line)[t32 < NT ? t32 : NT - 1];
    const uint32_t nib = half >> (8 * (i & 1) + 4 * (g >> 1));        // t0 is even: parity of the tile = parity of i
#pragma unroll
    for (int e = 0; e < 4; ++e) slice[i][e] = ((nib >> e) & 1u) ? slice[i][e] * scale : 0.f;
    if (st && t0 + i < NK) *(f32x4*)(dact_row + 16 * (t0 + i)) = slice[i];
  }
}

// One 16-row tile back through a whole network.  `row` (a valid row of the padded buffers for every lane) addresses
// the saved sign bits and the per-layer gradient buffer, `st` predicates the stores into the latter; gy4(k0) returns
// columns k0 .. k0+3 of this lane's row of the output gradient (zeros for a row that is not there); with want_dx,
// dx4(it, o) receives input-gradient columns 16 it + 4 g .. + 3 of this lane's row.
template <int NT, int W, class Gy4, class Dx4>
__device__ __forceinline__ void r16_backward_chain(const BwdArgs& a, float* lds, int lane, int wv, int64_t row, bool st,
                                                   bool want_dx, Gy4 gy4, Dx4 dx4) {
  constexpr int TW = Rows16<NT, W>::TW, NK = Rows16<NT, W>::NK, Hp = 32 * NT, MW = mask_words(NT);
  const int t0 = wv * TW;
  const int g = lane >> 4;
  const int64_t rp = (a.rows + 31) & ~(int64_t)31;
  const int64_t act_stride = rp * Hp, mask_stride = rp * 2 * MW;
  const int lf = a.L.nl - 1;
  float* dact_row = a.dacts + row * Hp + 4 * g;
  const uint32_t* mask_line = a.masks + (row * 2 + (g & 1)) * MW;
  const float dscale = a.drop_keep ? a.drop_scale : 1.f;
  const uint32_t lo = r16_lane_off(lane);

  f32x4 dA[NK], dT[NK], acc[TW];
  // ---- this wave's tiles of Wf^T gy
  {
    const int S = a.L.S_out;
    const __amdgpu_buffer_rsrc_t img = make_rsrc(a.packed + a.L.wt[lf], (uint32_t)(NT * S) * 1024u);
    uint32_t base[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) { base[i] = r16_tile_off(t0 + i, S); acc[i] = zero4(); }
    for (int t = 0; 2 * t < S; ++t) {
      const bool in_image = 2 * t + (g >> 1) < S;
      // gy rows are pitched to a multiple of 32 columns, so the half tile past an odd group count is readable
      f32x4 b = gy4(16 * t + 4 * g);
      if (!in_image) b = zero4();
      f32x4 w[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        w[i] = buf_load4(img, lo, base[i] + (uint32_t)t * 2048u);
        if (!in_image) w[i] = zero4();
      }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < TW; ++i) acc[i] = MFMA16(w[i][e], b[e], acc[i]);
    }
  }
  r16_mask_store<NT, W>(acc, lane, t0, mask_line + (lf - 1) * mask_stride, 1.f, dact_row + (lf - 1) * act_stride, st);
  r16_exchange<NT, W>(lds, lane, t0, acc, dA);

  for (int blk = a.nb - 1; blk >= 0; --blk) {
    const int l1 = 1 + 2 * blk, l2 = 2 + 2 * blk;
    r16_gemm<NT, W, true, false>((const f32x4*)(a.packed + a.L.wt[l2]), nullptr, lane, t0, dA, acc);
    r16_mask_store<NT, W>(acc, lane, t0, mask_line + l1 * mask_stride, dscale, dact_row + l1 * act_stride, st);
    r16_exchange<NT, W>(lds, lane, t0, acc, dT);
    r16_own_slice<NT, W>(dA, wv, acc);                                      // skip path
    r16_gemm<NT, W, false, false>((const f32x4*)(a.packed + a.L.wt[l1]), nullptr, lane, t0, dT, acc);
    r16_mask_store<NT, W>(acc, lane, t0, mask_line + (l1 - 1) * mask_stride, 1.f, dact_row + (l1 - 1) * act_stride, st);
    r16_exchange<NT, W>(lds, lane, t0, acc, dA);
  }

  if (want_dx) {
    constexpr int S = 4 * NT;
    const __amdgpu_buffer_rsrc_t img = make_rsrc(a.packed + a.L.wt[0], (uint32_t)(a.L.T_in * S) * 1024u);
    for (int it = wv; it < 2 * a.L.T_in; it += W) dx4(it, r16_gemm1<NT>(img, r16_tile_off(it, S), lane, dA, zero4()));
  }
}

template <int NT, int W>
__global__ __launch_bounds__(64 * W, 1) void k_mlp_bwd_rows16(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[Rows16<NT, W>::LDS_BYTES / 4];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int64_t row = a.row_begin + (int64_t)blockIdx.x * 16 + m;
  const bool live = row < a.rows;
  const int64_t rl = live ? row : a.rows - 1;
  // (the per-layer gradient buffer holds whole 32-row tiles, so rows past the batch store too: zeros)
  r16_backward_chain<NT, W>(
      a, lds, lane, wv, row, true, a.dx != nullptr,
      [&](int k0) { const f32x4 b = *(const f32x4*)(a.gy + rl * a.ldgy + k0); return live ? b : zero4(); },
      [&](int it, f32x4 o) { if (live) *(f32x4*)(a.dx + row * a.lddx + 16 * it + 4 * g) = o; });
}

// ------------------------------------------------------------------------------------------
// Backward of a small batch from the decoder outputs' gradient to the encoder's first layer in one launch:
// decoder input-gradient chain -> gradient w.r.t. the encoder outputs (mu, raw sigma) -> encoder input-gradient chain
// (what dmv_mlp_backward(dec), dmv_obj_backward_latent and dmv_mlp_backward(enc) do in three launches with two round
// trips through HBM between them).  For posteriors without a mixture term: every encoder row e (a datapoint, or one of
// its imputations) owns S sample rows n = s BG + e.  A workgroup takes Q = 16 / S encoder rows and their Q S sample rows
// as its decoder tile (row q S + s), so the sum over a row's samples that the latent-side gradient needs stays inside
// the workgroup: the decoder's input gradients go to LDS, the latent-side gradient is formed there (same terms, in the
// same order over s, as the own-Gaussian branch of k_obj_bwd_latent), and its Q rows are the encoder chain's tile (the
// other 16 - Q rows of that tile are idle; the encoder pass costs what a 16-row pass costs either way).  Per-layer
// gradients of both networks go to their workspaces for the weight-gradient GEMMs, which follow as separate launches.
// ------------------------------------------------------------------------------------------
struct VaeBwdArgs {
  BwdArgs dec, enc;
  int BG, S, Zd, Q, G;         // encoder rows, samples per row, latent width, encoder rows per tile, rows per datapoint
  int ldE, raw_off, ZP;
  float q_eps;
  float base_prior, base_qown, kl_coef;
  int own_live, use_zscale;
  const float* enc_out;        // [BG][ldE]
  const float* Z;              // [S BG][ldz]
  const float* eps;
  int64_t ldz;
  const float* wgt;            // nullable [S BG]
  const float* gb;             // nullable [BG / G]
  const float* zscale;         // nullable [S BG]
  const float* kl_wgt;         // nullable [BG]
  const float* gscale;         // nullable device scalar
  float* d_enc;                // [BG][ldE], every entry written
  // when dg_dec is set the kernel forms the gradient w.r.t. the decoder outputs itself (what k_obj_bwd_dec* compute)
  // and writes it to dec.gy for the weight-gradient GEMMs; otherwise it reads dec.gy
  const float* dg_dec;         // decoder outputs [N][dg_ld]
  float* dg_out;               // = dec.gy, writable
  int64_t dg_ld;
  const float* dg_X;           // [BG or BG / G][D]
  const uint8_t* dg_M;
  const float* dg_wgt;         // nullable [N]
  const float* dg_gb;          // nullable [BG / G]
  float dg_base_m, dg_base_f;
  int D, gen_dist, x_per_row;  // x_per_row: X / M have a row per encoder row (else per datapoint)
  int ldD;                     // pitch of dec.gy
  // mixture term (mix != 0): the unit of a tile is a DATAPOINT with its RU = S C sample rows n = (s B + b) C + g and its ER
  // encoder rows (1: the C components share a row, stratified mixture; C: a row per imputation); Q = 16 / RU
  int mix;                     // 0, DMV_Q_STRATIFIED or DMV_Q_IMPUTATION
  int B, C, RU, ER;
  int sg, logit_off;           // stride between components and offset of the logits inside a datapoint's encoder rows
  float base_qmix, base_w;
  int mix_live;
  const float* comp_lp;        // [N][C]
  const float* qmix;           // [N]
  const float* dlogit;         // nullable [B][C]
};

template <int NT>
__global__ __launch_bounds__(256, 1) void k_vae_bwd_rows16(VaeBwdArgs a) {
  constexpr int W = 4;
  __shared__ __attribute__((aligned(16))) float lds[Rows16<NT, W>::LDS_BYTES / 4];
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [16][ZP] decoder input gradients, [16][ldE] encoder output gradients
  float* dzt = tile;
  float* det = tile + 16 * a.ZP;
  float* ddt = det + 16 * a.ldE;                                // [16][ldD] decoder output gradients (with dg_dec)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int e0 = blockIdx.x * a.Q;
  // tile row mm -> sample row n, its datapoint b and the row of X / M it is scored against; false: no such row
  auto sample_row = [&](int mm, int64_t& n, int& b, int64_t& xrow) {
    if (!a.mix) {
      const int qd = mm / a.S, sd = mm - qd * a.S;
      const int e = e0 + qd;
      n = (int64_t)sd * a.BG + e;
      b = e / a.G;
      xrow = a.x_per_row ? e : b;
      return qd < a.Q && e < a.BG;
    }
    const int qd = mm / a.RU, rem = mm - qd * a.RU;
    const int sd = rem / a.C, gc = rem - sd * a.C;
    b = e0 + qd;
    n = ((int64_t)sd * a.B + b) * a.C + gc;
    xrow = a.x_per_row ? (int64_t)b * a.C + gc : b;
    return qd < a.Q && b < a.B;
  };

  // ---- gradient w.r.t. the decoder outputs of the tile's rows:
  //      d_dec[n][c] = gs gb[b] (M_d wgt[n] base_m + wgt[n] base_f) d log p(x_d | z) / d param_c
  if (a.dg_dec) {
    const float gs = a.gscale ? a.gscale[0] : 1.f;
    const bool normal = a.gen_dist <= DMV_GEN_NORMAL_EPS;
    const int P = normal ? 2 * a.D : a.D;
    const float geps = a.gen_dist == DMV_GEN_NORMAL_EPS ? DMV_EPSILON : 0.f;
    for (int i = threadIdx.x; i < 16 * a.ldD; i += 256) {
      const int mm = i / a.ldD, c = i - mm * a.ldD;
      float v = 0.f;
      int64_t n, xrow;
      int b;
      const bool live = sample_row(mm, n, b, xrow);
      if (live && c < P) {
        const int d = c < a.D ? c : c - a.D;
        const float* dec = a.dg_dec + n * a.dg_ld;
        const float gbb = gs * (a.dg_gb ? a.dg_gb[b] : 1.f);
        const float wm = gbb * (a.dg_wgt ? a.dg_wgt[n] : 1.f);
        const float x = a.dg_X[xrow * a.D + d];
        const float coef = (a.dg_M[xrow * a.D + d] ? wm * a.dg_base_m : 0.f) + wm * a.dg_base_f;
        if (normal) {
          const float mu = dec[d], raw = dec[a.D + d];
          const float sg = dmv_softplus(raw) + geps;
          const float rs = __builtin_amdgcn_rcpf(sg);
          const float u = (x - mu) * rs;
          v = c < a.D ? coef * (u * rs) : coef * ((u * u - 1.f) * rs * (raw > 20.f ? 1.f : dmv_sigmoid(raw)));
        } else {
          const float t = a.gen_dist == DMV_GEN_BERN ? x : (x + 1.f) * 0.5f;
          v = coef * (t - dmv_sigmoid(dec[d]));
        }
      }
      ddt[i] = v;
      if (live) a.dg_out[n * a.ldD + c] = v;
    }
    __syncthreads();
  }

  // ---- decoder chain on the tile's sample rows
  {
    int64_t n, xrow_;
    int b_;
    const bool live = sample_row(m, n, b_, xrow_);
    if (!live) n = 0;
    const float* gyr = a.dg_dec ? ddt + m * a.ldD : a.dec.gy + n * a.dec.ldgy;
    r16_backward_chain<NT, W>(
        a.dec, lds, lane, wv, n, live, true,
        [&](int k0) { const f32x4 b = *(const f32x4*)(gyr + k0); return live ? b : zero4(); },
        [&](int it, f32x4 o) { if (16 * it + 4 * g < a.ZP) *(f32x4*)(dzt + m * a.ZP + 16 * it + 4 * g) = o; });
  }
  for (int i = threadIdx.x; i < 16 * a.ldE; i += 256) det[i] = 0.f;
  __syncthreads();

  // ---- gradient w.r.t. (mu, raw sigma) of the tile's encoder rows
  if (!a.mix) {
    const float gs = a.gscale ? a.gscale[0] : 1.f;
    const bool zs_on = a.use_zscale && a.zscale;
    for (int i = threadIdx.x; i < a.Q * a.Zd; i += 256) {
      const int qq = i / a.Zd, j = i - qq * a.Zd;
      const int e = e0 + qq;
      if (e >= a.BG) continue;
      const float gbb = gs * (a.gb ? a.gb[e / a.G] : 1.f);
      const float mu = a.enc_out[(int64_t)e * a.ldE + j];
      const float raw = a.enc_out[(int64_t)e * a.ldE + a.raw_off + j];
      const float sg = dmv_softplus(raw) + a.q_eps;
      float acc_mu = 0.f, acc_sg = 0.f;
      for (int s = 0; s < a.S; ++s) {
        const int64_t n = (int64_t)s * a.BG + e;
        const float w = gbb * (a.wgt ? a.wgt[n] : 1.f);
        const float z = a.Z[n * a.ldz + j];
        float dz = dzt[(qq * a.S + s) * a.ZP + j];
        dz += a.base_prior * w * (-z);
        if (a.base_qown != 0.f) dz += a.base_qown * w * (-(z - mu) / (sg * sg));
        if (zs_on) dz *= a.zscale[n];
        acc_mu += dz;
        acc_sg += dz * a.eps[n * a.ldz + j];
        if (a.own_live && a.base_qown != 0.f) {
          const float df = z - mu;
          acc_mu += a.base_qown * w * (df / (sg * sg));
          acc_sg += a.base_qown * w * (df * df / (sg * sg * sg) - 1.f / sg);
        }
      }
      if (a.kl_coef != 0.f) {
        const float kc = a.kl_coef * gs * (a.kl_wgt ? a.kl_wgt[e] : 1.f);
        acc_mu += kc * mu;
        acc_sg += kc * (sg - 1.f / sg);
      }
      det[qq * a.ldE + j] = acc_mu;
      det[qq * a.ldE + a.raw_off + j] = acc_sg * (raw > 20.f ? 1.f : dmv_sigmoid(raw));
    }
  } else {
    // Mixture term (the arithmetic of k_obj_bwd_latent_mix on the tile's datapoints): per datapoint the components' mu,
    // 1 / sigma^2, 1 / sigma and log mixture weights, and per sample group s the latents, the mixture term's row weights
    // and the responsibilities r[g][c] = pi_c N_c(z_g) / q_mix(z_g) in LDS; accumulators of (mu, raw sigma, logit) too.
    const int C = a.C, Zd = a.Zd, GZ = C * Zd, per = 6 * GZ + 3 * C + C * C;
    float* mx = ddt + (a.dg_dec ? 16 * a.ldD : 0);                 // behind the decoder-output gradients
    const float gs = a.gscale ? a.gscale[0] : 1.f;
    const bool strat = a.mix == DMV_Q_STRATIFIED;
    const int64_t sb = (int64_t)a.ER * a.ldE;                        // a datapoint's encoder rows are consecutive
    auto fld = [&](int q, int f) { return mx + q * per + f * GZ; };  // f = 0..5: mu, 1/sigma^2, 1/sigma, z, acc_mu, acc_sigma
    auto sml = [&](int q, int f) { return mx + q * per + 6 * GZ + f * C; };   // f = 0..2: log pi, row weight, acc_logit
    auto rsp = [&](int q) { return mx + q * per + 6 * GZ + 3 * C; };
    for (int i = threadIdx.x; i < a.Q * GZ; i += 256) {
      const int q = i / GZ, r = i - q * GZ, gp = r / Zd, j = r - gp * Zd;
      const int b = e0 + q;
      if (b >= a.B) continue;
      const float* er = a.enc_out + (int64_t)b * sb + gp * a.sg;
      const float sgm = dmv_softplus(er[a.raw_off + j]) + a.q_eps;
      fld(q, 0)[r] = er[j];
      fld(q, 1)[r] = 1.f / (sgm * sgm);
      fld(q, 2)[r] = 1.f / sgm;
      fld(q, 4)[r] = 0.f;
      fld(q, 5)[r] = 0.f;
    }
    for (int i = threadIdx.x; i < a.Q * C; i += 256) {
      const int q = i / C, c = i - q * C;
      const int b = e0 + q;
      if (b >= a.B) continue;
      float lp = -logf((float)C), al = 0.f;
      if (strat) {
        const float* lg = a.enc_out + (int64_t)b * sb + a.logit_off;
        float lm = -INFINITY;
        for (int k = 0; k < C; ++k) lm = fmaxf(lm, lg[k]);
        float ss = 0.f;
        for (int k = 0; k < C; ++k) ss += expf(lg[k] - lm);
        lp = lg[c] - (lm + logf(ss));
        if (a.dlogit) al = a.base_w * gs * (a.gb ? a.gb[b] : 1.f) * a.dlogit[(int64_t)b * C + c];
      }
      sml(q, 0)[c] = lp;
      sml(q, 2)[c] = al;
    }
    for (int s = 0; s < a.S; ++s) {
      __syncthreads();
      for (int i = threadIdx.x; i < a.Q * GZ; i += 256) {
        const int q = i / GZ, r = i - q * GZ, gp = r / Zd, j = r - gp * Zd;
        const int b = e0 + q;
        if (b >= a.B) continue;
        const int64_t n = ((int64_t)s * a.B + b) * C + gp;
        fld(q, 3)[r] = a.Z[n * a.ldz + j];
      }
      for (int i = threadIdx.x; i < a.Q * C; i += 256) {
        const int q = i / C, c = i - q * C;
        const int b = e0 + q;
        if (b >= a.B) continue;
        const int64_t n = ((int64_t)s * a.B + b) * C + c;
        sml(q, 1)[c] = a.base_qmix * gs * (a.gb ? a.gb[b] : 1.f) * (a.wgt ? a.wgt[n] : 1.f);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < a.Q * C * C; i += 256) {
        const int q = i / (C * C), r = i - q * C * C, gg = r / C, c = r - gg * C;
        const int b = e0 + q;
        if (b >= a.B) continue;
        const int64_t n = ((int64_t)s * a.B + b) * C + gg;
        rsp(q)[r] = expf(sml(q, 0)[c] + a.comp_lp[n * C + c] - a.qmix[n]);
      }
      __syncthreads();
      for (int i = threadIdx.x; i < a.Q * GZ; i += 256) {
        const int q = i / GZ, r = i - q * GZ, gp = r / Zd, j = r - gp * Zd;
        const int b = e0 + q;
        if (b >= a.B) continue;
        const int64_t n = ((int64_t)s * a.B + b) * C + gp;
        const float gbb = gs * (a.gb ? a.gb[b] : 1.f);
        const float w = gbb * (a.wgt ? a.wgt[n] : 1.f);
        const float mu = fld(q, 0)[r], is2 = fld(q, 1)[r], isg = fld(q, 2)[r];
        const float z = fld(q, 3)[r];
        // (1) reparameterised path: only samples drawn from component gp depend on (mu, sigma)_{gp}
        float dz = dzt[(q * a.RU + s * C + gp) * a.ZP + j] + a.base_prior * w * (-z);
        if (a.base_qown != 0.f) dz += a.base_qown * w * (-(z - mu) * is2);
        float t = 0.f;
        for (int c = 0; c < C; ++c) t += rsp(q)[gp * C + c] * (-(z - fld(q, 0)[c * Zd + j]) * fld(q, 1)[c * Zd + j]);
        dz += sml(q, 1)[gp] * t;
        if (a.use_zscale && a.zscale) dz *= a.zscale[n];
        float am = dz, as = dz * a.eps[n * a.ldz + j];
        // (2) direct dependence of log q_own on its parameters
        if (a.own_live && a.base_qown != 0.f) {
          const float df = z - mu;
          am += a.base_qown * w * (df * is2);
          as += a.base_qown * w * (df * df * is2 * isg - isg);
        }
        // (3) direct dependence of log q_mix (evaluated at every sample of the datapoint) on component gp
        if (a.mix_live) {
          for (int gg = 0; gg < C; ++gg) {
            const float wr = sml(q, 1)[gg] * rsp(q)[gg * C + gp];
            const float df = fld(q, 3)[gg * Zd + j] - mu;
            am += wr * (df * is2);
            as += wr * (df * df * is2 * isg - isg);
          }
        }
        fld(q, 4)[r] += am;
        fld(q, 5)[r] += as;
      }
      if (strat && a.mix_live) {
        for (int i = threadIdx.x; i < a.Q * C; i += 256) {
          const int q = i / C, c = i - q * C;
          if (e0 + q >= a.B) continue;
          const float pi = expf(sml(q, 0)[c]);
          float al = 0.f;
          for (int gg = 0; gg < C; ++gg) al += sml(q, 1)[gg] * (rsp(q)[gg * C + c] - pi);
          sml(q, 2)[c] += al;
        }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.Q * GZ; i += 256) {
      const int q = i / GZ, r = i - q * GZ, gp = r / Zd, j = r - gp * Zd;
      const int b = e0 + q;
      if (b >= a.B) continue;
      const float raw = a.enc_out[(int64_t)b * sb + gp * a.sg + a.raw_off + j];
      float am = fld(q, 4)[r], as = fld(q, 5)[r];
      if (a.kl_coef != 0.f) {
        const float kc = a.kl_coef * gs * (a.kl_wgt ? a.kl_wgt[(int64_t)b * C + gp] : 1.f);
        const float isg = fld(q, 2)[r];
        am += kc * fld(q, 0)[r];
        as += kc * (1.f / isg - isg);
      }
      det[q * (int)sb + gp * a.sg + j] = am;
      det[q * (int)sb + gp * a.sg + a.raw_off + j] = as * (raw > 20.f ? 1.f : dmv_sigmoid(raw));
    }
    if (strat)
      for (int i = threadIdx.x; i < a.Q * C; i += 256) {
        const int q = i / C, c = i - q * C;
        if (e0 + q < a.B) det[q * (int)sb + a.logit_off + c] = sml(q, 2)[c];
      }
  }
  __syncthreads();
  // the tile's encoder rows: Q (one per unit), or Q ER (the encoder rows of Q datapoints)
  const int erows = a.mix ? a.Q * a.ER : a.Q;
  const int64_t erow0 = a.mix ? (int64_t)e0 * a.ER : e0;
  const int64_t erow_end = a.mix ? (int64_t)a.B * a.ER : a.BG;
  for (int i = threadIdx.x; i < erows * a.ldE; i += 256) {
    const int qq = i / a.ldE;
    if (erow0 + qq < erow_end) a.d_enc[erow0 * a.ldE + i] = det[i];
  }

  // ---- encoder chain on the tile's encoder rows (the other lanes idle)
  {
    const bool live = m < erows && erow0 + m < erow_end;
    const int64_t row = live ? erow0 + m : 0;
    r16_backward_chain<NT, W>(
        a.enc, lds, lane, wv, row, live, false,
        [&](int k0) { const f32x4 b = *(const f32x4*)(det + m * a.ldE + k0); return live ? b : zero4(); },
        [&](int, f32x4) {});
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
  int bt;              // tiles per block edge (4, or 2 for short batches)
  int nblk, kblk;      // blocks of bt x bt tiles
  int splits;          // row slices of this layer (fewer for layers whose blocks hold fewer tiles)
  int64_t rows_per_split;
  int wave0;           // first wave of this layer; a layer owns nblk*kblk*splits consecutive waves
  int64_t slab_w, slab_b;  // offsets (floats) of this layer's split-0 image inside the slab
  int64_t slab_stride;     // floats per split of this layer
  int64_t rows;            // rows of this layer's network
  float* slab;             // ... and its slab
};

// One launch takes the layers of one network or of two (an encoder and a decoder whose chains ran in one launch):
// every layer carries its own operands, row count and slab.
#define DMV_WGRAD_MAX_LAYERS 24
struct WgradArgs {
  WgradLayer ly[DMV_WGRAD_MAX_LAYERS];
  int nl;
  int total_waves;
};

// One block of TN x TK tiles over rows [r_begin, r_end).  Compile-time tile counts keep the loop free of
// branches.  Operands come through buffer loads whose resource covers exactly this row slice: the row offset is
// a scalar that advances by two rows per step, the per-lane offset never changes, and a row past the end of the
// slice reads as zero (adds nothing), so the loop needs neither vector address arithmetic nor a tail case.  A
// ring of reduction steps (2 rows each) is kept in flight, RING - 1 steps ahead of the MFMAs (one wave per SIMD,
// so latency has to be covered by the wave's own MFMAs).
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

  const uint32_t ybytes = (uint32_t)Ly.ldy * 4u, xbytes = (uint32_t)Ly.ldx * 4u;   // row pitches
  const uint32_t yl = (uint32_t)h * ybytes + 4u * j, xl = (uint32_t)h * xbytes + 4u * j;
  // small blocks issue few MFMAs per step, so they keep more steps in flight to cover the same latency
  constexpr int RING = (TN * TK >= 12) ? 6 : 10;
  auto fma_step = [&](const float (&yy)[TN], const float (&xx)[TK]) {
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      bsum[i] += yy[i];
#pragma unroll
      for (int k = 0; k < TK; ++k) acc[i][k] = MFMA(yy[i], xx[k], acc[i][k]);
    }
  };
  // buffer offsets are 32-bit: a slice longer than 1 GiB of either operand is walked in chunks (even row counts)
  const int64_t chunk_rows = ((int64_t)(1u << 30) / (ybytes > xbytes ? ybytes : xbytes)) & ~(int64_t)1;
  for (int64_t c0 = r_begin; c0 < r_end; c0 += chunk_rows) {
    const int64_t n_rows = (r_end - c0 < chunk_rows) ? r_end - c0 : chunk_rows;
    const __amdgpu_buffer_rsrc_t Yr = make_rsrc(Ly.dY + c0 * Ly.ldy + 32 * nb0, (uint32_t)n_rows * ybytes - 128u * nb0);
    const __amdgpu_buffer_rsrc_t Xr = make_rsrc(Ly.X + c0 * Ly.ldx + 32 * kb0, (uint32_t)n_rows * xbytes - 128u * kb0);
    float ya[RING][TN], xb[RING][TK];
    auto load_step = [&](uint32_t step, float (&yy)[TN], float (&xx)[TK]) {
      const uint32_t yo = 2u * step * ybytes, xo = 2u * step * xbytes;   // scalar
#pragma unroll
      for (int i = 0; i < TN; ++i)
        yy[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Yr, yl + 128u * i, yo, 0));
#pragma unroll
      for (int k = 0; k < TK; ++k)
        xx[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(Xr, xl + 128u * k, xo, 0));
    };
    const uint32_t n_steps = (uint32_t)((n_rows + 1) >> 1);
    const uint32_t n_groups = (n_steps + RING - 1) / RING;   // whole groups: steps past the slice read zeros
    // steps 0 .. RING-2 are requested up front; inside the loop the request for step s + RING - 1 is issued right
    // before the MFMAs of step s (into the slot step s - 1 has just released)
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) {
      load_step(u, ya[u], xb[u]);
      __builtin_amdgcn_sched_barrier(0);   // in step order: the wait counters at the loop head assume it
    }
    uint32_t step = 0;
    for (uint32_t g = 0; g < n_groups; ++g, step += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        load_step(step + u + RING - 1, ya[(u + RING - 1) % RING], xb[(u + RING - 1) % RING]);
        // pin the program order (requests ahead of the MFMAs they hide under)
        __builtin_amdgcn_sched_barrier(0);
        fma_step(ya[u], xb[u]);
      }
    }
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

template <int TN, int MAXB>
__device__ __forceinline__ void wgrad_dispatch_k(int tk, const WgradLayer& Ly, int nb0, int kb0, int64_t r0, int64_t r1,
                                                 float* slab, int lane) {
  if (MAXB <= 2) {
    if (tk == 1) wgrad_block<TN, 1>(Ly, nb0, kb0, r0, r1, slab, lane);
    else wgrad_block<TN, 2>(Ly, nb0, kb0, r0, r1, slab, lane);
    return;
  }
  switch (tk) {
    case 1: wgrad_block<TN, 1>(Ly, nb0, kb0, r0, r1, slab, lane); break;
    case 2: wgrad_block<TN, 2>(Ly, nb0, kb0, r0, r1, slab, lane); break;
    case 3: wgrad_block<TN, 3>(Ly, nb0, kb0, r0, r1, slab, lane); break;
    default: wgrad_block<TN, 4>(Ly, nb0, kb0, r0, r1, slab, lane); break;
  }
}

// One wave of the weight-gradient grid: finds its layer, block and row slice.  MAXB: the largest block edge the launch
// holds (4, or 2 when every layer was planned with 2x2 blocks: a quarter of the accumulators, so two waves fit a SIMD).
template <int MAXB>
__device__ __forceinline__ void wgrad_wave(const WgradArgs& a) {
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
  const int nb0 = (bi / Ly.kblk) * Ly.bt, kb0 = (bi % Ly.kblk) * Ly.bt;
  const int tn = min(Ly.bt, Ly.nt - nb0), tk = min(Ly.bt, Ly.kt - kb0);
  const int64_t r_begin = (int64_t)split * Ly.rows_per_split;
  const int64_t r_end = min(Ly.rows, r_begin + Ly.rows_per_split);
  float* slab = Ly.slab + (int64_t)split * Ly.slab_stride;
  TRACE(0);
  TRACE_REAL(13);
#ifdef DMV_TRACE
  if (lane == 0 && wid < 8192) dmv_trace_buf[16 * wid + 1] = l * 100 + tn * 10 + tk;
#endif
  if (MAXB <= 2) {
    if (tn == 1) wgrad_dispatch_k<1, MAXB>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane);
    else wgrad_dispatch_k<2, MAXB>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane);
  } else {
    switch (tn) {
      case 1: wgrad_dispatch_k<1, MAXB>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane); break;
      case 2: wgrad_dispatch_k<2, MAXB>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane); break;
      case 3: wgrad_dispatch_k<3, MAXB>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane); break;
      default: wgrad_dispatch_k<4, MAXB>(tk, Ly, nb0, kb0, r_begin, r_end, slab, lane); break;
    }
  }
  TRACE(15);
  TRACE_REAL(14);
}

#if DMV_PART(1)
__global__ __launch_bounds__(256, 1) void k_mlp_wgrad(WgradArgs a) { wgrad_wave<4>(a); }
__global__ __launch_bounds__(256, 2) void k_mlp_wgrad_small(WgradArgs a) { wgrad_wave<2>(a); }
#endif

struct ReduceArgs {
  int nl;
  int N[DMV_WGRAD_MAX_LAYERS], K[DMV_WGRAD_MAX_LAYERS], Kp[DMV_WGRAD_MAX_LAYERS];
  int64_t slab_w[DMV_WGRAD_MAX_LAYERS], slab_b[DMV_WGRAD_MAX_LAYERS];
  int64_t first[DMV_WGRAD_MAX_LAYERS + 1];  // prefix sums of N*K + N, each rounded up to a multiple of four
  float* gw[DMV_WGRAD_MAX_LAYERS];
  float* gb[DMV_WGRAD_MAX_LAYERS];
  int splits[DMV_WGRAD_MAX_LAYERS];
  int64_t slab_stride[DMV_WGRAD_MAX_LAYERS];
  const float* slab[DMV_WGRAD_MAX_LAYERS];
  float beta;
};

#if DMV_PART(1)
// One element of a layer's gradient: the splits are added in order (bitwise reproducible), their loads requested
// eight at a time.
__device__ __forceinline__ float reduce_one(const float* __restrict__ sp, int64_t ss, int ns) {
  float s = 0.f;
  int p = 0;
  for (; p + 8 <= ns; p += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = sp[(int64_t)(p + u) * ss];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; p < ns; ++p) s += sp[(int64_t)p * ss];
  return s;
}

// A thread folds four consecutive elements (the index space gives every layer a multiple of four slots: first[]): with
// 16-byte loads and one 16-byte store where the layer's rows and the buffers allow it (input width a multiple of four,
// aligned gradient tensors), else element by element.  Either way every element is the sum of its splits in order.
__global__ void k_wgrad_reduce(ReduceArgs a) {
  const int64_t total4 = a.first[a.nl] >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = q << 2;
    int l = 0;
    while (i >= a.first[l + 1]) ++l;
    const int64_t e = i - a.first[l];
    const int64_t nk = (int64_t)a.N[l] * a.K[l];
    const int64_t left = nk + a.N[l] - e;          // elements of this layer from e on (<= 0: padding slots)
    if (left <= 0) continue;
    const int64_t ss = a.slab_stride[l];
    const int ns = a.splits[l];
    const bool in_w = e < nk;
    bool vec = left >= 4 && (a.K[l] & 3) == 0;
    int64_t src;
    float* dst;
    if (vec) {
      if (in_w) {
        const int n = (int)(e / a.K[l]), k = (int)(e % a.K[l]);
        src = a.slab_w[l] + (int64_t)n * a.Kp[l] + k;
        dst = a.gw[l] + e;
      } else {
        src = a.slab_b[l] + (e - nk);
        dst = a.gb[l] + (e - nk);
      }
      vec = ((((uintptr_t)dst) | ((uintptr_t)(a.slab[l] + src)) | ((uintptr_t)ss * 4u)) & 15) == 0;
    }
    if (vec) {
      const float* __restrict__ sp = a.slab[l] + src;
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int p = 0;
      for (; p + 8 <= ns; p += 8) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(sp + (int64_t)(p + u) * ss);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; p < ns; ++p) s += *(const f32x4*)(sp + (int64_t)p * ss);
      if (a.beta != 0.f) {
        const f32x4 old = *(const f32x4*)dst;
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = a.beta * old[c] + s[c];
      }
      *(f32x4*)dst = s;
      continue;
    }
    const int cnt = left < 4 ? (int)left : 4;
    for (int c = 0; c < cnt; ++c) {
      const int64_t ec = e + c;
      if (ec < nk) {
        const int n = (int)(ec / a.K[l]), k = (int)(ec % a.K[l]);
        src = a.slab_w[l] + (int64_t)n * a.Kp[l] + k;
        dst = a.gw[l] + ec;
      } else {
        src = a.slab_b[l] + (ec - nk);
        dst = a.gb[l] + (ec - nk);
      }
      const float sc = reduce_one(a.slab[l] + src, ss, ns);
      *dst = (a.beta != 0.f) ? a.beta * (*dst) + sc : sc;
    }
  }
}

#endif

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// launchers of the other objects' kernels (see DMV_MLP_PART above)
struct VaeFwdArgs; struct VaeBwdArgs; struct LairArgs;
int dmv_part_fwd_wide(const FwdArgs& a, hipStream_t st);
int dmv_part_bwd_wide(const BwdArgs& a, hipStream_t st);
int dmv_part_dropout_keep_wide(int NT, const DropSrc& d, int nb, int64_t rows, int hidden, uint8_t* keep, hipStream_t st);
int dmv_part_vae_fwd(const VaeFwdArgs& a, size_t dyn_bytes, hipStream_t st);
int dmv_part_vae_bwd(const VaeBwdArgs& a, size_t dyn_bytes, hipStream_t st);
int dmv_part_lair(const LairArgs& a, size_t dyn_bytes, hipStream_t st);

#if DMV_PART(1)
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

extern "C" int dmv_mlp_pack_map(const DmvNet* net, float* packed, DmvPackMap* weights, DmvPackMap* biases) {
  int rc = dmv_check_net(net);
  if (rc) return rc;
  if (!packed || !weights || !biases) return DMV_E_BADARG;
  const DmvLayout L = dmv_layout(net);
  for (int l = 0; l < L.nl; ++l) {
    DmvPackMap w;
    w.image = packed; w.fwd_off = L.w[l]; w.tr_off = L.wt[l];
    w.N = L.N[l]; w.K = L.K[l]; w.S = L.S[l]; w.Sn = L.Sn[l];
    w.is_bias = 0; w.reserved = 0;
    weights[l] = w;
    DmvPackMap b = w;
    b.fwd_off = L.b[l]; b.tr_off = 0; b.K = 1; b.is_bias = 1;
    biases[l] = b;
  }
  return DMV_OK;
}

extern "C" size_t dmv_mlp_saved_floats(const DmvNet* net, int64_t rows) {
  if (dmv_check_net(net) != DMV_OK || rows < 0) return 0;
  const int64_t Hp = dmv_pad32(net->hidden), Ip = dmv_pad32(net->in_dim);
  const int64_t rp = (rows + 31) & ~(int64_t)31;   // whole 32-row wave tiles: the kernels store without predicates
  const int64_t mw = 2 * (((Hp / 32) + 1) / 2);    // sign-bit words per row and layer (two half-rows)
  return (size_t)(rp * Ip + (2 * net->nb + 1) * rp * (Hp + mw));
}

#endif

// How a relu pass over `rows` rows is cut into launches.  The chip holds 1024 one-wave tiles (32 rows), 256
// four-wave tiles or 256 sixteen-row tiles per round, and a round of those takes about 140 / 47 / 28 us at
// hidden 256.  Whole rounds of 1024 tiles go to the one-wave kernels; what is left takes whichever is shortest: one
// more one-wave round, rounds of four-wave tiles, or four-wave rounds with a last partial round of sixteen-row tiles.
enum { FAM_ONE = 0, FAM_COOP = 1, FAM_ROWS16 = 2 };
struct RowSeg { int family; int64_t begin, end; };

// wide (hidden above 256): no one-wave family, and every round costs proportionally more, so the same comparison of
// four-wave rounds against a last round of sixteen-row tiles decides.
static int plan_row_segments(int64_t rows, RowSeg (&seg)[3], bool wide = false) {
  int n = 0;
  if (dmv_use_rows16(rows)) { seg[n++] = RowSeg{FAM_ROWS16, 0, rows}; return n; }
  const int64_t tiles = (rows + 31) / 32;
  const int64_t full = wide ? 0 : (tiles / 1024) * 1024;
  if (full) seg[n++] = RowSeg{FAM_ONE, 0, std::min<int64_t>(rows, full * 32)};
  const int64_t rest = tiles - full;
  if (rest == 0) return n;
  const int64_t b = full * 32;
  const int64_t coop_rounds = (rest + 255) / 256, tail = rest % 256;
  const int64_t cost_one = wide ? INT64_MAX : 140, cost_coop = coop_rounds * 47;
  const int64_t cost_mixed = (tail > 0 && tail <= 128) ? (rest / 256) * 47 + 28 : INT64_MAX;
  if (cost_one <= cost_coop && cost_one <= cost_mixed) {
    if (n && seg[n - 1].family == FAM_ONE) seg[n - 1].end = rows;
    else seg[n++] = RowSeg{FAM_ONE, b, rows};
  } else if (cost_coop <= cost_mixed) {
    seg[n++] = RowSeg{FAM_COOP, b, rows};
  } else {
    const int64_t mid = b + (rest / 256) * 256 * 32;
    if (mid > b) seg[n++] = RowSeg{FAM_COOP, b, mid};
    seg[n++] = RowSeg{FAM_ROWS16, mid, rows};
  }
  return n;
}

template <int NT, bool SAVE, bool DROP>
static void launch_fwd_family(int family, const FwdArgs& a, hipStream_t st) {
  const int64_t n = a.row_end - a.row_begin;
  if (family == FAM_ROWS16)
    hipLaunchKernelGGL((k_mlp_fwd_rows16<NT, SAVE, DROP, 4>), dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, a);
  else if (family == FAM_COOP)
    hipLaunchKernelGGL((k_mlp_fwd_coop<NT, SAVE, DROP>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_mlp_fwd<NT, SAVE, DROP ? FWD_RELU_DROP : FWD_RELU>), dim3((unsigned)((n + 127) / 128)), dim3(256), 0, st, a);
}

template <int NT>
static int launch_fwd(const FwdArgs& a0, hipStream_t st) {
  FwdArgs a = a0;
  const bool drop = a.drop.mode != DROP_OFF;
  if (a.act != DMV_ACT_RELU) {
    a.row_begin = 0; a.row_end = a.rows;
    const unsigned grid = (unsigned)((a.rows + 127) / 128);
    if (a.acts) hipLaunchKernelGGL((k_mlp_fwd<NT, true, FWD_GENERIC>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_mlp_fwd<NT, false, FWD_GENERIC>), dim3(grid), dim3(256), 0, st, a);
    DMV_HIP_CHECK_LAUNCH();
    return DMV_OK;
  }
  RowSeg seg[3];
  const int n = plan_row_segments(a.rows, seg);
  for (int i = 0; i < n; ++i) {
    a.row_begin = seg[i].begin; a.row_end = seg[i].end;
    if (a.acts && drop) launch_fwd_family<NT, true, true>(seg[i].family, a, st);
    else if (a.acts) launch_fwd_family<NT, true, false>(seg[i].family, a, st);
    else if (drop) launch_fwd_family<NT, false, true>(seg[i].family, a, st);
    else launch_fwd_family<NT, false, false>(seg[i].family, a, st);
    DMV_HIP_CHECK_LAUNCH();
  }
  return DMV_OK;
}

template <int NT>
static int launch_bwd(const BwdArgs& a0, hipStream_t st) {
  BwdArgs a = a0;
  if (a.act != DMV_ACT_RELU) {
    a.row_begin = 0; a.row_end = a.rows;
    hipLaunchKernelGGL((k_mlp_bwd<NT, false>), dim3((unsigned)((a.rows + 127) / 128)), dim3(256), 0, st, a);
    DMV_HIP_CHECK_LAUNCH();
    return DMV_OK;
  }
  RowSeg seg[3];
  const int n = plan_row_segments(a.rows, seg);
  for (int i = 0; i < n; ++i) {
    a.row_begin = seg[i].begin; a.row_end = seg[i].end;
    const int64_t m = a.row_end - a.row_begin;
    if (seg[i].family == FAM_ROWS16)
      hipLaunchKernelGGL((k_mlp_bwd_rows16<NT, 4>), dim3((unsigned)((m + 15) / 16)), dim3(256), 0, st, a);
    else if (seg[i].family == FAM_COOP)
      hipLaunchKernelGGL((k_mlp_bwd_coop<NT>), dim3((unsigned)((m + 31) / 32)), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((k_mlp_bwd<NT, true>), dim3((unsigned)((m + 127) / 128)), dim3(256), 0, st, a);
    DMV_HIP_CHECK_LAUNCH();
  }
  return DMV_OK;
}

// hidden widths above 256 (9..16 tiles of 32): two activation sets no longer fit one wave's registers as 32-row
// tiles.  Up to 4096 rows the 16-row kernels (a quarter of Hp floats per lane and set); beyond that rounds of four-wave
// 32-row tiles (one full layer input + the wave's own slice of the residual stream) and a last round of 16-row tiles
template <int NT>
static int launch_fwd_wide(const FwdArgs& a0, hipStream_t st) {
  FwdArgs a = a0;
  const bool drop = a.drop.mode != DROP_OFF;
  RowSeg seg[3];
  int n = 1;
  seg[0] = RowSeg{FAM_ROWS16, 0, a.rows};
  if (!drop) n = plan_row_segments(a.rows, seg, true);     // (generated dropout: sixteen-row kernels at every batch size)
  for (int i = 0; i < n; ++i) {
    a.row_begin = seg[i].begin; a.row_end = seg[i].end;
    const int64_t m = a.row_end - a.row_begin;
    if (seg[i].family == FAM_COOP) {
      const unsigned tiles = (unsigned)((m + 31) / 32);
      if (a.acts) hipLaunchKernelGGL((k_mlp_fwd_coop<NT, true, false>), dim3(tiles), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_mlp_fwd_coop<NT, false, false>), dim3(tiles), dim3(256), 0, st, a);
    } else {
      const unsigned tiles = (unsigned)((m + 15) / 16);
      if (a.acts && drop) hipLaunchKernelGGL((k_mlp_fwd_rows16<NT, true, true, 4>), dim3(tiles), dim3(256), 0, st, a);
      else if (a.acts) hipLaunchKernelGGL((k_mlp_fwd_rows16<NT, true, false, 4>), dim3(tiles), dim3(256), 0, st, a);
      else if (drop) hipLaunchKernelGGL((k_mlp_fwd_rows16<NT, false, true, 4>), dim3(tiles), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((k_mlp_fwd_rows16<NT, false, false, 4>), dim3(tiles), dim3(256), 0, st, a);
    }
    DMV_HIP_CHECK_LAUNCH();
  }
  return DMV_OK;
}

template <int NT>
static int launch_bwd_wide(const BwdArgs& a0, hipStream_t st) {
  BwdArgs a = a0;
  RowSeg seg[3];
  const int n = plan_row_segments(a.rows, seg, true);
  for (int i = 0; i < n; ++i) {
    a.row_begin = seg[i].begin; a.row_end = seg[i].end;
    const int64_t m = a.row_end - a.row_begin;
    if (seg[i].family == FAM_COOP)
      hipLaunchKernelGGL((k_mlp_bwd_coop<NT>), dim3((unsigned)((m + 31) / 32)), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((k_mlp_bwd_rows16<NT, 4>), dim3((unsigned)((m + 15) / 16)), dim3(256), 0, st, a);
    DMV_HIP_CHECK_LAUNCH();
  }
  return DMV_OK;
}

#define DMV_DISPATCH_WIDE(NTV, FN, ...)           \
  switch (NTV) {                                  \
    case 9: return FN<9>(__VA_ARGS__);            \
    case 10: return FN<10>(__VA_ARGS__);          \
    case 11: return FN<11>(__VA_ARGS__);          \
    case 12: return FN<12>(__VA_ARGS__);          \
    case 13: return FN<13>(__VA_ARGS__);          \
    case 14: return FN<14>(__VA_ARGS__);          \
    case 15: return FN<15>(__VA_ARGS__);          \
    case 16: return FN<16>(__VA_ARGS__);          \
    default: return DMV_E_UNSUPPORTED;            \
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

// how the generated keep decisions of dropout probability p are drawn (DropSrc)
static void drop_plan(float p, int32_t* bits, uint32_t* thresh) {
  const double p2 = (double)p * 2.0, p256 = (double)p * 256.0;
  if (p2 == 1.0) { *bits = 1; *thresh = 1u; }
  else if (p256 == (double)(int64_t)p256) { *bits = 8; *thresh = (uint32_t)p256; }
  else { *bits = 16; *thresh = (uint32_t)((double)p * 65536.0 + 0.5); }
}

#if DMV_PART(1)
extern "C" size_t dmv_mlp_dropout_rng_elements(const DmvNet* net, int64_t rows) {
  if (dmv_check_net(net) || rows <= 0 || net->dropout_p <= 0.f) return 0;
  int32_t bits; uint32_t thresh;
  drop_plan(net->dropout_p, &bits, &thresh);
  return (size_t)net->nb * (size_t)rows * 2u * (size_t)drop_calls(bits, dmv_layout(net).NT) * 4u;
}

#endif

template <int NT>
__global__ void k_dropout_keep(DropSrc d, int nb, int64_t rows, int hidden, uint8_t* keep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)nb * rows * 2) return;
  const int h = (int)(i & 1);
  const int64_t rl = (i >> 1) % rows;
  const int blk = (int)((i >> 1) / rows);
  uint32_t kb[mask_words(NT)];
  drop_words<NT>(d, drop_first_call(d), rows, hidden, blk, rl, h, 0, NT, kb);
  uint8_t* out = keep + ((int64_t)blk * rows + rl) * hidden;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int n = 32 * t + 8 * (j >> 2) + 4 * h + (j & 3);
      if (n < hidden) out[n] = (uint8_t)((kb[t >> 1] >> (16 * (t & 1) + j)) & 1u);
    }
}

template <int NT>
static int launch_dropout_keep(const DropSrc& d, int nb, int64_t rows, int hidden, uint8_t* keep, hipStream_t st) {
  const int64_t n = (int64_t)nb * rows * 2;
  hipLaunchKernelGGL((k_dropout_keep<NT>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, nb, rows, hidden, keep);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

static int drop_source(const DmvNet* net, const uint8_t* drop_keep, const DmvDropRng* gen, DropSrc* d) {
  d->mode = DROP_OFF; d->bits = 0; d->thresh = 0; d->keep = nullptr; d->seed = 0; d->q0 = 0; d->offset_base = nullptr;
  if (net->dropout_p <= 0.f || net->nb == 0 || (!drop_keep && !gen)) return DMV_OK;
  if (drop_keep && gen) return DMV_E_BADARG;
  if (drop_keep) { d->mode = DROP_BYTES; d->keep = drop_keep; return DMV_OK; }
  if (net->act != DMV_ACT_RELU) return DMV_E_UNSUPPORTED;     // the backward of other activations reads keep bytes
  if (gen->offset & 3) return DMV_E_BADARG;
  d->mode = DROP_GEN;
  drop_plan(net->dropout_p, &d->bits, &d->thresh);
  d->seed = gen->seed; d->q0 = gen->offset >> 2; d->offset_base = gen->offset_base;
  return DMV_OK;
}

#if DMV_PART(2)
int dmv_part_fwd_wide(const FwdArgs& a, hipStream_t st) { DMV_DISPATCH_WIDE(a.L.NT, launch_fwd_wide, a, st); }
int dmv_part_bwd_wide(const BwdArgs& a, hipStream_t st) { DMV_DISPATCH_WIDE(a.L.NT, launch_bwd_wide, a, st); }
int dmv_part_dropout_keep_wide(int NT, const DropSrc& d, int nb, int64_t rows, int hidden, uint8_t* keep, hipStream_t st) {
  DMV_DISPATCH_WIDE(NT, launch_dropout_keep, d, nb, rows, hidden, keep, st);
}
#endif

#if DMV_PART(1)
extern "C" int dmv_mlp_dropout_keep(const DmvNet* net, int64_t rows, const DmvDropRng* gen, uint8_t* keep, void* stream) {
  int rc = dmv_check_net(net);
  if (rc) return rc;
  if (!gen || !keep || rows < 0 || net->dropout_p <= 0.f) return DMV_E_BADARG;
  if (rows == 0 || net->nb == 0) return DMV_OK;
  DropSrc d;
  rc = drop_source(net, nullptr, gen, &d);
  if (rc) return rc;
  const int NT = dmv_layout(net).NT;
  switch (NT) {
    case 1: return launch_dropout_keep<1>(d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
    case 2: return launch_dropout_keep<2>(d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
    case 3: return launch_dropout_keep<3>(d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
    case 4: return launch_dropout_keep<4>(d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
    case 5: return launch_dropout_keep<5>(d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
    case 6: return launch_dropout_keep<6>(d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
    case 7: return launch_dropout_keep<7>(d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
    case 8: return launch_dropout_keep<8>(d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
    default: return dmv_part_dropout_keep_wide(NT, d, net->nb, rows, net->hidden, keep, (hipStream_t)stream);
  }
}

extern "C" int dmv_mlp_forward(const DmvNet* net, const float* packed, int32_t in_mode, int32_t D, const float* x,
                               int64_t ldx, const uint8_t* m, int64_t ldm, int64_t rows, float* y, int64_t ldy,
                               float* saved, const uint8_t* drop_keep, const DmvDropRng* drop_rng, void* stream) {
  int rc = dmv_check_net(net);
  if (rc) return rc;
  if (!packed || !x || !y || rows < 0) return DMV_E_BADARG;
  if (rows == 0) return DMV_OK;
  if (in_mode < DMV_IN_PLAIN || in_mode > DMV_IN_CONCAT_MASKED) return DMV_E_UNSUPPORTED;
  if (in_mode != DMV_IN_PLAIN && !m) return DMV_E_BADARG;
  const int want_in = (in_mode >= DMV_IN_CONCAT) ? 2 * D : D;
  if (want_in != net->in_dim || ldx < D || (m && ldm < D)) return DMV_E_BADARG;
  if (ldy < dmv_pad32(net->out_dim) || (ldy & 3) || ((uintptr_t)y & 15)) return DMV_E_BADARG;
  if (net->dropout_p > 0.f && net->nb > 0 && !drop_keep && !drop_rng && saved) return DMV_E_BADARG;  // training-mode dropout needs its source
  FwdArgs a;
  rc = drop_source(net, drop_keep, drop_rng, &a.drop);
  if (rc) return rc;
  a.L = dmv_layout(net);
  a.packed = packed;
  a.nb = net->nb; a.in_dim = net->in_dim; a.hidden = net->hidden; a.act = net->act;
  a.drop_scale = 1.f / (1.f - net->dropout_p);
  a.in_mode = in_mode; a.D = D;
  a.x = x; a.ldx = ldx; a.m = m; a.ldm = ldm; a.rows = rows;
  a.y = y; a.ldy = ldy;
  a.xs = saved;
  const int64_t rp = (rows + 31) & ~(int64_t)31;
  a.acts = saved ? saved + rp * dmv_pad32(net->in_dim) : nullptr;
  a.masks = saved ? (uint32_t*)(a.acts + (2 * net->nb + 1) * rp * dmv_pad32(net->hidden)) : nullptr;
  if (a.L.NT > 8) {
    if (a.drop.mode != DROP_OFF && net->act != DMV_ACT_RELU) return DMV_E_UNSUPPORTED;
    return dmv_part_fwd_wide(a, (hipStream_t)stream);
  }
  DMV_DISPATCH_NT(a.L.NT, launch_fwd, a, (hipStream_t)stream);
}
#endif

// ---- encoder -> sample -> decoder in one launch
template <int NT>
static int launch_vae_fwd(const VaeFwdArgs& a, size_t dyn_bytes, hipStream_t st) {
  const unsigned tiles = a.obj_on ? (unsigned)dmv_ceil_div(a.B, 16 / (a.S * a.G)) : (unsigned)((a.N + 15) / 16);
  if (a.mix_C) {
    if (a.dec.acts) hipLaunchKernelGGL((k_vae_fwd_rows16<NT, true, true>), dim3(tiles), dim3(256), dyn_bytes, st, a);
    else hipLaunchKernelGGL((k_vae_fwd_rows16<NT, false, true>), dim3(tiles), dim3(256), dyn_bytes, st, a);
  } else if (a.dec.acts) hipLaunchKernelGGL((k_vae_fwd_rows16<NT, true, false>), dim3(tiles), dim3(256), dyn_bytes, st, a);
  else hipLaunchKernelGGL((k_vae_fwd_rows16<NT, false, false>), dim3(tiles), dim3(256), dyn_bytes, st, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

static bool vae_forward_fits(const DmvObjCfg* cfg, const DmvNet* enc, const DmvNet* dec) {
  // (cfg->G counts prior components too when dmv_vae_forward is given some)
  if (!cfg || dmv_check_net(enc) != DMV_OK || dmv_check_net(dec) != DMV_OK) return false;
  if (cfg->B < 1 || cfg->G < 1 || cfg->nz < 1 || cfg->I < 1 || cfg->Zd < 1) return false;
  const int64_t N = (int64_t)cfg->nz * cfg->I * cfg->B * cfg->G;
  if (N > 4096) return false;                                     // one round of 16-row tiles
  if (enc->act != DMV_ACT_RELU || dec->act != DMV_ACT_RELU) return false;
  const int NTe = dmv_ceil_div(enc->hidden, 32), NTd = dmv_ceil_div(dec->hidden, 32);
  if (NTe != NTd || NTe > 8) return false;                        // one tile shape for both chains
  if (dec->in_dim != cfg->Zd || cfg->Zd > 256) return false;
  const size_t dyn = sizeof(float) * 16 * ((size_t)dmv_pad32(enc->out_dim) + 16 * (size_t)dmv_ceil_div(cfg->Zd, 16));
  return dyn + (size_t)NTe * 2048 + 64 <= 65536;
}

#if DMV_PART(3)
int dmv_part_vae_fwd(const VaeFwdArgs& a, size_t dyn_bytes, hipStream_t st) { DMV_DISPATCH_NT(a.enc.L.NT, launch_vae_fwd, a, dyn_bytes, st); }
#endif

#if DMV_PART(1)
extern "C" int dmv_vae_forward_fits(const DmvObjCfg* cfg, const DmvNet* enc, const DmvNet* dec) {
  return vae_forward_fits(cfg, enc, dec) ? 1 : 0;
}

static int vae_forward_impl(const DmvObjCfg* cfg, const DmvNet* enc, const float* enc_packed, int32_t in_mode,
                            const float* X, const uint8_t* M, const DmvNet* dec, const float* dec_packed,
                            const float* eps, uint64_t seed, uint64_t offset, const uint64_t* offset_base,
                            int32_t n_prior, float prior_raw, float* enc_out, float* enc_saved, float* Z, int64_t ldz,
                            float* eps_out, float* dec_out, float* dec_saved, const DmvObjFwd* obj, void* stream) {
  if (!vae_forward_fits(cfg, enc, dec)) return DMV_E_UNSUPPORTED;
  if (obj) {
    // the bound in the same launch: a logsumexp bound over one Gaussian per datapoint, or over a stratified mixture with
    // its log mixture weights (one sample row per component); a datapoint's rows in one tile
    const int SG = cfg->nz * cfg->I * cfg->G;
    const bool own = cfg->G == 1 && obj->q_kind == DMV_Q_OWN && !obj->use_logpi && !obj->terms && !obj->comp_lp && !obj->dlogit;
    const bool strat = cfg->G > 1 && obj->q_kind == DMV_Q_STRATIFIED && obj->use_logpi && obj->mix_C == cfg->G && !obj->dreg &&
                       cfg->logit_off + cfg->G <= dmv_pad32(enc->out_dim);
    if (cfg->x_per_component || n_prior != 0 || SG > 16 || !(own || strat)) return DMV_E_UNSUPPORTED;
    if (obj->kind != DMV_BOUND_LSE || obj->objA || obj->objB) return DMV_E_UNSUPPORTED;
    if (!obj->bound || !M) return DMV_E_BADARG;
    if (cfg->gen_dist < DMV_GEN_NORMAL || cfg->gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
    if (dec->out_dim != (cfg->gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * cfg->D : cfg->D)) return DMV_E_BADARG;
    const bool want_means = obj->mean_out || obj->neg_mean_out;
    if (want_means && (!obj->ticket || dmv_ceil_div(cfg->B, 16 / SG) > 256)) return DMV_E_UNSUPPORTED;
    const size_t dyn = sizeof(float) * 16 * ((size_t)dmv_pad32(enc->out_dim) + 16 * (size_t)dmv_ceil_div(cfg->Zd, 16) +
                                              (size_t)dmv_pad32(dec->out_dim));
    if (dyn + (size_t)dmv_ceil_div(enc->hidden, 32) * 2048 + 640 > 65536) return DMV_E_UNSUPPORTED;
  }
  if (n_prior < 0 || (n_prior > 0 && (!cfg->x_per_component || n_prior >= cfg->G))) return DMV_E_BADARG;
  if (!enc_packed || !dec_packed || !X || !enc_out || !Z || !dec_out || ldz < cfg->Zd) return DMV_E_BADARG;
  if (in_mode < DMV_IN_PLAIN || in_mode > DMV_IN_CONCAT_MASKED) return DMV_E_UNSUPPORTED;
  if (in_mode != DMV_IN_PLAIN && !M) return DMV_E_BADARG;
  if (((in_mode >= DMV_IN_CONCAT) ? 2 * cfg->D : cfg->D) != enc->in_dim) return DMV_E_BADARG;
  if ((enc_saved != nullptr) != (dec_saved != nullptr)) return DMV_E_BADARG;
  // a training pass (activations saved) of networks with dropout draws keep decisions: the separate launches do that
  if (enc_saved && ((enc->dropout_p > 0.f && enc->nb > 0) || (dec->dropout_p > 0.f && dec->nb > 0))) return DMV_E_UNSUPPORTED;
  if ((((uintptr_t)enc_out | (uintptr_t)dec_out) & 15) != 0) return DMV_E_BADARG;
  const int64_t ldE = dmv_pad32(enc->out_dim);
  const int rows_per_b = cfg->x_per_component ? cfg->G - n_prior : 1;
  const int out_rows_per_b = cfg->x_per_component ? cfg->G : 1;
  // the addressing of (mu, raw sigma) must stay inside the encoder-output row that serves a sample
  const int64_t last = (out_rows_per_b == 1 ? (int64_t)(cfg->G - 1) * cfg->enc_sg : 0) + cfg->raw_off + cfg->Zd;
  if (cfg->enc_sb != out_rows_per_b * ldE || (out_rows_per_b != 1 && cfg->enc_sg != ldE) || last > ldE || cfg->raw_off < cfg->Zd)
    return DMV_E_BADARG;
  if (enc_saved && n_prior > 0) return DMV_E_UNSUPPORTED;       // (prior proposals belong to the samplers: no backward pass)
  VaeFwdArgs a;
  auto fill = [](FwdArgs& f, const DmvNet* net, const float* packed, int64_t rows, float* y, float* saved) {
    f.drop.mode = DROP_OFF; f.drop.bits = 0; f.drop.thresh = 0; f.drop.keep = nullptr; f.drop.seed = 0; f.drop.q0 = 0;
    f.drop.offset_base = nullptr;
    f.L = dmv_layout(net);
    f.packed = packed;
    f.nb = net->nb; f.in_dim = net->in_dim; f.hidden = net->hidden; f.act = net->act;
    f.drop_scale = 1.f;
    f.rows = rows; f.row_begin = 0; f.row_end = rows;
    f.y = y; f.ldy = dmv_pad32(net->out_dim);
    f.xs = saved;
    const int64_t rp = (rows + 31) & ~(int64_t)31;
    f.acts = saved ? saved + rp * dmv_pad32(net->in_dim) : nullptr;
    f.masks = saved ? (uint32_t*)(f.acts + (2 * net->nb + 1) * rp * dmv_pad32(net->hidden)) : nullptr;
  };
  a.N = (int64_t)cfg->nz * cfg->I * cfg->B * cfg->G;
  fill(a.enc, enc, enc_packed, (int64_t)cfg->B * rows_per_b, enc_out, enc_saved);
  a.enc.in_mode = in_mode; a.enc.D = cfg->D; a.enc.x = X; a.enc.ldx = cfg->D; a.enc.m = M; a.enc.ldm = cfg->D;
  fill(a.dec, dec, dec_packed, a.N, dec_out, dec_saved);
  a.dec.in_mode = DMV_IN_PLAIN; a.dec.D = cfg->Zd; a.dec.x = Z; a.dec.ldx = ldz; a.dec.m = nullptr; a.dec.ldm = 0;
  a.B = cfg->B; a.G = cfg->G; a.Zd = cfg->Zd; a.per_comp = cfg->x_per_component ? 1 : 0; a.rows_per_b = rows_per_b;
  a.n_prior = n_prior; a.prior_raw = prior_raw;
  a.q.sb = cfg->enc_sb; a.q.sg = cfg->enc_sg; a.q.raw_off = cfg->raw_off; a.q.logit_off = cfg->logit_off;
  a.q.eps = cfg->q_eps ? DMV_EPSILON : 0.f;
  a.eps = eps; a.eps_out = eps_out; a.Z = Z; a.ldz = ldz; a.seed = seed; a.offset = offset; a.offset_base = offset_base;
  a.ZP = 16 * dmv_ceil_div(cfg->Zd, 16);
  size_t dyn = sizeof(float) * 16 * ((size_t)ldE + (size_t)a.ZP);
  a.obj_on = obj ? 1 : 0;
  a.S = cfg->nz * cfg->I; a.nz = cfg->nz; a.I = cfg->I; a.D = cfg->D; a.gen_dist = cfg->gen_dist;
  a.dreg = 0; a.log_w = nullptr; a.bound = nullptr; a.wgt = nullptr; a.zscale = nullptr; a.ticket = nullptr;
  a.mean = nullptr; a.neg_mean = nullptr;
  a.mix_C = 0; a.terms = nullptr; a.comp_lp = nullptr; a.dlogit = nullptr;
  if (obj) {
    a.dreg = obj->dreg; a.log_w = obj->log_w; a.bound = obj->bound; a.wgt = obj->wgt; a.zscale = obj->zscale;
    if (obj->q_kind == DMV_Q_STRATIFIED) { a.mix_C = obj->mix_C; a.terms = obj->terms; a.comp_lp = obj->comp_lp; a.dlogit = obj->dlogit; }
    const bool want_means = obj->mean_out || obj->neg_mean_out;
    a.ticket = want_means ? (unsigned*)obj->ticket : nullptr;
    a.mean = obj->mean_out; a.neg_mean = obj->neg_mean_out;
    dyn += sizeof(float) * 16 * (size_t)dmv_pad32(dec->out_dim);
  }
  return dmv_part_vae_fwd(a, dyn, (hipStream_t)stream);
}

extern "C" int dmv_vae_forward(const DmvObjCfg* cfg, const DmvNet* enc, const float* enc_packed, int32_t in_mode,
                               const float* X, const uint8_t* M, const DmvNet* dec, const float* dec_packed,
                               const float* eps, uint64_t seed, uint64_t offset, const uint64_t* offset_base,
                               int32_t n_prior, float prior_raw, float* enc_out, float* enc_saved, float* Z, int64_t ldz,
                               float* eps_out, float* dec_out, float* dec_saved, void* stream) {
  return vae_forward_impl(cfg, enc, enc_packed, in_mode, X, M, dec, dec_packed, eps, seed, offset, offset_base, n_prior,
                          prior_raw, enc_out, enc_saved, Z, ldz, eps_out, dec_out, dec_saved, nullptr, stream);
}

extern "C" int dmv_vae_forward_bound(const DmvObjCfg* cfg, const DmvNet* enc, const float* enc_packed, int32_t in_mode,
                                     const float* X, const uint8_t* M, const DmvNet* dec, const float* dec_packed,
                                     const float* eps, uint64_t seed, uint64_t offset, const uint64_t* offset_base,
                                     float* enc_out, float* enc_saved, float* Z, int64_t ldz, float* eps_out, float* dec_out,
                                     float* dec_saved, const DmvObjFwd* obj, void* stream) {
  if (!obj) return DMV_E_BADARG;
  return vae_forward_impl(cfg, enc, enc_packed, in_mode, X, M, dec, dec_packed, eps, seed, offset, offset_base, 0, 0.f,
                          enc_out, enc_saved, Z, ldz, eps_out, dec_out, dec_saved, obj, stream);
}
#endif

// ---- LAIR's imputation update in one launch
template <int NT>
static int launch_lair(const LairArgs& a, size_t dyn_bytes, hipStream_t st) {
  static_assert(sizeof(LairArgs) <= 4096, "kernel arguments");
  const unsigned tiles = (unsigned)((a.B + a.DP - 1) / a.DP);
  hipLaunchKernelGGL((k_lair_rows16<NT>), dim3(tiles), dim3(256), dyn_bytes, st, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

static size_t lair_dyn_bytes(const DmvNet* enc, const DmvNet* dec, int K, int Kp, int D, int Zd, int T, int final_resample) {
  const size_t ldE = (size_t)dmv_pad32(enc->out_dim), ldD = (size_t)dmv_pad32(dec->out_dim), ZP = 16 * (size_t)dmv_ceil_div(Zd, 16);
  size_t fl = 16 * (ldE + ZP + ldD) + 16 * (size_t)Zd * 4 + 16 + 16;
  fl += 16 * (size_t)T;                                       // (the pooled log-weights' slot exists either way)
  if (final_resample) fl += 16 * (size_t)T * Zd;
  const size_t rx = (size_t)(16 / Kp) * (size_t)K;            // imputation rows of a tile: values, noise, uniforms, masks
  fl += 2 * rx * D + 16 + (rx * D + 3) / 4;
  return fl * sizeof(float);
}

static bool lair_fits(const DmvNet* enc, const DmvNet* dec, int B, int K, int n_prior, int T, int D, int Zd, int final_resample) {
  if (dmv_check_net(enc) != DMV_OK || dmv_check_net(dec) != DMV_OK) return false;
  if (B < 1 || K < 1 || n_prior < 0 || T < 1 || D < 1 || Zd < 1) return false;
  if (K + n_prior > 16 || T > DMV_LAIR_MAX_T) return false;   // a datapoint's proposals are rows of one 16-row tile
  if (enc->act != DMV_ACT_RELU || dec->act != DMV_ACT_RELU) return false;
  const int NTe = dmv_ceil_div(enc->hidden, 32), NTd = dmv_ceil_div(dec->hidden, 32);
  if (NTe != NTd || NTe > 8) return false;
  if (dec->in_dim != Zd || enc->out_dim < 2 * Zd || Zd > 256) return false;
  return lair_dyn_bytes(enc, dec, K, K + n_prior, D, Zd, T, final_resample) + (size_t)NTe * 2048 + 64 <= 65536;
}

#if DMV_PART(3)
int dmv_part_lair(const LairArgs& a, size_t dyn_bytes, hipStream_t st) { DMV_DISPATCH_NT(a.enc.L.NT, launch_lair, a, dyn_bytes, st); }
#endif

#if DMV_PART(1)
extern "C" int dmv_lair_update_fits(const DmvNet* enc, const DmvNet* dec, int32_t B, int32_t K, int32_t n_prior, int32_t T,
                                    int32_t D, int32_t Zd, int32_t final_resample) {
  return lair_fits(enc, dec, B, K, n_prior, T, D, Zd, final_resample) ? 1 : 0;
}

static int sampler_update(int resample, const DmvNet* enc, const float* enc_packed, int32_t in_mode, const DmvNet* dec,
                          const float* dec_packed, int32_t B, int32_t K, int32_t n_prior, int32_t T, int32_t D, int32_t Zd,
                          int32_t gen_dist, int32_t q_eps, float prior_raw, int32_t final_resample, float* X,
                          const uint8_t* M, const float* eps, const int64_t* draws, const int64_t* final_draws,
                          const float* noise, uint64_t seed, const uint64_t* off_eps, const uint64_t* off_draw,
                          const uint64_t* off_tail, const uint64_t* offset_base, const float* clip_min,
                          const float* clip_max, float* buf, const int64_t* idx_imp, float* lw_out, void* stream) {
  if (!lair_fits(enc, dec, B, K, n_prior, T, D, Zd, final_resample)) return DMV_E_UNSUPPORTED;
  if (!enc_packed || !dec_packed || !X || !M || (buf && !idx_imp)) return DMV_E_BADARG;
  if (gen_dist < DMV_GEN_NORMAL || gen_dist > DMV_GEN_BERN_M1P1) return DMV_E_UNSUPPORTED;
  if (in_mode < DMV_IN_PLAIN || in_mode > DMV_IN_CONCAT_MASKED) return DMV_E_UNSUPPORTED;
  if (((in_mode >= DMV_IN_CONCAT) ? 2 * D : D) != enc->in_dim) return DMV_E_BADARG;
  if (dec->out_dim != (gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * D : D)) return DMV_E_BADARG;
  if ((!eps && !off_eps) || (!noise && !off_tail)) return DMV_E_BADARG;
  if (resample && (!draws || (final_resample && !final_draws)) && !off_draw) return DMV_E_BADARG;
  LairArgs a;
  auto fill = [](FwdArgs& f, const DmvNet* net, const float* packed, int64_t rows) {
    f.drop.mode = DROP_OFF; f.drop.bits = 0; f.drop.thresh = 0; f.drop.keep = nullptr; f.drop.seed = 0; f.drop.q0 = 0;
    f.drop.offset_base = nullptr;
    f.L = dmv_layout(net);
    f.packed = packed;
    f.nb = net->nb; f.in_dim = net->in_dim; f.hidden = net->hidden; f.act = net->act;
    f.drop_scale = 1.f;
    f.rows = rows; f.row_begin = 0; f.row_end = rows;
    f.y = nullptr; f.ldy = dmv_pad32(net->out_dim);
    f.xs = nullptr; f.acts = nullptr; f.masks = nullptr;
  };
  const int Kp = K + n_prior;
  fill(a.enc, enc, enc_packed, (int64_t)B * K);
  a.enc.in_mode = in_mode; a.enc.D = D; a.enc.x = X; a.enc.ldx = D; a.enc.m = M; a.enc.ldm = D;
  fill(a.dec, dec, dec_packed, (int64_t)B * Kp);
  a.dec.in_mode = DMV_IN_PLAIN; a.dec.D = Zd; a.dec.x = nullptr; a.dec.ldx = 0; a.dec.m = nullptr; a.dec.ldm = 0;
  a.B = B; a.K = K; a.Kp = Kp; a.Zd = Zd; a.D = D; a.T = T; a.DP = 16 / Kp;
  a.gen_dist = gen_dist; a.final_resample = final_resample ? 1 : 0; a.resample = resample;
  a.prior_raw = prior_raw; a.q_eps = q_eps ? DMV_EPSILON : 0.f; a.raw_off = Zd;
  a.ZP = 16 * dmv_ceil_div(Zd, 16); a.ldE = dmv_pad32(enc->out_dim); a.ldD = dmv_pad32(dec->out_dim);
  a.seed = seed;
  for (int t = 0; t <= DMV_LAIR_MAX_T; ++t) {
    if (t < DMV_LAIR_MAX_T) a.off_eps[t] = (off_eps && t < T) ? off_eps[t] : 0;
    a.off_draw[t] = (off_draw && t <= T) ? off_draw[t] : 0;
    a.off_tail[t] = (off_tail && t <= T) ? off_tail[t] : 0;
  }
  a.offset_base = offset_base;
  a.eps = eps; a.draws = draws; a.final_draws = final_draws; a.noise = noise;
  a.M = M; a.X = X; a.clip_min = clip_min; a.clip_max = clip_max; a.buf = buf; a.idx_imp = idx_imp;
  a.lw_out = lw_out;
  const size_t dyn = lair_dyn_bytes(enc, dec, K, Kp, D, Zd, T, final_resample);
  return dmv_part_lair(a, dyn, (hipStream_t)stream);
}

extern "C" int dmv_lair_update(const DmvNet* enc, const float* enc_packed, int32_t in_mode, const DmvNet* dec,
                               const float* dec_packed, int32_t B, int32_t K, int32_t n_prior, int32_t T, int32_t D, int32_t Zd,
                               int32_t gen_dist, int32_t q_eps, float prior_raw, int32_t final_resample, float* X,
                               const uint8_t* M, const float* eps, const int64_t* draws, const int64_t* final_draws,
                               const float* noise, uint64_t seed, const uint64_t* off_eps, const uint64_t* off_draw,
                               const uint64_t* off_tail, const uint64_t* offset_base, const float* clip_min,
                               const float* clip_max, float* buf, const int64_t* idx_imp, float* lw_out, void* stream) {
  return sampler_update(1, enc, enc_packed, in_mode, dec, dec_packed, B, K, n_prior, T, D, Zd, gen_dist, q_eps, prior_raw,
                        final_resample, X, M, eps, draws, final_draws, noise, seed, off_eps, off_draw, off_tail, offset_base,
                        clip_min, clip_max, buf, idx_imp, lw_out, stream);
}

extern "C" int dmv_pgibbs_update_fits(const DmvNet* enc, const DmvNet* dec, int32_t rows, int32_t T, int32_t D, int32_t Zd) {
  return lair_fits(enc, dec, rows, 1, 0, T, D, Zd, 0) ? 1 : 0;
}

extern "C" int dmv_pgibbs_update(const DmvNet* enc, const float* enc_packed, int32_t in_mode, const DmvNet* dec,
                                 const float* dec_packed, int32_t rows, int32_t T, int32_t D, int32_t Zd, int32_t gen_dist,
                                 int32_t q_eps, float* X, const uint8_t* M, const float* eps, const float* noise, uint64_t seed,
                                 const uint64_t* off_eps, const uint64_t* off_tail, const uint64_t* offset_base,
                                 const float* clip_min, const float* clip_max, float* buf, const int64_t* idx_imp,
                                 void* stream) {
  return sampler_update(0, enc, enc_packed, in_mode, dec, dec_packed, rows, 1, 0, T, D, Zd, gen_dist, q_eps, 0.f, 0, X, M, eps,
                        nullptr, nullptr, noise, seed, off_eps, nullptr, off_tail, offset_base, clip_min, clip_max, buf, idx_imp,
                        nullptr, stream);
}

#endif

// ---- backward host side
// Work decomposition of the weight-gradient GEMMs.  A wave owns one block (up to bt x bt tiles) of one layer over
// one row slice; slices are sized so that every wave issues about the same number of MFMAs (a layer whose
// blocks hold c of bt^2 tiles gets slices bt^2/c times longer), and the wave count aims at one full round of the
// 1024 wave slots (256 CUs x 4 SIMDs, one wave per SIMD).  Long batches take 4x4 blocks (each operand row feeds
// four MFMAs).  Up to 16384 rows (24576 at hidden 256) the blocks are 2x2: four times as many blocks fill the chip with a quarter of the
// row slices, and every slice costs one more copy of all gradients written to the slab and read back by the
// reduction (at 2560 rows and hidden 256: 15 slices instead of 40, 33 MB instead of 88 MB each way; 30 us instead
// of 47 us for both kernels).  Measured crossover: 12 k rows at hidden 200; at hidden 256 the 2x2 blocks still win at
// 20480 rows (127 us against 149 us) and lose at 28672 (196 us against 186 us).
// Fills the layers l0 .. l0 + nl - 1 of wa (and ra): l0 > 0 appends a second network to a planned first one.
static int64_t wgrad_plan(const DmvNet* net, int64_t rows, WgradArgs* wa, ReduceArgs* ra, int l0 = 0, double slots = 1024.0,
                          double* weight_out = nullptr) {
  DmvLayout L = dmv_layout(net);
  const int bt = rows <= (L.NT >= 8 ? 24576 : 16384) ? 2 : 4, full = bt * bt;
  double weight = 0.0;   // waves per unit of slices, in units of a full block
  int cost[DMV_MAX_LAYERS];
  for (int l = 0; l < L.nl; ++l) {
    WgradLayer& y = wa->ly[l0 + l];
    y.nt = (L.N[l] + 31) / 32;
    y.kt = (L.K[l] + 31) / 32;
    y.bt = bt;
    y.nblk = (y.nt + bt - 1) / bt;
    y.kblk = (y.kt + bt - 1) / bt;
    cost[l] = (y.nt < bt ? y.nt : bt) * (y.kt < bt ? y.kt : bt);
    if (cost[l] < full / 2) cost[l] = full / 2;   // a small block is bound by load latency, not MFMA issue: at most 2x longer slices
    weight += (double)y.nblk * y.kblk * cost[l] / full;
  }
  if (weight_out) *weight_out = weight;
  int64_t base_splits = (int64_t)(slots / weight);
  if (base_splits < 1) base_splits = 1;
  const int64_t max_splits = (rows + 63) / 64;
  if (base_splits > max_splits) base_splits = max_splits;
  int64_t base_rps = (rows + base_splits - 1) / base_splits;
  int wave0 = l0 ? wa->total_waves : 0;
  int64_t slab = 0;
  for (int l = 0; l < L.nl; ++l) {
    WgradLayer& y = wa->ly[l0 + l];
    int64_t rps = base_rps * full / cost[l];
    rps = (rps + 15) / 16 * 16;
    y.rows_per_split = rps;
    y.splits = (int)((rows + rps - 1) / rps);
    y.wave0 = wave0;
    wave0 += y.nblk * y.kblk * y.splits;
    y.slab_stride = (int64_t)y.nt * 32 * y.kt * 32 + (int64_t)y.nt * 32;
    y.slab_w = slab;
    y.slab_b = slab + (int64_t)y.nt * 32 * y.kt * 32;
    y.rows = rows;
    slab += y.slab_stride * y.splits;
  }
  wa->nl = l0 + L.nl;
  wa->total_waves = wave0;
  if (ra) {
    ra->nl = l0 + L.nl;
    int64_t first = l0 ? ra->first[l0] : 0;
    for (int l = 0; l < L.nl; ++l) {
      const int m = l0 + l;
      ra->N[m] = L.N[l]; ra->K[m] = L.K[l]; ra->Kp[m] = wa->ly[m].kt * 32;
      ra->slab_w[m] = wa->ly[m].slab_w; ra->slab_b[m] = wa->ly[m].slab_b;
      ra->splits[m] = wa->ly[m].splits; ra->slab_stride[m] = wa->ly[m].slab_stride;
      ra->first[m] = first;
      first += ((int64_t)L.N[l] * L.K[l] + L.N[l] + 3) & ~(int64_t)3;   // (k_wgrad_reduce: four slots per thread)
    }
    ra->first[l0 + L.nl] = first;
  }
  return slab;
}

#if DMV_PART(1)
extern "C" size_t dmv_mlp_backward_ws_bytes(const DmvNet* net, int64_t rows) {
  if (dmv_check_net(net) != DMV_OK || rows < 0) return 0;
  if (rows == 0) return 16;
  WgradArgs wa;
  const int64_t slab_floats = wgrad_plan(net, rows, &wa, nullptr);
  const int64_t Hp = dmv_pad32(net->hidden);
  const int64_t dacts = (2 * net->nb + 1) * ((rows + 31) & ~(int64_t)31) * Hp;
  return (size_t)(dacts + slab_floats) * sizeof(float);
}

// Plans a network's weight-gradient GEMMs into the layers l0 .. of wa / ra and points them at what its input-gradient
// chain left in ws.
static int wgrad_bind(const DmvNet* net, const float* saved, const float* gy, int64_t ldgy, int64_t rows,
                      float* const* grad_w, float* const* grad_b, void* ws, size_t ws_bytes, WgradArgs* wa, ReduceArgs* ra,
                      int l0, double slots = 1024.0) {
  const int64_t Hp = dmv_pad32(net->hidden), Ip = dmv_pad32(net->in_dim);
  const int64_t rp = (rows + 31) & ~(int64_t)31;
  const float* xs = saved;
  const float* acts = saved + rp * Ip;
  float* dacts = (float*)ws;
  float* slab = dacts + (2 * net->nb + 1) * rp * Hp;
  const int64_t slab_floats = wgrad_plan(net, rows, wa, ra, l0, slots);
  if (((2 * net->nb + 1) * rp * Hp + slab_floats) * (int64_t)sizeof(float) > (int64_t)ws_bytes) return DMV_E_WORKSPACE;
  const int nl = wa->nl - l0;
  for (int l = 0; l < nl; ++l) {
    if (!grad_w[l] || !grad_b[l]) return DMV_E_BADARG;
    WgradLayer& y = wa->ly[l0 + l];
    if (l == 0) { y.X = xs; y.ldx = Ip; } else { y.X = acts + (int64_t)(l - 1) * rp * Hp; y.ldx = Hp; }
    if (l == nl - 1) { y.dY = gy; y.ldy = ldgy; } else { y.dY = dacts + (int64_t)l * rp * Hp; y.ldy = Hp; }
    y.slab = slab;
    ra->slab[l0 + l] = slab;
    ra->gw[l0 + l] = grad_w[l];
    ra->gb[l0 + l] = grad_b[l];
  }
  return DMV_OK;
}

static int wgrad_launch(const WgradArgs& wa, ReduceArgs& ra, float beta, hipStream_t st) {
  ra.beta = beta;
  // 2x2 blocks only: the kernel built for them (a quarter of the accumulator registers; gas step at B = 4096
  // 0.497 -> 0.485 ms, other steps unchanged; a second wave per SIMD - 2048 wave slots - was measured and gains nothing)
  bool small = true;
  for (int l = 0; l < wa.nl; ++l) small = small && wa.ly[l].bt == 2;
  if (small) hipLaunchKernelGGL(k_mlp_wgrad_small, dim3((unsigned)((wa.total_waves + 3) / 4)), dim3(256), 0, st, wa);
  else hipLaunchKernelGGL(k_mlp_wgrad, dim3((unsigned)((wa.total_waves + 3) / 4)), dim3(256), 0, st, wa);
  DMV_HIP_CHECK_LAUNCH();
  const int64_t total = ra.first[ra.nl] >> 2;
  unsigned rgrid = (unsigned)((total + 255) / 256);
  if (rgrid > 4096) rgrid = 4096;
  hipLaunchKernelGGL(k_wgrad_reduce, dim3(rgrid), dim3(256), 0, st, ra);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

// run_chain: the input-gradient chain (fills the per-layer gradients in ws); grad_w != NULL: the weight-gradient GEMMs
// over what the chain left in ws
static int mlp_backward_impl(const DmvNet* net, const float* packed, const float* saved, const uint8_t* drop_keep,
                             const float* gy, int64_t ldgy, int64_t rows, float* dx, int64_t lddx,
                             float* const* grad_w, float* const* grad_b, float beta, void* ws, size_t ws_bytes,
                             void* stream, bool run_chain) {
  int rc = dmv_check_net(net);
  if (rc) return rc;
  if ((run_chain && !packed) || !saved || !gy || rows < 0 || !ws) return DMV_E_BADARG;
  if ((grad_w == nullptr) != (grad_b == nullptr)) return DMV_E_BADARG;
  if (rows == 0) return DMV_OK;
  if (ldgy < dmv_pad32(net->out_dim) || (ldgy & 3) || ((uintptr_t)gy & 15)) return DMV_E_BADARG;
  if (dx && (lddx < dmv_pad32(net->in_dim) || (lddx & 3) || ((uintptr_t)dx & 15))) return DMV_E_BADARG;
  if (run_chain && net->dropout_p > 0.f && net->nb > 0 && !drop_keep) return DMV_E_BADARG;
  if (drop_keep == DMV_DROP_GENERATED && net->act != DMV_ACT_RELU) return DMV_E_UNSUPPORTED;
  if (ws_bytes < dmv_mlp_backward_ws_bytes(net, rows)) return DMV_E_WORKSPACE;
  if ((uintptr_t)ws & 15) return DMV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t Hp = dmv_pad32(net->hidden), Ip = dmv_pad32(net->in_dim);
  const int64_t rp = (rows + 31) & ~(int64_t)31;
  const float* acts = saved + rp * Ip;
  float* dacts = (float*)ws;

  BwdArgs b;
  b.L = dmv_layout(net);
  b.packed = packed;
  b.nb = net->nb; b.hidden = net->hidden; b.act = net->act;
  b.drop_scale = 1.f / (1.f - net->dropout_p);
  b.acts = acts; b.drop_keep = drop_keep;
  b.masks = (const uint32_t*)(acts + (2 * net->nb + 1) * rp * Hp);
  b.gy = gy; b.ldgy = ldgy; b.rows = rows;
  b.dacts = dacts; b.dx = dx; b.lddx = lddx;
  if (run_chain) {
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
      default: r2 = dmv_part_bwd_wide(b, st);
    }
    if (r2) return r2;
  }
  if (!grad_w) return DMV_OK;

  WgradArgs wa;
  ReduceArgs ra;
  rc = wgrad_bind(net, saved, gy, ldgy, rows, grad_w, grad_b, ws, ws_bytes, &wa, &ra, 0);
  if (rc) return rc;
  return wgrad_launch(wa, ra, beta, st);
}

extern "C" int dmv_mlp_backward(const DmvNet* net, const float* packed, const float* saved, const uint8_t* drop_keep,
                                const float* gy, int64_t ldgy, int64_t rows, float* dx, int64_t lddx,
                                float* const* grad_w, float* const* grad_b, float beta, void* ws, size_t ws_bytes,
                                void* stream) {
  return mlp_backward_impl(net, packed, saved, drop_keep, gy, ldgy, rows, dx, lddx, grad_w, grad_b, beta, ws, ws_bytes,
                           stream, true);
}

extern "C" int dmv_mlp_backward_wgrad(const DmvNet* net, const float* saved, const float* gy, int64_t ldgy, int64_t rows,
                                      float* const* grad_w, float* const* grad_b, float beta, void* ws, size_t ws_bytes,
                                      void* stream) {
  if (!grad_w || !grad_b) return DMV_E_BADARG;
  return mlp_backward_impl(net, nullptr, saved, nullptr, gy, ldgy, rows, nullptr, 0, grad_w, grad_b, beta, ws, ws_bytes,
                           stream, false);
}

static int wgrad_operands_ok(const DmvNet* net, const float* saved, const float* gy, int64_t ldgy, int64_t rows,
                             float* const* grad_w, float* const* grad_b, void* ws, size_t ws_bytes) {
  int rc = dmv_check_net(net);
  if (rc) return rc;
  if (!saved || !gy || rows <= 0 || !ws || !grad_w || !grad_b) return DMV_E_BADARG;
  if (ldgy < dmv_pad32(net->out_dim) || (ldgy & 3) || ((uintptr_t)gy & 15) || ((uintptr_t)ws & 15)) return DMV_E_BADARG;
  if (ws_bytes < dmv_mlp_backward_ws_bytes(net, rows)) return DMV_E_WORKSPACE;
  return DMV_OK;
}

extern "C" int dmv_mlp_backward_wgrad_pair(const DmvNet* net_a, const float* saved_a, const float* gy_a, int64_t ldgy_a,
                                           int64_t rows_a, float* const* grad_w_a, float* const* grad_b_a, void* ws_a,
                                           size_t ws_bytes_a, const DmvNet* net_b, const float* saved_b, const float* gy_b,
                                           int64_t ldgy_b, int64_t rows_b, float* const* grad_w_b, float* const* grad_b_b,
                                           void* ws_b, size_t ws_bytes_b, float beta, void* stream) {
  int rc = wgrad_operands_ok(net_a, saved_a, gy_a, ldgy_a, rows_a, grad_w_a, grad_b_a, ws_a, ws_bytes_a);
  if (rc) return rc;
  rc = wgrad_operands_ok(net_b, saved_b, gy_b, ldgy_b, rows_b, grad_w_b, grad_b_b, ws_b, ws_bytes_b);
  if (rc) return rc;
  const int nl_a = 2 * net_a->nb + 2, nl_b = 2 * net_b->nb + 2;
  if (nl_a + nl_b > DMV_WGRAD_MAX_LAYERS) return DMV_E_UNSUPPORTED;
  WgradArgs wa;
  ReduceArgs ra;
  // the two networks share one round of wave slots in proportion to their MFMA work (two full rounds would queue one
  // behind the other): fewer, longer row slices than in two separate calls, hence sums in a different (fixed) order
  double wt_a = 0, wt_b = 0;
  wgrad_plan(net_a, rows_a, &wa, nullptr, 0, 1024.0, &wt_a);
  wgrad_plan(net_b, rows_b, &wa, nullptr, 0, 1024.0, &wt_b);
  const double ta = wt_a * (double)rows_a, tb = wt_b * (double)rows_b;
  const double sa = 1024.0 * ta / (ta + tb), sb = 1024.0 * tb / (ta + tb);
  rc = wgrad_bind(net_a, saved_a, gy_a, ldgy_a, rows_a, grad_w_a, grad_b_a, ws_a, ws_bytes_a, &wa, &ra, 0, sa);
  if (rc) return rc;
  rc = wgrad_bind(net_b, saved_b, gy_b, ldgy_b, rows_b, grad_w_b, grad_b_b, ws_b, ws_bytes_b, &wa, &ra, nl_a, sb);
  if (rc) return rc;
  return wgrad_launch(wa, ra, beta, (hipStream_t)stream);
}

// ---- decoder chain -> latent-side gradient -> encoder chain in one launch
#endif

template <int NT>
static int launch_vae_bwd(const VaeBwdArgs& a, size_t dyn_bytes, hipStream_t st) {
  const int units = a.mix ? a.B : a.BG;
  const unsigned tiles = (unsigned)((units + a.Q - 1) / a.Q);
  hipLaunchKernelGGL((k_vae_bwd_rows16<NT>), dim3(tiles), dim3(256), dyn_bytes, st, a);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

static bool vae_backward_fits(const DmvObjCfg* cfg, const DmvNet* enc, const DmvNet* dec, bool mixture) {
  if (!vae_forward_fits(cfg, enc, dec)) return false;
  const int S = cfg->nz * cfg->I;
  if (mixture) return S * cfg->G <= 16;                           // a datapoint's samples of all components share a tile
  if (S > 16) return false;                                       // an encoder row's samples share a tile
  if (!cfg->x_per_component && cfg->G != 1) return false;         // one Gaussian per encoder row
  return true;
}

#if DMV_PART(3)
int dmv_part_vae_bwd(const VaeBwdArgs& a, size_t dyn_bytes, hipStream_t st) { DMV_DISPATCH_NT(a.dec.L.NT, launch_vae_bwd, a, dyn_bytes, st); }
#endif

#if DMV_PART(1)
extern "C" int dmv_vae_backward_fits(const DmvObjCfg* cfg, const DmvNet* enc, const DmvNet* dec, int32_t mixture) {
  return vae_backward_fits(cfg, enc, dec, mixture != 0) ? 1 : 0;
}

extern "C" int dmv_vae_backward(const DmvObjCfg* cfg, const DmvLatentBwd* p, const DmvDecGrad* dg, const DmvNet* dec,
                                const float* dec_packed, const float* dec_saved, const uint8_t* dec_drop_keep, float* d_dec, int64_t ld_ddec,
                                void* dec_ws, size_t dec_ws_bytes, const DmvNet* enc, const float* enc_packed,
                                const float* enc_saved, const uint8_t* enc_drop_keep, float* d_enc, void* enc_ws,
                                size_t enc_ws_bytes, const float* enc_out, const float* Z, const float* eps, int64_t ldz,
                                const float* wgt, const float* gb, const float* zscale, const float* kl_wgt,
                                const float* comp_lp, const float* qmix, const float* dlogit, const float* gscale,
                                void* stream) {
  if (!p) return DMV_E_BADARG;
  const bool mixture = p->base_qmix != 0.f;
  if (!vae_backward_fits(cfg, enc, dec, mixture)) return DMV_E_UNSUPPORTED;
  if (mixture && (p->mix_kind != DMV_Q_STRATIFIED && p->mix_kind != DMV_Q_IMPUTATION)) return DMV_E_BADARG;
  if (mixture && (!comp_lp || !qmix)) return DMV_E_BADARG;
  if (mixture && ((p->mix_kind == DMV_Q_IMPUTATION) != (cfg->x_per_component != 0))) return DMV_E_BADARG;
  if (!dec_packed || !dec_saved || !d_dec || !dec_ws || !enc_packed || !enc_saved || !d_enc || !enc_ws || !enc_out || !Z || !eps)
    return DMV_E_BADARG;
  const int64_t ldE = dmv_pad32(enc->out_dim);
  const int rows_per_b = cfg->x_per_component ? cfg->G : 1;
  const int64_t BG = (int64_t)cfg->B * rows_per_b, N = (int64_t)cfg->nz * cfg->I * cfg->B * cfg->G;
  if (ldz < cfg->Zd || ld_ddec < dmv_pad32(dec->out_dim) || (ld_ddec & 3) || ((uintptr_t)d_dec & 15) || ((uintptr_t)d_enc & 15))
    return DMV_E_BADARG;
  if (cfg->enc_sb != rows_per_b * ldE || (rows_per_b != 1 && cfg->enc_sg != ldE) || cfg->raw_off < cfg->Zd ||
      cfg->raw_off + cfg->Zd > ldE)
    return DMV_E_BADARG;
  if (mixture && rows_per_b == 1 &&
      ((int64_t)(cfg->G - 1) * cfg->enc_sg + cfg->raw_off + cfg->Zd > ldE || cfg->logit_off + cfg->G > ldE || cfg->enc_sg < cfg->Zd))
    return DMV_E_BADARG;
  if ((dec->dropout_p > 0.f && dec->nb > 0 && !dec_drop_keep) || (enc->dropout_p > 0.f && enc->nb > 0 && !enc_drop_keep))
    return DMV_E_BADARG;
  if (dec_ws_bytes < dmv_mlp_backward_ws_bytes(dec, N) || enc_ws_bytes < dmv_mlp_backward_ws_bytes(enc, BG)) return DMV_E_WORKSPACE;
  if (((uintptr_t)dec_ws | (uintptr_t)enc_ws) & 15) return DMV_E_BADARG;
  VaeBwdArgs a;
  auto fill = [](BwdArgs& b, const DmvNet* net, const float* packed, const float* saved, const uint8_t* keep, int64_t rows,
                 void* ws) {
    const int64_t Hp = dmv_pad32(net->hidden), Ip = dmv_pad32(net->in_dim);
    const int64_t rp = (rows + 31) & ~(int64_t)31;
    b.L = dmv_layout(net);
    b.packed = packed;
    b.nb = net->nb; b.hidden = net->hidden; b.act = net->act;
    b.drop_scale = 1.f / (1.f - net->dropout_p);
    b.acts = saved + rp * Ip; b.drop_keep = keep;
    b.masks = (const uint32_t*)(b.acts + (2 * net->nb + 1) * rp * Hp);
    b.gy = nullptr; b.ldgy = 0; b.rows = rows; b.row_begin = 0; b.row_end = rows;
    b.dacts = (float*)ws; b.dx = nullptr; b.lddx = 0;
  };
  fill(a.dec, dec, dec_packed, dec_saved, dec_drop_keep, N, dec_ws);
  a.dec.gy = d_dec; a.dec.ldgy = ld_ddec;
  fill(a.enc, enc, enc_packed, enc_saved, enc_drop_keep, BG, enc_ws);
  a.enc.gy = d_enc; a.enc.ldgy = ldE;
  a.BG = (int)BG; a.S = cfg->nz * cfg->I; a.Zd = cfg->Zd; a.Q = 16 / a.S; a.G = rows_per_b;
  a.ldE = (int)ldE; a.raw_off = (int)cfg->raw_off; a.ZP = 16 * dmv_ceil_div(cfg->Zd, 16);
  a.q_eps = cfg->q_eps ? DMV_EPSILON : 0.f;
  a.base_prior = p->base_prior; a.base_qown = p->base_qown; a.kl_coef = p->kl_coef;
  a.own_live = p->own_live; a.use_zscale = p->use_zscale;
  a.enc_out = enc_out; a.Z = Z; a.eps = eps; a.ldz = ldz;
  a.wgt = wgt; a.gb = gb; a.zscale = zscale; a.kl_wgt = kl_wgt; a.gscale = gscale; a.d_enc = d_enc;
  size_t dyn = sizeof(float) * 16 * ((size_t)ldE + (size_t)a.ZP);
  a.mix = mixture ? p->mix_kind : 0;
  a.B = cfg->B; a.C = cfg->G; a.RU = a.S * cfg->G; a.ER = rows_per_b;
  a.sg = (int)cfg->enc_sg; a.logit_off = (int)cfg->logit_off;
  a.base_qmix = p->base_qmix; a.base_w = p->base_w; a.mix_live = p->mix_live;
  a.comp_lp = comp_lp; a.qmix = qmix; a.dlogit = dlogit;
  if (mixture) {
    a.Q = 16 / a.RU;
    const size_t GZ = (size_t)cfg->G * cfg->Zd;
    dyn += sizeof(float) * (size_t)a.Q * (6 * GZ + 3 * (size_t)cfg->G + (size_t)cfg->G * cfg->G);
  }
  a.dg_dec = nullptr; a.dg_out = d_dec; a.dg_ld = 0; a.dg_X = nullptr; a.dg_M = nullptr; a.dg_wgt = nullptr; a.dg_gb = nullptr;
  a.dg_base_m = 0.f; a.dg_base_f = 0.f;
  a.D = cfg->D; a.gen_dist = cfg->gen_dist; a.x_per_row = cfg->x_per_component; a.ldD = (int)ld_ddec;
  if (dg) {
    if (cfg->gen_dist < DMV_GEN_NORMAL || cfg->gen_dist > DMV_GEN_BERN_M1P1 || cfg->D < 1) return DMV_E_UNSUPPORTED;
    const int P = cfg->gen_dist <= DMV_GEN_NORMAL_EPS ? 2 * cfg->D : cfg->D;
    if (!dg->dec || !dg->X || !dg->M || dg->ld_dec < P || ld_ddec < P) return DMV_E_BADARG;
    dyn += sizeof(float) * 16 * (size_t)ld_ddec;
    if (dyn + (size_t)a.dec.L.NT * 2048 + 64 > 65536) return DMV_E_UNSUPPORTED;      // wide decoder rows: form d_dec beforehand
    a.dg_dec = dg->dec; a.dg_ld = dg->ld_dec; a.dg_X = dg->X; a.dg_M = dg->M; a.dg_wgt = dg->wgt; a.dg_gb = dg->gb;
    a.dg_base_m = dg->base_m; a.dg_base_f = dg->base_f;
  }
  if (dyn + (size_t)a.dec.L.NT * 2048 + 64 > 65536) return DMV_E_UNSUPPORTED;
  return dmv_part_vae_bwd(a, dyn, (hipStream_t)stream);
}
#endif
