// Counter-based RNG for the device samplers: Philox4x32-10 (Salmon et al., SC'11) keyed by the 64-bit
// seed.  One call yields four 32-bit words = the draws of four consecutive elements: element idx uses
// counter idx >> 2 (normals: Box-Muller on word pairs, cos for even idx, sin for odd).  The value of an
// element depends only on (seed, idx), not on the launch geometry or on whether a kernel asks for one
// element or for an aligned group of four, so a backward pass can regenerate the same noise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct dmv_u4 { uint32_t x, y, z, w; };

__host__ __device__ inline dmv_u4 dmv_philox4x32_10(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  return dmv_u4{c0, c1, c2, c3};
}

// uniform in (0, 1]: never 0, so log() is finite
__host__ __device__ inline float dmv_u01(uint32_t w) { return ((float)(w >> 8) + 1.0f) * (1.0f / 16777216.0f); }

struct dmv_f4 { float v[4]; };

// standard normals for elements 4 q .. 4 q + 3 of stream `seed`
__device__ inline dmv_f4 dmv_philox_normal4(uint64_t seed, uint64_t q) {
  const dmv_u4 r = dmv_philox4x32_10(seed, q, 0x6e6f726dull /* "norm" */);
  // Box-Muller; the angle goes to the hardware sine / cosine, which take their argument in revolutions
  const float r0 = __builtin_amdgcn_sqrtf(-2.f * __logf(dmv_u01(r.x))), r1 = __builtin_amdgcn_sqrtf(-2.f * __logf(dmv_u01(r.z)));
  const float a0 = dmv_u01(r.y), a1 = dmv_u01(r.w);
  return dmv_f4{{r0 * __builtin_amdgcn_cosf(a0), r0 * __builtin_amdgcn_sinf(a0), r1 * __builtin_amdgcn_cosf(a1),
                 r1 * __builtin_amdgcn_sinf(a1)}};
}
__device__ inline float dmv_philox_normal(uint64_t seed, uint64_t idx) {
  return dmv_philox_normal4(seed, idx >> 2).v[idx & 3];
}

// uniforms in [0, 1) for elements 4 q .. 4 q + 3
__device__ inline dmv_f4 dmv_philox_uniform4(uint64_t seed, uint64_t q) {
  const dmv_u4 r = dmv_philox4x32_10(seed, q, 0x756e6966ull /* "unif" */);
  const float k = 1.0f / 16777216.0f;
  return dmv_f4{{(float)(r.x >> 8) * k, (float)(r.y >> 8) * k, (float)(r.z >> 8) * k, (float)(r.w >> 8) * k}};
}
__device__ inline float dmv_philox_uniform(uint64_t seed, uint64_t idx) {
  return dmv_philox_uniform4(seed, idx >> 2).v[idx & 3];
}
