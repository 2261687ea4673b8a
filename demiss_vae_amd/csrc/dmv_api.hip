// Version and error strings of the C ABI.
#include "dmv_common.h"

extern "C" int dmv_version(void) { return DMV_VERSION; }

extern "C" const char* dmv_strerror(int rc) {
  switch (rc) {
    case DMV_OK: return "ok";
    case DMV_E_BADARG: return "dmv: bad argument (null pointer, negative size, or inconsistent dimensions / pitches)";
    case DMV_E_UNSUPPORTED: return "dmv: unsupported value (enum out of range, hidden > 512 or > 256 with a non-relu activation, nb > 8, more than 64 mixture components, or a shape beyond an entry point's stated limits)";
    case DMV_E_WORKSPACE: return "dmv: workspace too small";
    default: break;
  }
  if (rc > 0) return hipGetErrorString((hipError_t)rc);
  return "dmv: unknown error";
}

// ------------------------------------------------------------------------------------------ shader clock
// One wave that reads the shader-clock counter (s_memtime: one tick per shader cycle) and the constant-rate counter
// (s_memrealtime) when it starts, naps until the constant-rate counter has advanced by `ticks`, and reads both again.
// Launched on a side stream while the kernels of interest run on another, the two differences give the clock the
// chip actually held under that load: (c1 - c0) / (r1 - r0) x wall-clock rate.  Both exits of the loop are reached
// by the wave whatever the counters do (tick limit or iteration limit); the stamps go out through vector stores.
__global__ void __launch_bounds__(64) k_clock_watch(long long ticks, unsigned long long* out) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r = r0;
  for (int it = 0; it < (1 << 20) && (long long)(r - r0) < ticks; ++it) {
    __builtin_amdgcn_s_sleep(64);
    r = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[0] = c0;
    out[1] = r0;
    out[2] = c1;
    out[3] = r;
  }
}

extern "C" int dmv_clock_watch(int64_t ticks, uint64_t* out4, void* stream) {
  if (!out4 || ticks < 1 || ticks > (int64_t)1 << 28) return DMV_E_BADARG;
  k_clock_watch<<<1, 64, 0, (hipStream_t)stream>>>((long long)ticks, (unsigned long long*)out4);
  DMV_HIP_CHECK_LAUNCH();
  return DMV_OK;
}

extern "C" int dmv_wall_clock_khz(int32_t* khz) {
  if (!khz) return DMV_E_BADARG;
  int dev = 0, v = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, dev);
  if (e != hipSuccess) return (int)e;
  *khz = v;
  return DMV_OK;
}
