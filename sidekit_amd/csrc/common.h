// Shared device/host helpers for the sidekit_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>

#include "../../include/sidekit_amd.h"
#include <stdlib.h>

// A/B tuning switches.  The product library reads exactly two environment variables (SIDEKIT_AMD_LANES, SIDEKIT_AMD_SMALL_GRID); every other
// SIDEKIT_AMD_* switch, the alternative convolution shapes of sk_bench_conv and the in-convolution SE-gate forms (se_gate_inl.h) exist only in the
// A/B build of the same sources (`make ab` -> libsidekit_amd_ab.so, -DSK_AB), which the tests that compare a product path with its A/B partner load
// in a child process (tests/test_gpu_01_ab_variant.py).
#ifdef SK_AB
#define SK_AB_GETENV(name) getenv(name)
#define SK_AB_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define SK_AB_GETENV(name) ((const char*)nullptr)
#define SK_AB_ENV_INT(name, dflt) (dflt)
#endif

namespace sk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- error plumbing (no exception crosses the C ABI) -------------------------------------
// error classes SK_OK / SK_E* come from the public header
void set_error(const char* fmt, ...);
const char* last_error();

#define SK_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      sk::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SK_EHIP;                                                                     \
    }                                                                                         \
  } while (0)

#define SK_CHECK(cond, code, ...)   \
  do {                              \
    if (!(cond)) {                  \
      sk::set_error(__VA_ARGS__);   \
      return code;                  \
    }                               \
  } while (0)

#define SK_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != 0) return rc_;   \
  } while (0)

// ---- bf16 <-> f32 ----------------------------------------------------------------------
struct bf16_t { uint16_t v; };

__host__ __device__ inline float bf16_to_f32(uint16_t b) {
  uint32_t u = ((uint32_t)b) << 16;
  return __builtin_bit_cast(float, u);
}
// round-to-nearest-even; NaN stays NaN (quiet)
__host__ __device__ inline uint16_t f32_to_bf16(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
// ReLU that lets NaN through like torch.relu (fmaxf would swallow it): IEEE-754-2019 maximum = one v_maximum3_f32 on gfx950
__device__ inline float relu_nan(float x) { return __builtin_elementwise_maximum(x, 0.f); }
// device-side conversions use the hardware v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN stays NaN): the integer
// formulation above costs ~8 VALU per value, which made the conv epilogues VALU-bound (measured with in-kernel stamps)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ inline float round_bf16(float x) { return (float)(__bf16)x; }

template <typename T> struct elem;
template <> struct elem<float> {
  static constexpr int bytes = 4;
  __device__ static inline float load(const float* p) { return *p; }
};
template <> struct elem<bf16_t> {
  static constexpr int bytes = 2;
};

// rows after n stride-2 (k3, p1) convolutions: h -> floor((h-1)/2)+1 = ceil(h/2)
__host__ __device__ inline int halve(int h, int n) {
  for (int i = 0; i < n; ++i) h = (h + 1) >> 1;
  return h;
}

// per-utterance feature frames: `frames` may be null (uniform batch)
struct Lens {
  const int* frames;  // device, [B] or nullptr
  int uniform;        // used when frames == nullptr
  __device__ inline int get(int b) const { return frames ? frames[b] : uniform; }
  // the same for a wave-uniform b, through the scalar cache (the array is not written while a kernel runs).  A persistent workgroup that
  // reads its next tile's length with a VECTOR load waits -- vmcnt(0), stores count on it on gfx9 -- for every store of the tile it has
  // just finished before it can issue the next tile's DMA.
  __device__ inline int get_uniform(int b) const {
    return frames ? *reinterpret_cast<const __attribute__((address_space(4))) int*>((unsigned long long)(frames + b)) : uniform;
  }
};

// Sum over each 32-lane half of the wave on the VALU's DPP path (no LDS crossbar traffic, 5 adds): xor-1 and xor-2 inside
// quads, half-mirror and mirror inside the 16-lane rows, then row_bcast:15 carries row 0 / row 2's sum into row 1 / row 3.
// The result is valid in lanes 16..31 (sum of lanes 0..31) and 48..63 (sum of lanes 32..63) only.
__device__ inline float half_sum_upper_row(float v) {
  auto dpp = [](float x, auto ctrl, auto row_mask) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false));
  };
  using std::integral_constant;
  v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
  v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
  v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});   // row_half_mirror
  v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});   // row_mirror
  v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});   // row_bcast:15 into rows 1 and 3 (others add 0)
  return v;
}

// Sum over each 16-lane row of the wave (the first four DPP steps of half_sum_upper_row): valid in every lane of the row.
__device__ inline float row_sum16(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  using std::integral_constant;
  v += dpp(v, integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, integral_constant<int, 0x140>{});   // row_mirror
  return v;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

}  // namespace sk
