// Diagnostic (not product): synthetic co-runners for the front-end kernel -- LDS read traffic, MFMA traffic, both, VALU only.
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>   // 1: ds_read_b128 spam, 2: MFMA spam, 3: both, 4: VALU spam
__global__ __launch_bounds__(256) void aggr(float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[49152];
  const int tid = threadIdx.x;
  for (int i = tid; i < 49152 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(i * 2654435761u, i, ~i, i * 40503u);
  __syncthreads();
  f32x16 acc;
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  uint4 a = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), b = a;
  float v = (float)tid;
  unsigned off = (unsigned)tid * 16u;
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint4 x = *reinterpret_cast<const uint4*>(smem + ((off + k * 4096u + (unsigned)it * 272u) % 49152u & ~15u));
        // random, ever-changing operands with sane exponents (bf16 values in [1, 2) with random mantissas and signs)
        a = make_uint4((x.x & 0x807f807fu) | 0x3f803f80u, (x.y & 0x807f807fu) | 0x3f803f80u, (x.z & 0x807f807fu) | 0x3f803f80u, (x.w & 0x807f807fu) | 0x3f803f80u);
        b = make_uint4(a.w ^ 0x00150015u, a.x ^ 0x002a002au, a.y ^ 0x00330033u, a.z ^ 0x004c004cu);
      }
    }
    if (MODE & 2) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
        a.x ^= 0x00010001u * (unsigned)(k + 1); b.z ^= 0x00020002u * (unsigned)(k + 1);
      }
      if ((it & 63) == 63) for (int q = 0; q < 16; ++q) acc[q] *= 1e-3f;   // keep the accumulators finite
    }
    if (MODE & 4) {
#pragma unroll
      for (int k = 0; k < 32; ++k) v = fmaf(v, 1.0001f, 0.5f);
    }
  }
  float s = v + (float)(a.x & 1) + (float)(b.y & 1);
  for (int q = 0; q < 16; ++q) s += acc[q];
  if (s == 12345.678f) sink[0] = s;
}
extern "C" int aggressor_launch(void* stream, int mode, int blocks, int iters, float* sink) {
  hipStream_t st = (hipStream_t)stream;
  if (mode == 1) hipLaunchKernelGGL(aggr<1>, dim3(blocks), dim3(256), 0, st, sink, iters);
  else if (mode == 2) hipLaunchKernelGGL(aggr<2>, dim3(blocks), dim3(256), 0, st, sink, iters);
  else if (mode == 3) hipLaunchKernelGGL(aggr<3>, dim3(blocks), dim3(256), 0, st, sink, iters);
  else hipLaunchKernelGGL(aggr<4>, dim3(blocks), dim3(256), 0, st, sink, iters);
  return (int)hipGetLastError();
}
