// Probe (diagnostic, not part of the library): what clock and FLOP rate the chip holds for the trunk convolution's inner
// loop shapes on RANDOM operands -- bare MFMA chains, + LDS fragment reads, + L2 weight-fragment loads, for the two bf16 MFMA
// shapes and 1..3 workgroups per CU.  The conv kernels run at 1.05-1.35 GHz in-kernel (scripts/conv_bench.py STAMPS=1):
// this separates what the MFMAs themselves cost from what their operand traffic costs.
//   hipcc -O3 --offload-arch=gfx950 scripts/probe_mfma_power.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// SHAPE 0: 32x32x16, 5 accumulator tiles, one k-step = 5 MFMAs (160 MFMA cycles), NB activation fragments read per k-step
// SHAPE 1: 16x16x32, 20 accumulator tiles (2 cout x 10 position tiles), one k-step = 20 MFMAs (320 cycles)
// LDSF: activation fragments (1 KB wave reads) per MFMA-32-cycles x 2 (0 = none, 1 = one per 64 cycles, 2 = one per 32 cycles)
// GLW : weight fragment (1 KB) from global/L2 per k-step (0/1)
template <int SHAPE, int LDSF, int GLW, int OCC>
__global__ __launch_bounds__(256, OCC) void probe(const uint4* __restrict__ wsrc, const uint4* __restrict__ xsrc, float* out, int iters,
                                                  unsigned long long* clk) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[48 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 48 * 1024 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = xsrc[(blockIdx.x * 131 + i) & 65535];
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  uint4 w = wsrc[lane], x[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) x[i] = reinterpret_cast<const uint4*>(smem)[i * 64 + lane];
  if constexpr (SHAPE == 0) {
    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (GLW) w = wsrc[((it * 8 + ks) * 64 + lane + blockIdx.x * 64 * 13) & 65535];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          if (LDSF == 2 || (LDSF == 1 && ((ks * 5 + i) & 1) == 0))
            x[i] = reinterpret_cast<const uint4*>(smem)[(((ks * 5 + i) * 64 + lane + it * 17) * 1) & 3071];
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x[i]), acc[i], 0, 0, 0);
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) s += acc[i][q];
    if (s == 12345.678f) out[tid] = s;
  } else {
    f32x4 acc[2][10];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][i][q] = 0.f;
    uint4 w2 = wsrc[64 + lane], xx[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) xx[i] = reinterpret_cast<const uint4*>(smem)[i * 64 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {   // 4 k-steps of 32 = the same K = 128 as 8 steps of 16
        if (GLW) {
          w = wsrc[((it * 8 + ks * 2) * 64 + lane + blockIdx.x * 64 * 13) & 65535];
          w2 = wsrc[((it * 8 + ks * 2 + 1) * 64 + lane + blockIdx.x * 64 * 13) & 65535];
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          if (LDSF == 2 || (LDSF == 1 && (i & 1) == 0))
            xx[i] = reinterpret_cast<const uint4*>(smem)[((ks * 10 + i) * 64 + lane + it * 17) & 3071];
          acc[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, xx[i]), acc[0][i], 0, 0, 0);
          acc[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w2), __builtin_bit_cast(bf16x8, xx[i]), acc[1][i], 0, 0, 0);
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) s += acc[j][i][q];
    if (s == 12345.678f) out[tid] = s;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE, int LDSF, int GLW, int OCC>
static int run(const char* name, const uint4* w, const uint4* x, float* out, unsigned long long* clk, int zero) {
  const int iters = 2000, grid = 256 * OCC;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {   // first launch warms up
    CK(hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL((probe<SHAPE, LDSF, GLW, OCC>), dim3(grid), dim3(256), 0, nullptr, w, x, out, iters, clk);
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid * 2);
  CK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < grid; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
  const double flops = (double)grid * 4 /*waves*/ * iters * 8 * 5 * 32768.0;
  printf("%-44s %s  %7.3f ms  %7.1f TFLOP/s  clock %.2f GHz  MFMA-cycle utilisation %.2f\n", name, zero ? "zeros " : "random", ms,
         flops / ms * 1e-9, cyc / rt * 0.1, (double)iters * 8 * 5 * 32 * OCC / (cyc / grid) );
  return 0;
}

int main() {
  const size_t n = 65536;
  std::vector<uint32_t> hw(n * 4), hx(n * 4);
  uint4 *w, *x; float* out; unsigned long long* clk;
  CK(hipMalloc(&w, n * 16)); CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&clk, 256 * 3 * 16));
  for (int zero = 0; zero < 2; ++zero) {
    uint32_t s = 12345u;
    auto bf = [&](float amp) {   // uniform [-amp, amp) bf16
      s = s * 1664525u + 1013904223u;
      float f = ((float)((s >> 8) & 0xffff) / 32768.f - 1.f) * amp;
      uint32_t u = __builtin_bit_cast(uint32_t, f);
      return (uint32_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    for (auto& v : hw) v = zero ? 0u : (bf(0.05f) | (bf(0.05f) << 16));
    for (auto& v : hx) v = zero ? 0u : (bf(1.f) | (bf(1.f) << 16));
    CK(hipMemcpy(w, hw.data(), n * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), n * 16, hipMemcpyHostToDevice));
    if (run<0, 0, 0, 1>("32x32x16 bare, 1 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<0, 0, 0, 3>("32x32x16 bare, 3 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<0, 2, 0, 1>("32x32x16 + LDS frag per MFMA, 1 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<0, 2, 0, 3>("32x32x16 + LDS frag per MFMA, 3 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<0, 1, 0, 3>("32x32x16 + LDS frag per 2 MFMA, 3 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<0, 2, 1, 3>("32x32x16 + LDS + L2 weights, 3 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<0, 2, 1, 2>("32x32x16 + LDS + L2 weights, 2 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<0, 2, 1, 1>("32x32x16 + LDS + L2 weights, 1 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<0, 1, 1, 3>("32x32x16 + LDS/2 + L2 weights, 3 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<1, 0, 0, 1>("16x16x32 bare, 1 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<1, 0, 0, 3>("16x16x32 bare, 3 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<1, 2, 0, 3>("16x16x32 + LDS frag per 2 MFMA(16c), 3 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<1, 2, 1, 3>("16x16x32 + LDS + L2 weights, 3 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<1, 2, 1, 2>("16x16x32 + LDS + L2 weights, 2 WG/CU", w, x, out, clk, zero)) return 1;
    if (run<1, 1, 1, 3>("16x16x32 + LDS/2 + L2 weights, 3 WG/CU", w, x, out, clk, zero)) return 1;
  }
  return 0;
}
