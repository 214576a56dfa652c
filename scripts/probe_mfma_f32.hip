// Probe (diagnostic): clock and FLOP rate of bare exact-f32 MFMA chains on random operands, 32x32x2 vs 16x16x4 -- the bf16
// shapes differ by 26 % in the clock the chip grants (probe_mfma_power.hip); does the f32 pair?
//   hipcc -O3 --offload-arch=gfx950 scripts/probe_mfma_f32.hip -o /tmp/probe32 && /tmp/probe32
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* out, int iters, unsigned long long* clk) {
  const int tid = threadIdx.x, lane = tid & 63;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = src[(lane * 8 + i + blockIdx.x * 37) & 65535]; b[i] = src[(lane * 8 + i + 4096 + blockIdx.x * 91) & 65535]; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  if constexpr (SHAPE == 0) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + i) & 7], b[k], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) s += acc[i][q];
  } else {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + i) & 7], b[(k + (i >> 2)) & 7], acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) s += acc[i][q];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (s == 12345.678f) out[tid] = s;
  if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
int run(const char* name, const float* src, float* out, unsigned long long* clk, int wgs_per_cu) {
  const int iters = 4000, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<SHAPE>, dim3(grid), dim3(256), 0, 0, src, out, iters, clk);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid * 2);
  CK(hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost));
  double cyc = 0, rt = 0;
  for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
  const double flops = (double)grid * 4 * iters * 32.0 * 4096.0;   // 32 MFMAs of 4096 FLOP (or 64 of 2048) per iteration per wave
  printf("%-28s %d WG/CU: %.3f ms  %.1f TFLOP/s  in-kernel clock %.2f GHz\n", name, wgs_per_cu, ms, flops / ms * 1e-9, cyc / rt * 0.1);
  return 0;
}

int main() {
  std::vector<float> h(65536);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 22)); }
  float *src, *out; unsigned long long* clk;
  CK(hipMalloc(&src, 65536 * 4)); CK(hipMalloc(&out, 1024 * 4)); CK(hipMalloc(&clk, 4096 * 16));
  CK(hipMemcpy(src, h.data(), 65536 * 4, hipMemcpyHostToDevice));
  for (int w = 1; w <= 2; ++w) {
    if (run<0>("v_mfma_f32_32x32x2_f32", src, out, clk, w)) return 1;
    if (run<1>("v_mfma_f32_16x16x4_f32", src, out, clk, w)) return 1;
  }
  std::vector<float> z(65536, 0.f);
  CK(hipMemcpy(src, z.data(), 65536 * 4, hipMemcpyHostToDevice));
  printf("zero operands:\n");
  if (run<0>("v_mfma_f32_32x32x2_f32", src, out, clk, 2)) return 1;
  if (run<1>("v_mfma_f32_16x16x4_f32", src, out, clk, 2)) return 1;
  return 0;
}
