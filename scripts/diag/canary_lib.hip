// Diagnostic (not product): LDS canaries as a tiny shared library for ctypes -- see lds_canary.hip.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(64) void canary_k(unsigned long long* errors, int iters, int words) {
  extern __shared__ volatile unsigned buf[];
  const unsigned tag = 0xC0DE0000u ^ (blockIdx.x * 2654435761u);
  for (int i = threadIdx.x; i < words; i += 64) buf[i] = tag + i;
  __syncthreads();
  unsigned long long bad = 0;
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_sleep(64);
    for (int i = threadIdx.x; i < words; i += 64) {
      const unsigned v = buf[i];
      if (v != tag + i) { ++bad; buf[i] = tag + i; }
    }
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
}
// four independent waves per workgroup, each with its own quarter of the workgroup's LDS and NO barrier anywhere (the structure of
// stft_power_fft_kernel): a wave writes its pattern, idles, re-reads
__global__ __launch_bounds__(256) void canary4_k(unsigned long long* errors, int iters, int words_per_wave) {
  extern __shared__ volatile unsigned buf[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  volatile unsigned* mine = buf + wave * words_per_wave;
  const unsigned tag = 0xC0DE0000u ^ ((blockIdx.x * 4 + wave) * 2654435761u);
  unsigned long long bad = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = lane; i < words_per_wave; i += 64) mine[i] = tag + i + it;
    if (wave & 1) __builtin_amdgcn_s_sleep(40); else __builtin_amdgcn_s_sleep(24);
    for (int i = lane; i < words_per_wave; i += 64) {
      const unsigned v = mine[i];
      if (v != tag + i + it) ++bad;
    }
  }
  if (bad) atomicAdd(errors, bad);
}
extern "C" int canary_launch(void* stream, unsigned long long* d_err, int blocks, int iters, int lds_bytes) {
  hipLaunchKernelGGL(canary_k, dim3(blocks), dim3(64), lds_bytes, (hipStream_t)stream, d_err, iters, lds_bytes / 4);
  return (int)hipGetLastError();
}
extern "C" int canary4_launch(void* stream, unsigned long long* d_err, int blocks, int iters, int lds_bytes) {
  hipLaunchKernelGGL(canary4_k, dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, d_err, iters, lds_bytes / 16);
  return (int)hipGetLastError();
}
