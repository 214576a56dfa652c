// Diagnostic (not product): does a trunk convolution kernel write LDS outside its own allocation?  A canary kernel keeps a pattern in
// its 16 KB of LDS on every CU (several workgroups per CU, running beside whatever else is on the chip) and re-checks it for a few
// milliseconds while sk_bench_conv launches one convolution shape on the default stream.  hipcc lds_canary.hip -L../../sidekit_amd/csrc -lsidekit_amd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/sidekit_amd.h"

__global__ __launch_bounds__(64) void canary(unsigned long long* errors, int iters) {
  __shared__ volatile unsigned buf[4096];   // 16 KB
  const unsigned tag = 0xC0DE0000u ^ (blockIdx.x * 2654435761u);
  for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = tag + i;
  __syncthreads();
  unsigned long long bad = 0;
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_s_sleep(64);
    for (int i = threadIdx.x; i < 4096; i += 64) {
      const unsigned v = buf[i];
      if (v != tag + i) { ++bad; buf[i] = tag + i; }
    }
    __syncthreads();
  }
  if (bad) atomicAdd(errors, bad);
}

int main(int argc, char** argv) {
  unsigned long long* d_err; hipMalloc(&d_err, 8);
  hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  const int Ts[11] = {401, 401, 401, 401, 201, 201, 201, 101, 101, 101, 51};
  const char* names[11] = {"L1", "L1S", "L2A", "L2S", "L2", "L3A", "L3S", "L3", "L4A", "L4S", "L4"};
  for (int dt = 1; dt >= 0; --dt)
    for (int shape = 0; shape < 11; ++shape)
      for (int variant : {0, 8, 16}) {
        hipMemset(d_err, 0, 8);
        hipLaunchKernelGGL(canary, dim3(256 * 6), dim3(64), 0, s2, d_err, 4000);
        float ms = 0;
        int rc = sk_bench_conv(shape, dt ? XT_BF16 : XT_F32, B, Ts[shape], 3, variant, &ms, nullptr);
        hipStreamSynchronize(s2);
        unsigned long long e = 0; hipMemcpy(&e, d_err, 8, hipMemcpyDeviceToHost);
        if (rc != 0) { continue; }
        printf("%s %-4s variant %2d: %8.1f us/launch, canary words overwritten: %llu\n", dt ? "bf16" : "fp32", names[shape], variant, ms * 1e3, e);
        fflush(stdout);
      }
  return 0;
}
