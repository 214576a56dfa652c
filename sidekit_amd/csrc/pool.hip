// Temporal statistics kernels: mean/std pooling (pooling.py:55-70), attentive statistics
// (pooling.py:165-168), InstanceNorm1d CMVN (preprocessor.py:263,281), L2 normalisation
// (loss.py:91-100, xvector.py:903).  All are column-wise reductions over an utterance's own rows
// (SURVEY N2), HBM/L2-bound; threads map to the contiguous column axis so every wave reads whole
// 256-B lines, and the row loop runs over the utterance's true length only.
#include "kernels.h"

namespace sk {

__device__ inline float ld_elem(const void* x, int bf16, long idx) {
  return bf16 ? bf16_to_f32(reinterpret_cast<const uint16_t*>(x)[idx]) : reinterpret_cast<const float*>(x)[idx];
}

__global__ __launch_bounds__(256) void mean_std_kernel(const void* __restrict__ x, int x_bf16, long ld, int D, RowSpan rs,
                                                       float* __restrict__ out) {
  const int b = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
  if (d >= D) return;
  const long r0 = rs.row0(b);
  const int n = rs.count(b);
  float s = 0.f;
  for (int t = 0; t < n; ++t) s += ld_elem(x, x_bf16, (r0 + t) * ld + d);
  const float mean = s / (float)n;
  float v = 0.f;
  for (int t = 0; t < n; ++t) {
    const float e = ld_elem(x, x_bf16, (r0 + t) * ld + d) - mean;
    v = fmaf(e, e, v);
  }
  out[(long)b * 2 * D + d] = mean;
  out[(long)b * 2 * D + D + d] = sqrtf(v / (float)(n - 1));  // unbiased (torch.std default); n == 1 -> NaN as the reference
}

int launch_mean_std(const void* x, int x_bf16, long ld, int D, RowSpan rs, float* out, int B, hipStream_t s) {
  hipLaunchKernelGGL(mean_std_kernel, dim3(B, cdiv(D, 256)), dim3(256), 0, s, x, x_bf16, ld, D, rs, out);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

__global__ __launch_bounds__(256) void att_stats_kernel(const void* __restrict__ x, int x_bf16, const float* __restrict__ e,
                                                        long ld, int D, RowSpan rs, float* __restrict__ out) {
  const int b = blockIdx.x, d = blockIdx.y * 256 + threadIdx.x;
  if (d >= D) return;
  const long r0 = rs.row0(b);
  const int n = rs.count(b);
  float mx = -INFINITY;
  for (int t = 0; t < n; ++t) mx = fmaxf(mx, e[(r0 + t) * ld + d]);
  float z = 0.f, s1 = 0.f, s2 = 0.f;
  for (int t = 0; t < n; ++t) {
    const float w = expf(e[(r0 + t) * ld + d] - mx);
    const float xv = ld_elem(x, x_bf16, (r0 + t) * ld + d);
    z += w;
    s1 = fmaf(xv, w, s1);
    s2 = fmaf(xv * xv, w, s2);
  }
  const float mu = s1 / z;
  out[(long)b * 2 * D + d] = mu;
  out[(long)b * 2 * D + D + d] = sqrtf(fmaxf(s2 / z - mu * mu, 1e-9f));
}

int launch_att_stats(const void* x, int x_bf16, const float* e, long ld, int D, RowSpan rs, float* out, int B, hipStream_t s) {
  hipLaunchKernelGGL(att_stats_kernel, dim3(B, cdiv(D, 256)), dim3(256), 0, s, x, x_bf16, e, ld, D, rs, out);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

// CMVN: blockDim = (D, TG); threads with the same x reduce over interleaved rows through LDS.
__global__ void cmvn_kernel(float* __restrict__ x, long ld, int D, RowSpan rs, float eps) {
  extern __shared__ float red[];
  const int b = blockIdx.x, d = threadIdx.x, tg = threadIdx.y, TG = blockDim.y;
  const long r0 = rs.row0(b);
  const int n = rs.count(b);
  float s = 0.f;
  for (int t = tg; t < n; t += TG) s += x[(r0 + t) * ld + d];
  red[tg * D + d] = s;
  __syncthreads();
  float tot = 0.f;
  for (int q = 0; q < TG; ++q) tot += red[q * D + d];
  const float mean = tot / (float)n;
  __syncthreads();
  float v = 0.f;
  for (int t = tg; t < n; t += TG) {
    const float e = x[(r0 + t) * ld + d] - mean;
    v = fmaf(e, e, v);
  }
  red[tg * D + d] = v;
  __syncthreads();
  tot = 0.f;
  for (int q = 0; q < TG; ++q) tot += red[q * D + d];
  const float inv = 1.f / sqrtf(tot / (float)n + eps);
  for (int t = tg; t < n; t += TG) {
    const long i = (r0 + t) * ld + d;
    x[i] = (x[i] - mean) * inv;
  }
}

int launch_cmvn(float* x, long ld, int D, RowSpan rs, float eps, int B, hipStream_t s) {
  SK_CHECK(D <= 256, SK_EARG, "cmvn: D=%d > 256", D);
  const int TG = 1024 / D > 12 ? 12 : 1024 / D;
  hipLaunchKernelGGL(cmvn_kernel, dim3(B), dim3(D, TG), (size_t)D * TG * sizeof(float), s, x, ld, D, rs, eps);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, float* __restrict__ out, int D) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int d = tid; d < D; d += 256) { const float v = x[(long)b * D + d]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float n1 = sqrtf(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  // second pass: F.normalize(x / n1, eps=1e-12)
  float s2 = 0.f;
  for (int d = tid; d < D; d += 256) { const float v = x[(long)b * D + d] / n1; s2 = fmaf(v, v, s2); }
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) red[tid >> 6] = s2;
  __syncthreads();
  const float n2 = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
  for (int d = tid; d < D; d += 256) out[(long)b * D + d] = (x[(long)b * D + d] / n1) / n2;
}

int launch_l2norm(const float* x, float* out, int D, int B, hipStream_t s) {
  hipLaunchKernelGGL(l2norm_kernel, dim3(B), dim3(256), 0, s, x, out, D);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
