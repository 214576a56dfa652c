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

// Both pooling kernels: one workgroup per (utterance, 1024-column slab for bf16 / 512 for f32); a thread owns 16 B of
// consecutive columns (8 bf16 / 4 f32) and every TG-th row, so a row leaves HBM/L2 as whole 16-B-per-lane loads and the
// row loop is TG times shorter (one 2-B element per lane per row was latency-bound: 56 us for the 67 MB layer4 output);
// the row groups meet in LDS in a fixed order.
constexpr int POOL_TG = 4, POOL_CG = 64;   // 4 row groups x 64 column groups = 256 threads

template <int VEC>
__device__ inline void ld_vec(const void* x, long idx, float* v) {   // VEC = 8: bf16 elements, VEC = 4: f32 elements
  const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(x) + idx * (16 / VEC));
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
  if constexpr (VEC == 8) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = bf16_to_f32((uint16_t)(w[q] & 0xffff)); v[2 * q + 1] = bf16_to_f32((uint16_t)(w[q] >> 16)); }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = __builtin_bit_cast(float, w[q]);
  }
}

// sums the POOL_TG row-group partials of `vals` (VEC per thread) in group order and broadcasts the totals
template <int VEC>
__device__ inline void rowgroup_sum(float (*red)[POOL_CG * 8], int tg, int cg, float* vals) {
  __syncthreads();
#pragma unroll
  for (int q = 0; q < VEC; ++q) red[tg][cg * VEC + q] = vals[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < POOL_TG; ++g) t += red[g][cg * VEC + q];
    vals[q] = t;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void mean_std_kernel(const void* __restrict__ x, long ld, int D, RowSpan rs, float* __restrict__ out) {
  __shared__ float red[POOL_TG][POOL_CG * 8];
  const int b = blockIdx.x, cg = threadIdx.x % POOL_CG, tg = threadIdx.x / POOL_CG;
  const int d0 = (blockIdx.y * POOL_CG + cg) * VEC;
  const bool live = d0 < D;
  const long r0 = rs.row0(b);
  const int n = rs.count(b);
  float s[VEC], v[VEC], e[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) { s[q] = 0.f; v[q] = 0.f; }
  if (live)
    for (int t = tg; t < n; t += POOL_TG) {
      ld_vec<VEC>(x, (r0 + t) * ld + d0, e);
#pragma unroll
      for (int q = 0; q < VEC; ++q) s[q] += e[q];
    }
  rowgroup_sum<VEC>(red, tg, cg, s);
  if (live)
    for (int t = tg; t < n; t += POOL_TG) {
      ld_vec<VEC>(x, (r0 + t) * ld + d0, e);
#pragma unroll
      for (int q = 0; q < VEC; ++q) { const float c = e[q] - s[q] / (float)n; v[q] = fmaf(c, c, v[q]); }
    }
  rowgroup_sum<VEC>(red, tg, cg, v);
  if (live && tg == 0) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      out[(long)b * 2 * D + d0 + q] = s[q] / (float)n;
      out[(long)b * 2 * D + D + d0 + q] = sqrtf(v[q] / (float)(n - 1));  // unbiased (torch.std default); n == 1 -> NaN as the reference
    }
  }
}

int launch_mean_std(const void* x, int x_bf16, long ld, int D, RowSpan rs, float* out, int B, hipStream_t s) {
  const int VEC = x_bf16 ? 8 : 4;
  SK_CHECK(D % VEC == 0 && ld % VEC == 0, SK_EARG, "mean_std: D=%d, ld=%ld must be multiples of %d", D, ld, VEC);
  const dim3 grid(B, cdiv(D, POOL_CG * VEC));
  if (x_bf16) hipLaunchKernelGGL(mean_std_kernel<8>, grid, dim3(256), 0, s, x, ld, D, rs, out);
  else hipLaunchKernelGGL(mean_std_kernel<4>, grid, dim3(256), 0, s, x, ld, D, rs, out);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

template <int VEC>
__global__ __launch_bounds__(256) void att_stats_kernel(const void* __restrict__ x, const float* __restrict__ e, long ld, int D,
                                                        RowSpan rs, float* __restrict__ out) {
  __shared__ float red[POOL_TG][POOL_CG * 8];
  const int b = blockIdx.x, cg = threadIdx.x % POOL_CG, tg = threadIdx.x / POOL_CG;
  const int d0 = (blockIdx.y * POOL_CG + cg) * VEC;
  const bool live = d0 < D;
  const long r0 = rs.row0(b);
  const int n = rs.count(b);
  float mx[VEC], z[VEC], s1[VEC], s2[VEC], ev[VEC], xv[VEC];
#pragma unroll
  for (int q = 0; q < VEC; ++q) { mx[q] = -INFINITY; z[q] = 0.f; s1[q] = 0.f; s2[q] = 0.f; }
  auto ld_e = [&](int t) {
#pragma unroll
    for (int q = 0; q < VEC; q += 4) {
      const float4 f = *reinterpret_cast<const float4*>(e + (r0 + t) * ld + d0 + q);
      ev[q] = f.x; ev[q + 1] = f.y; ev[q + 2] = f.z; ev[q + 3] = f.w;
    }
  };
  if (live)
    for (int t = tg; t < n; t += POOL_TG) {
      ld_e(t);
#pragma unroll
      for (int q = 0; q < VEC; ++q) mx[q] = fmaxf(mx[q], ev[q]);
    }
  // max over the row groups
  __syncthreads();
#pragma unroll
  for (int q = 0; q < VEC; ++q) red[tg][cg * VEC + q] = mx[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < VEC; ++q) {
    float m = red[0][cg * VEC + q];
#pragma unroll
    for (int g = 1; g < POOL_TG; ++g) m = fmaxf(m, red[g][cg * VEC + q]);
    mx[q] = m;
  }
  if (live)
    for (int t = tg; t < n; t += POOL_TG) {
      ld_e(t);
      ld_vec<VEC>(x, (r0 + t) * ld + d0, xv);
#pragma unroll
      for (int q = 0; q < VEC; ++q) {
        const float w = expf(ev[q] - mx[q]);
        z[q] += w;
        s1[q] = fmaf(xv[q], w, s1[q]);
        s2[q] = fmaf(xv[q] * xv[q], w, s2[q]);
      }
    }
  rowgroup_sum<VEC>(red, tg, cg, z);
  rowgroup_sum<VEC>(red, tg, cg, s1);
  rowgroup_sum<VEC>(red, tg, cg, s2);
  if (live && tg == 0) {
#pragma unroll
    for (int q = 0; q < VEC; ++q) {
      const float mu = s1[q] / z[q];
      out[(long)b * 2 * D + d0 + q] = mu;
      out[(long)b * 2 * D + D + d0 + q] = sqrtf(fmaxf(s2[q] / z[q] - mu * mu, 1e-9f));
    }
  }
}

int launch_att_stats(const void* x, int x_bf16, const float* e, long ld, int D, RowSpan rs, float* out, int B, hipStream_t s) {
  const int VEC = x_bf16 ? 8 : 4;
  SK_CHECK(D % VEC == 0 && ld % VEC == 0, SK_EARG, "att_stats: D=%d, ld=%ld must be multiples of %d", D, ld, VEC);
  const dim3 grid(B, cdiv(D, POOL_CG * VEC));
  if (x_bf16) hipLaunchKernelGGL(att_stats_kernel<8>, grid, dim3(256), 0, s, x, e, ld, D, rs, out);
  else hipLaunchKernelGGL(att_stats_kernel<4>, grid, dim3(256), 0, s, x, e, ld, D, rs, out);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

// ---- attention.4 + softmax over time + weighted statistics in one kernel (bf16 path) ---------------------------------------
// pooling.py:161-168: e = conv1x1(h) (128 -> D) + bias; w = softmax_t(e); mu = sum_t x w; rh = sqrt(clamp(sum_t x^2 w - mu^2, 1e-9)).
// The separate form writes e (B*T' x D f32 = 134 MB at B = 256, 4 s) and reads it back twice.  Here the utterance's h rows stream
// through LDS in chunks of 64 and every 32 x 32 tile of e lives only in the MFMA accumulators (A = h rows so that a lane holds
// ONE column d and 16 time steps), first to find each column's maximum over time, then -- recomputed, K is only 128 -- for
// exp / sums.  Same operand rounding and k order as gemm_bf16_kernel, so e is bit-identical to the separate form; only the order
// of the time sums differs.
constexpr int AF_LD = 128 * 2 + 16;   // LDS row: 128 bf16 + 16 B pad (conflict-free ds_read_b128 fragments)

// One workgroup = (utterance, 128 columns); each of its four waves owns 32 columns for ALL time steps (two 32-row MFMA tiles per
// 64-row chunk), keeps its 32 x 128 weight slice in 32 registers for its whole life and never meets the other waves again: the
// only shared state is the chunk of h rows in LDS.
__global__ __launch_bounds__(256, 4) void att_fused_kernel(const uint16_t* __restrict__ x, const float* __restrict__ hmat, const uint16_t* __restrict__ w2,
                                                        const float* __restrict__ b2, long ld, int D, RowSpan rs, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned char Hs[64 * AF_LD];
  __shared__ __attribute__((aligned(16))) unsigned char Xs[64 * AF_LD];   // the chunk's x[.., 128 columns] (bf16), same row pitch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x, dcol = blockIdx.y * 128 + wave * 32 + r;   // this lane's column
  const long r0 = rs.row0(b);
  const int n = rs.count(b);
  uint4 wreg[8];                                    // B-operand fragments: W2[dcol][16 kk + 8 hh .. + 7]
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) wreg[kk] = *reinterpret_cast<const uint4*>(w2 + (long)dcol * 128 + kk * 16 + 8 * hh);
  const float bias = b2[dcol];
  // rows c0 .. c0 + 63 of the utterance: h (f32 -> bf16) and, when asked, the workgroup's 128 columns of x, both as 16-byte
  // coalesced requests issued together; rows past the utterance are zero.  (x read per lane straight from HBM is one 2-byte
  // request per element with a 64-bit address each: 168 registers, three workgroups per CU.)
  auto stage = [&](int c0, bool with_x) {
    __syncthreads();
    uint4 xv[4];
    if (with_x) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k, row = i >> 4, ch = i & 15;
        xv[k] = make_uint4(0, 0, 0, 0);
        if (c0 + row < n) xv[k] = *reinterpret_cast<const uint4*>(x + (r0 + c0 + row) * ld + blockIdx.y * 128 + ch * 8);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + 256 * k, row = i >> 4, ch = i & 15;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c0 + row < n) {
        const float* p = hmat + (r0 + c0 + row) * 128 + ch * 8;
        const float4 a = *reinterpret_cast<const float4*>(p), c = *reinterpret_cast<const float4*>(p + 4);
        v = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(c.x, c.y), pack_bf16x2(c.z, c.w));
      }
      *reinterpret_cast<uint4*>(Hs + row * AF_LD + ch * 16) = v;
    }
    if (with_x) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = tid + 256 * k, row = i >> 4, ch = i & 15;
        *reinterpret_cast<uint4*>(Xs + row * AF_LD + ch * 16) = xv[k];
      }
    }
    __syncthreads();
  };
  auto etile = [&](int mt) {                         // e[t = 32 mt + rowmap][d = dcol] of the staged chunk, bias added
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const uint4 a = *reinterpret_cast<const uint4*>(Hs + (mt * 32 + r) * AF_LD + (kk * 16 + 8 * hh) * 2);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, wreg[kk]), acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += bias;
    return acc;
  };
  auto lrow = [&](int mt, int q) { return mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * hh; };   // chunk row of accumulator element q
  const unsigned char* xcol = Xs + (wave * 32 + r) * 2;
  float z = 0.f, s1 = 0.f, s2 = 0.f;
  auto accumulate = [&](const f32x16& e, int c0, int mt, float mx) {   // weights and weighted sums of one e tile
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int lr = lrow(mt, q);
      const float w = c0 + lr < n ? expf(e[q] - mx) : 0.f;
      const float xv = bf16_to_f32(*reinterpret_cast<const uint16_t*>(xcol + lr * AF_LD));
      z += w;
      s1 = fmaf(xv, w, s1);
      s2 = fmaf(xv * xv, w, s2);
    }
  };
  if (n <= 64) {
    // the common case (T' = 51 for 4 s): one chunk, both e tiles computed once and kept: maximum, weights and sums straight from
    // the accumulators
    stage(0, true);
    const f32x16 e0 = etile(0), e1 = etile(1);
    float mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      mx = lrow(0, q) < n ? fmaxf(mx, e0[q]) : mx;
      mx = lrow(1, q) < n ? fmaxf(mx, e1[q]) : mx;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    accumulate(e0, 0, 0, mx);
    accumulate(e1, 0, 1, mx);
  } else {
    // longer utterances -- pass 1: column maxima over the utterance's own rows
    float mx = -INFINITY;
    for (int c0 = 0; c0 < n; c0 += 64) {
      stage(c0, false);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (c0 + mt * 32 >= n) break;                  // wave-uniform
        const f32x16 e = etile(mt);
#pragma unroll
        for (int q = 0; q < 16; ++q) mx = c0 + lrow(mt, q) < n ? fmaxf(mx, e[q]) : mx;
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // pass 2: weights and weighted sums (e recomputed: K is only 128)
    for (int c0 = 0; c0 < n; c0 += 64) {
      stage(c0, true);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        if (c0 + mt * 32 >= n) break;
        accumulate(etile(mt), c0, mt, mx);
      }
    }
  }
  z += __shfl_xor(z, 32);
  s1 += __shfl_xor(s1, 32);
  s2 += __shfl_xor(s2, 32);
  if (hh == 0) {
    const float mu = s1 / z;
    out[(long)b * 2 * D + dcol] = mu;
    out[(long)b * 2 * D + D + dcol] = sqrtf(fmaxf(s2 / z - mu * mu, 1e-9f));
  }
}

int launch_att_fused(const void* x_bf16, const float* h, const void* w2_bf16, const float* b2, long ld, int D, RowSpan rs, float* out, int B,
                     hipStream_t s) {
  SK_CHECK(D % 128 == 0 && ld >= D, SK_EARG, "att_fused: D=%d must be a multiple of 128", D);
  hipLaunchKernelGGL(att_fused_kernel, dim3(B, D / 128), dim3(256), 0, s, (const uint16_t*)x_bf16, h, (const uint16_t*)w2_bf16, b2, ld, D, rs, out);
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

// The same arithmetic with the thread's rows held in registers between the three passes (mean, variance, normalise): one read of the
// features instead of three dependent sweeps (4 s utterances: 401 frames = 34 rows per thread).  NR * TG >= the longest utterance.
template <int NR>
__global__ __launch_bounds__(1024) void cmvn_reg_kernel(float* __restrict__ x, long ld, int D, RowSpan rs, float eps) {
  extern __shared__ float red[];
  const int b = blockIdx.x, d = threadIdx.x, tg = threadIdx.y, TG = blockDim.y;
  const long r0 = rs.row0(b);
  const int n = rs.count(b);
  float v[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) { const int t = tg + i * TG; v[i] = t < n ? x[(r0 + t) * ld + d] : 0.f; }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) s += (tg + i * TG < n) ? v[i] : 0.f;
  red[tg * D + d] = s;
  __syncthreads();
  float tot = 0.f;
  for (int q = 0; q < TG; ++q) tot += red[q * D + d];
  const float mean = tot / (float)n;
  __syncthreads();
  float w = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const float e = v[i] - mean;
    w = (tg + i * TG < n) ? fmaf(e, e, w) : w;
  }
  red[tg * D + d] = w;
  __syncthreads();
  tot = 0.f;
  for (int q = 0; q < TG; ++q) tot += red[q * D + d];
  const float inv = 1.f / sqrtf(tot / (float)n + eps);
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int t = tg + i * TG;
    if (t < n) x[(r0 + t) * ld + d] = (v[i] - mean) * inv;
  }
}

int launch_cmvn(float* x, long ld, int D, RowSpan rs, float eps, int B, hipStream_t s) {
  SK_CHECK(D <= 256, SK_EARG, "cmvn: D=%d > 256", D);
  const int TG = 1024 / D > 12 ? 12 : 1024 / D;
  const size_t lds = (size_t)D * TG * sizeof(float);
  const int tmax = rs.stride;   // rows of the longest utterance of the batch (both the padded and the ragged row layout carry it)
  if (tmax > 0 && tmax <= 36 * TG) hipLaunchKernelGGL(cmvn_reg_kernel<36>, dim3(B), dim3(D, TG), lds, s, x, ld, D, rs, eps);
  else hipLaunchKernelGGL(cmvn_kernel, dim3(B), dim3(D, TG), lds, s, x, ld, D, rs, eps);
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

// out = x / max(||x||_2, eps): torch.nn.functional.normalize(x, dim=1) as the reference applies it to cohort / test x-vectors before
// cosine scoring (sidekit/score_normalization.py:128, sidekit/nnet/xvector.py:243,258-259)
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int D, float eps) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int d = tid; d < D; d += 256) { const float v = x[(long)b * D + d]; s = fmaf(v, v, s); }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float n = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), eps);
  for (int d = tid; d < D; d += 256) out[(long)b * D + d] = x[(long)b * D + d] / n;
}

int launch_normalize_rows(const float* x, float* out, int D, int B, float eps, hipStream_t s) {
  hipLaunchKernelGGL(normalize_rows_kernel, dim3(B), dim3(256), 0, s, x, out, D, eps);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

int launch_l2norm(const float* x, float* out, int D, int B, hipStream_t s) {
  hipLaunchKernelGGL(l2norm_kernel, dim3(B), dim3(256), 0, s, x, out, D);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
