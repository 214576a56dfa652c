// Trial-matrix scoring: cosine (sidekit/iv_scoring.py:98-109), fast PLDA (:448-462) and the
// per-trial cosine of sidekit/bin/compute_spk_cosine.py:18-26.
//
// cosine is one f32 MFMA GEMM over the already normalised rows.  PLDA keeps the reference's
// float64 arithmetic end to end: the 256x256 algebra (Phi, Psi, constant) stays on the host, the
// N^2 part runs here as a tiled f64 FMA GEMM (C = A . B^T) whose epilogue adds the two quadratic
// terms and the constant, so the (Ne x Nt) matrix is written exactly once.
#include "../../include/sidekit_amd.h"
#include "kernels.h"

namespace sk {

constexpr int DT = 64, DK = 16;

// C[m][n] = alpha * (sum_k A[m][k] Bop(k,n) + rowterm[m] + colterm[n] + cst)
//   B_KN = false: B is [N][K] (C = A . B^T);  B_KN = true: B is [K][N] (C = A . B)
template <bool B_KN>
__global__ __launch_bounds__(256) void dgemm_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                                       int M, int N, int K, const double* __restrict__ rowterm,
                                                       const double* __restrict__ colterm, double cst, double alpha) {
  __shared__ double As[DK][DT + 1];
  __shared__ double Bs[DK][DT + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * DT, n0 = blockIdx.x * DT;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int k0 = 0; k0 < K; k0 += DK) {
    for (int i = tid; i < DT * DK; i += 256) {
      const int row = i / DK, kk = i % DK;
      const int m = m0 + row, n = n0 + row, k = k0 + kk;
      As[kk][row] = (m < M && k < K) ? A[(long)m * K + k] : 0.0;
      Bs[kk][row] = (n < N && k < K) ? (B_KN ? B[(long)k * N + n] : B[(long)n * K + k]) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < DK; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty + 16 * i]; b[i] = Bs[kk][tx + 16 * i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 16 * i;
    if (m >= M) continue;
    const double rt = rowterm ? rowterm[m] : 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx + 16 * j;
      if (n < N) C[(long)m * N + n] = alpha * (acc[i][j] + rt + (colterm ? colterm[n] : 0.0) + cst);
    }
  }
}

// q[i] = 0.5 * sum_k X[i][k] * Y[i][k]
__global__ void half_rowdot_kernel(const double* __restrict__ X, const double* __restrict__ Y, double* __restrict__ q, int N, int D) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= N) return;
  double s = 0.0;
  for (int k = lane; k < D; k += 64) s = fma(X[(long)i * D + k], Y[(long)i * D + k], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) q[i] = 0.5 * s;
}

__global__ void cosine_trials_kernel(const float* __restrict__ E, const float* __restrict__ T, int D, const int* __restrict__ ei,
                                     const int* __restrict__ ti, long n, double* __restrict__ out) {
  const long k = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= n) return;
  const float* e = E + (long)ei[k] * D;
  const float* t = T + (long)ti[k] * D;
  double uv = 0, uu = 0, vv = 0;
  for (int d = lane; d < D; d += 64) {
    const double a = e[d], b = t[d];
    uv = fma(a, b, uv); uu = fma(a, a, uu); vv = fma(b, b, vv);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { uv += __shfl_xor(uv, o); uu += __shfl_xor(uu, o); vv += __shfl_xor(vv, o); }
  if (lane == 0) out[k] = uv / (sqrt(uu) * sqrt(vv));  // 1 - scipy.spatial.distance.cosine
}

// ---- adaptive s-norm support (sidekit/score_normalization.py:120-140) -----------------------------------------
// Mean and unbiased std of the k largest values of every row: an exact radix select on the order-preserving
// integer image of the floats (four 8-bit passes narrow the k-th largest key), then one pass of sums.  Ties at the
// threshold contribute exactly the copies torch.topk would keep, so the statistics equal those of any valid top-k.
__device__ inline unsigned fkey(float f) {
  const unsigned u = __builtin_bit_cast(unsigned, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // larger float <=> larger key
}

__global__ __launch_bounds__(256) void topk_stats_kernel(const float* __restrict__ x, int ncols, int k, float* __restrict__ mean,
                                                         float* __restrict__ stdv) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_remaining;
  __shared__ double red[2 * 256];
  const float* row = x + (size_t)blockIdx.x * ncols;
  const int tid = threadIdx.x;
  unsigned prefix = 0, mask = 0;
  unsigned remaining = (unsigned)k;   // how many of the still-undecided keys belong to the top-k
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < ncols; i += 256) {
      const unsigned key = fkey(row[i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned acc = 0;
      int b = 255;
      for (; b > 0; --b) {
        if (acc + hist[b] >= remaining) break;
        acc += hist[b];
      }
      s_prefix = prefix | ((unsigned)b << shift);
      s_remaining = remaining - acc;
    }
    __syncthreads();
    prefix = s_prefix;
    remaining = s_remaining;
    mask |= 255u << shift;
    __syncthreads();
  }
  // prefix == key of the k-th largest value; `remaining` copies of it are inside the top-k
  double s1 = 0.0, s2 = 0.0;
  float tval = 0.f;
  for (int i = tid; i < ncols; i += 256) {
    const float v = row[i];
    const unsigned key = fkey(v);
    if (key > prefix) { s1 += (double)v; s2 += (double)v * (double)v; }
    if (key == prefix) tval = v;
  }
  red[tid] = s1; red[256 + tid] = s2;
  __shared__ float s_tval;
  if (fkey(tval) == prefix) s_tval = tval;   // every writer holds the same value
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int q = 0; q < 256; ++q) { a += red[q]; b += red[256 + q]; }
    const double tv = (double)s_tval;
    a += tv * (double)remaining;
    b += tv * tv * (double)remaining;
    const double m = a / (double)k;
    mean[blockIdx.x] = (float)m;
    const double var = (b - (double)k * m * m) / (double)(k - 1);
    stdv[blockIdx.x] = (float)sqrt(var > 0.0 ? var : 0.0);
  }
}

// S[i][j] <- 0.5 * ((S[i][j] - me[i]) / se[i] + (S[i][j] - mt[j]) / st[j])
__global__ void snorm_apply_kernel(float* __restrict__ S, int ne, int nt, const float* __restrict__ me, const float* __restrict__ se,
                                   const float* __restrict__ mt, const float* __restrict__ st) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= (long)ne * nt) return;
  const int r = (int)(i / nt), c = (int)(i % nt);
  const float v = S[i];
  S[i] = 0.5f * ((v - me[r]) / se[r]) + 0.5f * ((v - mt[c]) / st[c]);
}

static int dgemm(bool b_kn, const double* A, const double* B, double* C, int M, int N, int K, const double* rt, const double* ct,
                 double cst, double alpha, hipStream_t s) {
  const dim3 grid(cdiv(N, DT), cdiv(M, DT));
  if (b_kn) hipLaunchKernelGGL(dgemm_kernel<true>, grid, dim3(256), 0, s, A, B, C, M, N, K, rt, ct, cst, alpha);
  else hipLaunchKernelGGL(dgemm_kernel<false>, grid, dim3(256), 0, s, A, B, C, M, N, K, rt, ct, cst, alpha);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk

using namespace sk;

extern "C" {

int sc_cosine(const float* d_E, int32_t Ne, const float* d_T, int32_t Nt, int32_t D, float* d_out, void* stream) {
  SK_CHECK(d_E && d_T && d_out && Ne > 0 && Nt > 0 && D > 0 && D % 4 == 0, SK_EARG, "sc_cosine: bad arguments (D must be a multiple of 4)");
  GemmArgs g = gemm_args();
  g.A = d_E; g.lda = D; g.a_rows = Ne; g.W = d_T; g.ldw = D; g.C = d_out; g.ldc = Nt; g.M = Ne; g.N = Nt; g.K = D;
  return launch_gemm(g, (hipStream_t)stream);
}

int sc_plda_fast(const double* d_E, int32_t Ne, const double* d_T, int32_t Nt, int32_t D, const double* d_Phi, const double* d_Psi,
                 double cst, double scaling, double* d_out, void* stream) {
  SK_CHECK(d_E && d_T && d_Phi && d_Psi && d_out && Ne > 0 && Nt > 0 && D > 0, SK_EARG, "sc_plda_fast: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  double *tmp = nullptr, *qe = nullptr, *qt = nullptr;
  const int Nmax = Ne > Nt ? Ne : Nt;
  SK_HIP(hipMallocAsync((void**)&tmp, (size_t)Nmax * D * 8, st));
  SK_HIP(hipMallocAsync((void**)&qe, (size_t)Ne * 8, st));
  SK_HIP(hipMallocAsync((void**)&qt, (size_t)Nt * 8, st));
  int rc = SK_OK;
  do {
    // model_part / seg_part = 0.5 * diag(X Phi X')   (iv_scoring.py:449-450)
    if ((rc = dgemm(true, d_E, d_Phi, tmp, Ne, D, D, nullptr, nullptr, 0.0, 1.0, st))) break;
    hipLaunchKernelGGL(half_rowdot_kernel, dim3(cdiv(Ne, 4)), dim3(256), 0, st, tmp, d_E, qe, Ne, D);
    if ((rc = dgemm(true, d_T, d_Phi, tmp, Nt, D, D, nullptr, nullptr, 0.0, 1.0, st))) break;
    hipLaunchKernelGGL(half_rowdot_kernel, dim3(cdiv(Nt, 4)), dim3(256), 0, st, tmp, d_T, qt, Nt, D);
    // scoremat = (model_part[:, None] + seg_part + cst + E Psi T') * scaling   (:458-460)
    if ((rc = dgemm(true, d_E, d_Psi, tmp, Ne, D, D, nullptr, nullptr, 0.0, 1.0, st))) break;
    rc = dgemm(false, tmp, d_T, d_out, Ne, Nt, D, qe, qt, cst, scaling, st);
  } while (0);
  (void)hipFreeAsync(tmp, st); (void)hipFreeAsync(qe, st); (void)hipFreeAsync(qt, st);
  return rc;
}

int sc_topk_stats(const float* d_scores, int32_t n_rows, int32_t n_cols, int32_t k, float* d_mean, float* d_std, void* stream) {
  SK_CHECK(d_scores && d_mean && d_std && n_rows > 0 && k > 1 && k <= n_cols, SK_EARG, "sc_topk_stats: need 1 < k <= n_cols (k=%d, n_cols=%d)", k, n_cols);
  hipLaunchKernelGGL(topk_stats_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, d_scores, n_cols, k, d_mean, d_std);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

int sc_snorm_apply(float* d_S, int32_t Ne, int32_t Nt, const float* d_mean_e, const float* d_std_e, const float* d_mean_t,
                   const float* d_std_t, void* stream) {
  SK_CHECK(d_S && d_mean_e && d_std_e && d_mean_t && d_std_t && Ne > 0 && Nt > 0, SK_EARG, "sc_snorm_apply: bad arguments");
  hipLaunchKernelGGL(snorm_apply_kernel, dim3((unsigned)(((long)Ne * Nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_S, Ne, Nt,
                     d_mean_e, d_std_e, d_mean_t, d_std_t);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

int sc_cosine_trials(const float* d_E, const float* d_T, int32_t D, const int32_t* d_enr_idx, const int32_t* d_tst_idx, int64_t n_trials,
                     double* d_out, void* stream) {
  SK_CHECK(d_E && d_T && d_enr_idx && d_tst_idx && d_out && D > 0 && n_trials > 0, SK_EARG, "sc_cosine_trials: bad arguments");
  hipLaunchKernelGGL(cosine_trials_kernel, dim3((unsigned)((n_trials + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d_E, d_T, D,
                     d_enr_idx, d_tst_idx, (long)n_trials, d_out);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // extern "C"
