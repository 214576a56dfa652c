// Trial-matrix scoring: cosine (sidekit/iv_scoring.py:98-109), fast PLDA (:448-462) and the
// per-trial cosine of sidekit/bin/compute_spk_cosine.py:18-26.
//
// cosine is one f32 MFMA GEMM over the already normalised rows.  PLDA keeps the reference's
// float64 arithmetic end to end: the 256x256 algebra (Phi, Psi, constant) stays on the host, the
// N^2 part runs here as an f64 MFMA GEMM (C = A . B^T, v_mfma_f64_16x16x4_f64) whose epilogue adds
// the two quadratic terms and the constant, so the (Ne x Nt) matrix is written exactly once.  For
// trial sets too large to materialise, sc_cosine_hist counts target / non-target scores into
// histograms straight from the accumulators.
#include "../../include/sidekit_amd.h"
#include "kernels.h"

namespace sk {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int DT = 64, DK = 16, DLD = DK + 1;   // 64 x 64 output tile per workgroup, 16-deep k-tiles, LDS rows padded by one double

// C[m][n] = alpha * (sum_k A[m][k] Bop(k,n) + rowterm[m] + colterm[n] + cst), float64 end to end on the matrix cores:
// v_mfma_f64_16x16x4_f64 (A: lane l holds A[l & 15][l >> 4], B: B[l >> 4][l & 15], D: four doubles per lane at
// row (l >> 4) + 4 * reg, col l & 15 -- the f64 map, which differs from the f32 one).  Four waves, each a 32 x 32 quadrant = 2 x 2
// MFMA tiles; operands staged through LDS as [row][k] with a 17-double row stride, which makes the ds_read_b64 fragment reads
// (16 rows x 2 k per half-wave) hit 32 different bank pairs.
//   B_KN = false: B is [N][K] (C = A . B^T);  B_KN = true: B is [K][N] (C = A . B)
template <bool B_KN>
__global__ __launch_bounds__(256) void dgemm_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                                       int M, int N, int K, const double* __restrict__ rowterm,
                                                       const double* __restrict__ colterm, double cst, double alpha) {
  __shared__ __attribute__((aligned(16))) double As[DT * DLD];
  __shared__ __attribute__((aligned(16))) double Bs[DT * DLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lk = lane >> 4;
  const int m0 = blockIdx.y * DT, n0 = blockIdx.x * DT;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += DK) {
    // stage: 64 rows x 16 k per operand, consecutive threads along the contiguous dimension of each source
    for (int i = tid; i < DT * DK; i += 256) {
      const int row = i / DK, kk = i % DK, m = m0 + row, k = k0 + kk;
      As[row * DLD + kk] = (m < M && k < K) ? A[(long)m * K + k] : 0.0;
      if constexpr (!B_KN) {
        const int n = n0 + row;
        Bs[row * DLD + kk] = (n < N && k < K) ? B[(long)n * K + k] : 0.0;
      }
    }
    if constexpr (B_KN) {
      for (int i = tid; i < DT * DK; i += 256) {
        const int kk = i / DT, col = i % DT, n = n0 + col, k = k0 + kk;
        Bs[col * DLD + kk] = (n < N && k < K) ? B[(long)k * N + n] : 0.0;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < DK; kk += 4) {
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = As[(wm * 32 + i * 16 + lr) * DLD + kk + lk];
        b[i] = Bs[(wn * 32 + i * 16 + lr) * DLD + kk + lk];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + wm * 32 + i * 16 + lk + 4 * q;
      if (m >= M) continue;
      const double rt = rowterm ? rowterm[m] : 0.0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + lr;
        if (n < N) C[(long)m * N + n] = alpha * (acc[i][j][q] + rt + (colterm ? colterm[n] : 0.0) + cst);
      }
    }
}

// ---- all-pairs cosine scoring WITHOUT the score matrix (100k x 100k trials are 40 GB of float32) ---------------------
// Persistent workgroups walk the 256 x 256 tiles of E . T^T (f32 MFMA, the arithmetic of sc_cosine), classify every score as
// target / non-target from the two label vectors and count it into a private LDS histogram pair; the histograms are added to
// the global 64-bit counters once, at the end.  EER / ROCCH then come from the counts (bosaris.detplot.eer_from_histograms).
constexpr int HB = 8192;   // bins per histogram: 2 x 32 KB of LDS per workgroup, one persistent workgroup per CU

// Tile: 256 x 256 per 1024-thread workgroup, sixteen waves of 64 x 64 (2 x 2 accumulator tiles each), k-tiles of 32 staged through
// LDS with the next k-tile's operands prefetched into registers.  The 64 KB of histograms allow only ONE workgroup per CU, so the
// latency hiding has to come from inside it: four waves per SIMD put 16 k matrix-pipe cycles between a prefetch and its use
// (64 x 64 tiles with load-then-compute kept the pipes 26 % busy, 128 x 128 with four waves 33 %).
constexpr int HT = 256, HLD = 36, HTHREADS = 1024;

__global__ __launch_bounds__(HTHREADS) void cosine_hist_kernel(const float* __restrict__ E, int Ne, const float* __restrict__ T, int Nt, int D,
                                                               const int* __restrict__ le, const int* __restrict__ lt, int self_offset,
                                                               float lo, float inv_width, unsigned long long* __restrict__ hist_tar,
                                                               unsigned long long* __restrict__ hist_non) {
  __shared__ unsigned hist[2 * HB];
  __shared__ __attribute__((aligned(16))) float Es[HT * HLD];
  __shared__ __attribute__((aligned(16))) float Ts[HT * HLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
  for (int i = tid; i < 2 * HB; i += HTHREADS) hist[i] = 0u;
  const long tiles_n = (Nt + HT - 1) / HT, ntiles = (long)((Ne + HT - 1) / HT) * tiles_n;
  const int srow = tid >> 3, sk4 = (tid & 7) * 4;   // 128 rows x 8 chunks per pass, two passes per operand
  const int nk = (D + 31) / 32;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = (int)(tile / tiles_n) * HT, n0 = (int)(tile % tiles_n) * HT;
    float4 re[2], rt[2];
    auto fetch = [&](int k0) {
      const int k = k0 + sk4;
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int row = srow + q * 128;
        re[q] = (m0 + row < Ne && k < D) ? *reinterpret_cast<const float4*>(E + (long)(m0 + row) * D + k) : z;
        rt[q] = (n0 + row < Nt && k < D) ? *reinterpret_cast<const float4*>(T + (long)(n0 + row) * D + k) : z;
      }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    fetch(0);
    for (int kt = 0; kt < nk; ++kt) {
      __syncthreads();            // every wave is done reading the previous k-tile
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<float4*>(&Es[(srow + q * 128) * HLD + sk4]) = re[q];
        *reinterpret_cast<float4*>(&Ts[(srow + q * 128) * HLD + sk4]) = rt[q];
      }
      __syncthreads();
      if (kt + 1 < nk) fetch((kt + 1) * 32);
#pragma unroll
      for (int kk = 0; kk < 32; kk += 8) {
        float4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const float4*>(&Es[(wm * 64 + i * 32 + r) * HLD + kk + 4 * h]);
          b[i] = *reinterpret_cast<const float4*>(&Ts[(wn * 64 + i * 32 + r) * HLD + kk + 4 * h]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + r;
      if (n >= Nt) continue;
      const int ln = lt[n];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          if (m >= Ne || (self_offset >= 0 && m + self_offset == n)) continue;
          int bin = (int)floorf((acc[i][j][q] - lo) * inv_width);
          bin = bin < 0 ? 0 : (bin >= HB ? HB - 1 : bin);
          atomicAdd(&hist[(le[m] == ln ? 0 : HB) + bin], 1u);
        }
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * HB; i += HTHREADS) {
    const unsigned c = hist[i];
    if (c) atomicAdd((i < HB ? hist_tar : hist_non) + (i & (HB - 1)), (unsigned long long)c);
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

int sc_cosine_hist(const float* d_E, int32_t Ne, const float* d_T, int32_t Nt, int32_t D, const int32_t* d_labels_e, const int32_t* d_labels_t,
                   int32_t self_offset, float lo, float hi, int32_t nbins, uint64_t* d_hist_tar, uint64_t* d_hist_non, void* stream) {
  SK_CHECK(d_E && d_T && d_labels_e && d_labels_t && d_hist_tar && d_hist_non && Ne > 0 && Nt > 0 && D > 0 && D % 4 == 0, SK_EARG,
           "sc_cosine_hist: bad arguments (D must be a multiple of 4)");
  SK_CHECK(nbins == HB && hi > lo, SK_EARG, "sc_cosine_hist: nbins must be %d and hi > lo", HB);
  hipStream_t st = (hipStream_t)stream;
  SK_HIP(hipMemsetAsync(d_hist_tar, 0, (size_t)HB * 8, st));
  SK_HIP(hipMemsetAsync(d_hist_non, 0, (size_t)HB * 8, st));
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long ntiles = (long)cdiv(Ne, HT) * cdiv(Nt, HT);
  const int grid = (int)(ntiles < (long)cus ? ntiles : (long)cus);   // persistent: one workgroup per CU (64 KB of histograms + 74 KB of operand tiles)
  hipLaunchKernelGGL(cosine_hist_kernel, dim3(grid), dim3(HTHREADS), 0, st, d_E, Ne, d_T, Nt, D, d_labels_e, d_labels_t, self_offset, lo,
                     (float)HB / (hi - lo), (unsigned long long*)d_hist_tar, (unsigned long long*)d_hist_non);
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
