// Trial-matrix scoring: cosine (sidekit/iv_scoring.py:98-109), fast PLDA (:448-462) and the
// per-trial cosine of sidekit/bin/compute_spk_cosine.py:18-26.
//
// cosine is one f32 MFMA GEMM over the already normalised rows.  PLDA keeps the reference's
// float64 arithmetic end to end: the 256x256 algebra (Phi, Psi, constant) stays on the host; here ONE
// launch forms E.Psi and the two quadratic forms, and the N^2 part runs as an f64 MFMA GEMM
// (C = A . B^T, v_mfma_f64_16x16x4_f64) whose epilogue adds the quadratic terms and the constant, so
// the (Ne x Nt) matrix is written exactly once.  For
// trial sets too large to materialise, sc_cosine_hist counts target / non-target scores into
// histograms straight from the accumulators.
#include <map>
#include <mutex>
#include <utility>

#include "../../include/sidekit_amd.h"
#include "kernels.h"

namespace sk {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// f64 GEMM tile on the matrix cores, the core of fast / full PLDA scoring.
//   v_mfma_f64_16x16x4_f64: A: lane l holds A[l & 15][l >> 4], B: B[l >> 4][l & 15], D: four doubles per lane at row (l >> 4) + 4 * reg,
//   col l & 15 (the f64 map, which differs from the f32 one); 64 matrix-pipe cycles each (78.6 TFLOP/s = 32 FLOP / clk / SIMD).
// Four waves in a 2 x 2 grid, each WT x WT MFMA tiles: WT = 2 -> 64 x 64 per workgroup (small trial sets: enough workgroups to fill 256
// CUs), WT = 4 -> 128 x 128 (half the operand traffic per FLOP, 64 accumulator doubles per lane, two workgroups per CU).  Operands are
// staged through LDS as [row][k] with a 17-double row stride (ds_read_b64 fragment reads: 16 rows x 2 k per half-wave land on 32
// different bank pairs but one), the next k-tile's operands are fetched into registers (16-byte loads) while this one is multiplied.
//   B_KN = false: B is [N][K] (C = A . B^T);  B_KN = true: B is [K][N] (C = A . B)
constexpr int DK = 16, DLD = DK + 1;

template <int WT, bool B_KN>
__device__ inline void dgemm_tile(const double* __restrict__ A, const double* __restrict__ B, int M, int N, int K, int m0, int n0,
                                  double* As, double* Bs, f64x4 (&acc)[WT][WT]) {
  constexpr int T = 32 * WT;            // workgroup tile edge
  constexpr int PA = T * DK / 2 / 256;  // double pairs per thread and operand
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lk = lane >> 4;
  const bool vec_a = (K & 1) == 0 && (reinterpret_cast<size_t>(A) & 15) == 0;
  const bool vec_b = ((B_KN ? N : K) & 1) == 0 && (reinterpret_cast<size_t>(B) & 15) == 0;
  double2 ra[PA], rb[PA];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int idx = tid + 256 * q;
      {  // A: T rows x 8 pairs along k
        const int row = idx >> 3, kp = (idx & 7) * 2, m = m0 + row, k = k0 + kp;
        double2 v = {0.0, 0.0};
        if (m < M) {
          const double* src = A + (long)m * K + k;
          if (vec_a && k + 1 < K) v = *reinterpret_cast<const double2*>(src);
          else { if (k < K) v.x = src[0]; if (k + 1 < K) v.y = src[1]; }
        }
        ra[q] = v;
      }
      if constexpr (!B_KN) {
        const int row = idx >> 3, kp = (idx & 7) * 2, n = n0 + row, k = k0 + kp;
        double2 v = {0.0, 0.0};
        if (n < N) {
          const double* src = B + (long)n * K + k;
          if (vec_b && k + 1 < K) v = *reinterpret_cast<const double2*>(src);
          else { if (k < K) v.x = src[0]; if (k + 1 < K) v.y = src[1]; }
        }
        rb[q] = v;
      } else {  // B [K][N]: 16 k x T/2 pairs along n
        const int kk = idx / (T / 2), np = (idx % (T / 2)) * 2, n = n0 + np, k = k0 + kk;
        double2 v = {0.0, 0.0};
        if (k < K) {
          const double* src = B + (long)k * N + n;
          if (vec_b && n + 1 < N) v = *reinterpret_cast<const double2*>(src);
          else { if (n < N) v.x = src[0]; if (n + 1 < N) v.y = src[1]; }
        }
        rb[q] = v;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx >> 3, kp = (idx & 7) * 2;
      As[row * DLD + kp] = ra[q].x; As[row * DLD + kp + 1] = ra[q].y;
      if constexpr (!B_KN) { Bs[row * DLD + kp] = rb[q].x; Bs[row * DLD + kp + 1] = rb[q].y; }
      else {
        const int kk = idx / (T / 2), np = (idx % (T / 2)) * 2;
        Bs[np * DLD + kk] = rb[q].x; Bs[(np + 1) * DLD + kk] = rb[q].y;
      }
    }
  };
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += DK) {
    __syncthreads();   // every wave is done reading the previous k-tile
    stage();
    __syncthreads();
    if (k0 + DK < K) fetch(k0 + DK);
#pragma unroll
    for (int kk = 0; kk < DK; kk += 4) {
      double a[WT], b[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) {
        a[i] = As[(wm * 16 * WT + i * 16 + lr) * DLD + kk + lk];
        b[i] = Bs[(wn * 16 * WT + i * 16 + lr) * DLD + kk + lk];
      }
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
}

// C[m][n] = alpha * (sum_k A[m][k] B[n][k] + rowterm(m) + colterm(n) + cst); the two terms arrive as `nparts` partial sums each
// (rowterm(m) = sum_p rowpart[p * M + m], fixed order), which is how plda_prep_kernel leaves the quadratic forms.
template <int WT>
__global__ __launch_bounds__(256, WT == 4 ? 2 : 4) void dgemm_nt_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C,
                                                          int M, int N, int K, const double* __restrict__ rowpart,
                                                          const double* __restrict__ colpart, int nparts, double cst, double alpha) {
  constexpr int T = 32 * WT;
  __shared__ __attribute__((aligned(16))) double As[T * DLD];
  __shared__ __attribute__((aligned(16))) double Bs[T * DLD];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lk = lane >> 4;
  const int m0 = blockIdx.y * T, n0 = blockIdx.x * T;
  f64x4 acc[WT][WT];
  dgemm_tile<WT, false>(A, B, M, N, K, m0, n0, As, Bs, acc);
  double ct[WT];
#pragma unroll
  for (int j = 0; j < WT; ++j) {
    const int n = n0 + wn * 16 * WT + j * 16 + lr;
    double t = 0.0;
    if (n < N) for (int p = 0; p < nparts; ++p) t += colpart[(long)p * N + n];
    ct[j] = t;
  }
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + wm * 16 * WT + i * 16 + lk + 4 * q;
      if (m >= M) continue;
      double rt = 0.0;
      for (int p = 0; p < nparts; ++p) rt += rowpart[(long)p * M + m];
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const int n = n0 + wn * 16 * WT + j * 16 + lr;
        if (n < N) C[(long)m * N + n] = alpha * (acc[i][j][q] + rt + ct[j] + cst);
      }
    }
}

// Everything fast PLDA needs before its N^2 product, in ONE launch (sidekit/iv_scoring.py:449-458):
//   qpart_e[p][i] = 0.5 * sum_{n in column tile p} (E Phi)[i][n] E[i][n]      -> model_part = sum_p qpart_e[p]   (:449)
//   qpart_t[p][j] = the same for T                                          -> seg_part                        (:450)
//   EPsi = E . Psi                                                           (left factor of :458)
// Grid: x = column tile of [Phi | Psi] (D/64 + D/64), y = row tile of [E ; T]; the (T rows, Psi columns) workgroups have nothing to
// do.  The products X Phi are never written: each workgroup folds its 64 x 64 tile with the matching slice of X straight from the
// accumulators (16-lane reduction, then the two column waves through LDS; deterministic).
__global__ __launch_bounds__(256) void plda_prep_kernel(const double* __restrict__ E, int Ne, const double* __restrict__ Tm, int Nt, int D,
                                                        const double* __restrict__ Phi, const double* __restrict__ Psi,
                                                        double* __restrict__ qpart_e, double* __restrict__ qpart_t, double* __restrict__ EPsi) {
  constexpr int WT = 2, T = 64;
  __shared__ __attribute__((aligned(16))) double As[T * DLD];
  __shared__ __attribute__((aligned(16))) double Bs[T * DLD];
  __shared__ double red[2][T];
  const int ctiles = (D + T - 1) / T, etiles = (Ne + T - 1) / T;
  const bool psi = (int)blockIdx.x >= ctiles, is_t = (int)blockIdx.y >= etiles;
  if (psi && is_t) return;
  const double* X = is_t ? Tm : E;
  const int M = is_t ? Nt : Ne;
  const int m0 = (is_t ? (int)blockIdx.y - etiles : (int)blockIdx.y) * T, p = psi ? (int)blockIdx.x - ctiles : (int)blockIdx.x, n0 = p * T;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lk = lane >> 4;
  f64x4 acc[WT][WT];
  dgemm_tile<WT, true>(X, psi ? Psi : Phi, M, D, D, m0, n0, As, Bs, acc);
  if (psi) {
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = m0 + wm * 32 + i * 16 + lk + 4 * q;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < WT; ++j) {
          const int n = n0 + wn * 32 + j * 16 + lr;
          if (n < D) EPsi[(long)m * D + n] = acc[i][j][q];
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = wm * 32 + i * 16 + lk + 4 * q, m = m0 + r;
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const int n = n0 + wn * 32 + j * 16 + lr;
        if (m < M && n < D) s = fma(acc[i][j][q], X[(long)m * D + n], s);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);   // the 16 lanes that share lk hold this row's 16 columns of each tile
      if (lr == 0) red[wn][r] = s;
    }
  __syncthreads();
  if (threadIdx.x < T && m0 + (int)threadIdx.x < M)
    (is_t ? qpart_t : qpart_e)[(long)p * M + m0 + threadIdx.x] = 0.5 * (red[0][threadIdx.x] + red[1][threadIdx.x]);
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
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5, wm = wave >> 2, wn = wave & 3;
  for (int i = tid; i < 2 * HB; i += HTHREADS) hist[i] = 0u;
  const long tiles_n = (Nt + HT - 1) / HT, ntiles = (long)((Ne + HT - 1) / HT) * tiles_n;
  const int srow = tid >> 3, sk4 = (tid & 7) * 4;   // 128 rows x 8 chunks per pass, two passes per operand
  const int nk = (D + 31) / 32;
  auto flush = [&]() {   // LDS counts -> the global 64-bit counters
    for (int i = tid; i < 2 * HB; i += HTHREADS) {
      const unsigned c = hist[i];
      if (c) atomicAdd((i < HB ? hist_tar : hist_non) + (i & (HB - 1)), (unsigned long long)c);
      hist[i] = 0u;
    }
  };
  int since_flush = 0;
  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // a tile adds at most 65 536 counts to one 32-bit LDS bin: flush before 2^15 tiles could wrap it (a few million x a few
    // million trials per workgroup get there in the central non-target bins)
    if (++since_flush == (1 << 15)) { __syncthreads(); flush(); since_flush = 0; __syncthreads(); }
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
  flush();
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

// ---- workspace of sc_plda_fast: E . Psi (Ne x D) and the partial quadratic forms, cached per (device, stream) so that a call
// allocates nothing in the steady state (three hipMallocAsync / hipFreeAsync pairs per call were most of a 1000 x 1000 scoring).  A
// stream's calls are ordered, so reuse needs no further synchronisation; growth waits for that stream's earlier calls first.  The
// caller holds g_plda_mu from the lookup until its two kernels are enqueued (with the lock dropped in between, a second host
// thread growing the same stream's entry could free the buffer a first thread was about to launch on).  Entries live until
// sc_release_workspace() (called by the Python shim at interpreter exit and by tests); a stream handle the runtime recycles finds the
// old entry, which is harmless for the same reason reuse is: work on one stream handle is ordered.
struct PldaWs { void* p = nullptr; size_t bytes = 0; };
static std::mutex g_plda_mu;
static std::map<std::pair<int, hipStream_t>, PldaWs> g_plda_ws;

static int plda_workspace_locked(hipStream_t st, size_t bytes, void** out) {
  int dev = 0;
  SK_HIP(hipGetDevice(&dev));
  PldaWs& w = g_plda_ws[{dev, st}];
  if (bytes > w.bytes) {
    if (w.p) { SK_HIP(hipStreamSynchronize(st)); SK_HIP(hipFree(w.p)); w.p = nullptr; w.bytes = 0; }
    const size_t want = bytes + bytes / 4;
    SK_HIP(hipMalloc(&w.p, want));
    w.bytes = want;
  }
  *out = w.p;
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
  const int ctiles = cdiv(D, 64);
  const size_t n_epsi = ((size_t)Ne * D + 1) & ~(size_t)1, n_qe = (size_t)ctiles * Ne, n_qt = (size_t)ctiles * Nt;
  void* ws = nullptr;
  std::lock_guard<std::mutex> lock(g_plda_mu);   // held until both launches are enqueued (see plda_workspace_locked)
  SK_TRY(plda_workspace_locked(st, (n_epsi + n_qe + n_qt) * 8, &ws));
  double* epsi = (double*)ws;
  double* qe = epsi + n_epsi;
  double* qt = qe + n_qe;
  // 1) model_part / seg_part = 0.5 * diag(X Phi X') as per-column-tile partials, and E . Psi   (iv_scoring.py:449-450,458)
  hipLaunchKernelGGL(plda_prep_kernel, dim3(2 * ctiles, cdiv(Ne, 64) + cdiv(Nt, 64)), dim3(256), 0, st, d_E, Ne, d_T, Nt, D, d_Phi, d_Psi, qe, qt, epsi);
  SK_HIP(hipGetLastError());
  // 2) scoremat = (model_part[:, None] + seg_part + cst + (E Psi) T') * scaling   (:458-460)
  const bool big = (long)cdiv(Ne, 128) * cdiv(Nt, 128) >= 512;   // two 128 x 128 workgroups per CU and still two rounds of them
  if (big) hipLaunchKernelGGL(dgemm_nt_kernel<4>, dim3(cdiv(Nt, 128), cdiv(Ne, 128)), dim3(256), 0, st, epsi, d_T, d_out, Ne, Nt, D, qe, qt, ctiles, cst, scaling);
  else hipLaunchKernelGGL(dgemm_nt_kernel<2>, dim3(cdiv(Nt, 64), cdiv(Ne, 64)), dim3(256), 0, st, epsi, d_T, d_out, Ne, Nt, D, qe, qt, ctiles, cst, scaling);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

int sc_normalize_rows(const float* d_X, int32_t N, int32_t D, float* d_out, void* stream) {
  SK_CHECK(d_X && d_out && N > 0 && D > 0, SK_EARG, "sc_normalize_rows: bad arguments");
  return launch_normalize_rows(d_X, d_out, D, N, 1e-12f, (hipStream_t)stream);
}

int sc_release_workspace(void) {
  std::lock_guard<std::mutex> lock(g_plda_mu);
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto& kv : g_plda_ws) {
    if (!kv.second.p) continue;
    if (hipSetDevice(kv.first.first) != hipSuccess) continue;
    (void)hipDeviceSynchronize();
    (void)hipFree(kv.second.p);
  }
  g_plda_ws.clear();
  (void)hipSetDevice(cur);
  return SK_OK;
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
  int dev = 0, cus = 256;   // the CU count of the device the stream belongs to (the NULL stream: the current device)
  hipDevice_t sdev;
  if (st && hipStreamGetDevice(st, &sdev) == hipSuccess) dev = (int)sdev;
  else (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (cus <= 0) cus = 256;
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
