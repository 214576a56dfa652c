// Generic fp32 GEMM on the matrix cores (v_mfma_f32_32x32x2_f32: bit-exact f32 FMA chain, no
// reduced-precision path on gfx950) with pluggable A-operand loaders and a fused epilogue.
// Serves every dense contraction of the path that is NOT a 3x3 trunk convolution:
//   STFT-as-DFT and mel / DCT projections (torchaudio MelSpectrogram / MFCC semantics),
//   attentive-pooling 1x1 convs (pooling.py:136-141), embedding linears (xvector.py:489-491,
//   578-581), AAM cosine logits (loss.py:307-310), TDNN dilated conv1d stack (xvector.py:467-483),
//   cosine trial scoring (iv_scoring.py:108-109).
// Tile: 64x64x32 per 256-thread workgroup, four waves each owning a 32x32 accumulator;
// global->register prefetch of tile k+1 overlaps the MFMAs of tile k; LDS rows padded to 36
// floats so the ds_read_b128 fragment reads are bank-conflict free.
#include "kernels.h"

namespace sk {

GemmArgs gemm_args() {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.alpha = 1.f;
  return g;
}

constexpr int BM = 64, BN = 64, BK = 32, LDT = BK + 4;

struct LoadPlain {
  __device__ static inline float4 load(const GemmArgs& g, int m, int k) {
    long row = m;
    int c = k;
    if (g.kc) { row += (long)(k / g.kc) * g.dil; c = k % g.kc; }
    if (m >= g.M || k >= g.K || row >= g.a_rows) return make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.a_bf16) {
      const uint2 v = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(g.A) + row * g.lda + c);
      return make_float4(bf16_to_f32(v.x & 0xffff), bf16_to_f32(v.x >> 16), bf16_to_f32(v.y & 0xffff), bf16_to_f32(v.y >> 16));
    }
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(g.A) + row * g.lda + c);
  }
};

struct LoadFrames {
  // augmentation.py:70-74 pre-emphasis (reflect pad 1 on the left) folded into the STFT framing
  // (torch.stft center=True, pad_mode='reflect'; hann window centred in the n_fft frame).
  __device__ static inline float sample(const float* w, int L, int i, float coef) {
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    const int p = (i == 0) ? 1 : i - 1;
    return w[i] - coef * w[p];
  }
  __device__ static inline float4 load(const GemmArgs& g, int m, int k) {
    if (m >= g.M || k >= g.K) return make_float4(0.f, 0.f, 0.f, 0.f);
    int b, t;
    if (g.row_b) { b = g.row_b[m]; t = g.row_t[m]; } else { b = m / g.t_max; t = m % g.t_max; }
    const int L = g.nsamples ? g.nsamples[b] : g.nsamples_uniform;
    if (t > L / g.hop) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float* w = reinterpret_cast<const float*>(g.A) + (long)b * g.wav_ld;
    const int i0 = t * g.hop - g.K / 2 + k;
    const float4 win = *reinterpret_cast<const float4*>(g.window + k);
    return make_float4(win.x * sample(w, L, i0, g.preemph), win.y * sample(w, L, i0 + 1, g.preemph),
                       win.z * sample(w, L, i0 + 2, g.preemph), win.w * sample(w, L, i0 + 3, g.preemph));
  }
};

struct LoadPower {
  __device__ static inline float4 load(const GemmArgs& g, int m, int k) {
    if (m >= g.M || k >= g.K) return make_float4(0.f, 0.f, 0.f, 0.f);
    const float* p = reinterpret_cast<const float*>(g.A) + (long)m * g.lda + k;
    const float4 re = *reinterpret_cast<const float4*>(p);
    const float4 im = *reinterpret_cast<const float4*>(p + g.kc);
    return make_float4(re.x * re.x + im.x * im.x, re.y * re.y + im.y * im.y, re.z * re.z + im.z * im.z,
                       re.w * re.w + im.w * im.w);
  }
};

__device__ inline float gemm_epilogue(const GemmArgs& g, float v, int m, int n) {
  if (g.bias) v += g.bias[n];
  if (g.rowbias) v += g.rowbias[(long)(m / g.rows_per_group) * g.N + n];
  const float sc = g.scale ? g.scale[n] : 1.f, sh = g.scale ? g.shift[n] : 0.f;
  switch (g.act) {
    case ACT_RELU: v = relu_nan(v); break;
    case ACT_LRELU02: v = v > 0.f ? v : 0.2f * v; break;
    case ACT_RELU_BN_TANH: v = tanhf(relu_nan(v) * sc + sh); break;
    case ACT_LOG_EPS: v = logf(v + 1e-6f); break;
    default: break;
  }
  if (g.act != ACT_RELU_BN_TANH) v = v * sc + sh;
  return v * g.alpha;
}

template <class LA>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[BM * LDT];
  __shared__ __attribute__((aligned(16))) float Ws[BN * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  // staging assignment: 2 float4 of A and 2 of W per thread per k-tile
  const int srow = tid >> 3, sk4 = (tid & 7) * 4;
  float4 ra[2], rw[2];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = srow + q * 32;
      ra[q] = LA::load(g, m0 + row, k0 + sk4);
      const int n = n0 + row, k = k0 + sk4;
      rw[q] = (n < g.N && k < g.K) ? *reinterpret_cast<const float4*>(g.W + (long)n * g.ldw + k)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // K is cut into g.kslices slices of whole k-tiles and a result is ALWAYS 0 + slice 0 + slice 1 + ... (each slice an FMA chain
  // from zero), whether the slices run as blockIdx.z (small M: parallelism; gemm_splitk_epilogue_kernel adds them) or one after the
  // other in this workgroup (large M) -- an utterance's embedding does not depend on how many others share its batch.
  f32x16 acc, tot;
#pragma unroll
  for (int q = 0; q < 16; ++q) tot[q] = 0.f;
  const int nk_all = (g.K + BK - 1) / BK;
  const int per = (nk_all + g.kslices - 1) / g.kslices;
  const int z0 = g.ksplit > 1 ? (int)blockIdx.z : 0, z1 = g.ksplit > 1 ? z0 + 1 : g.kslices;
  for (int z = z0; z < z1; ++z) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    const int kt0 = z * per, kt1 = (kt0 + per < nk_all) ? kt0 + per : nk_all;
    if (kt0 < kt1) fetch(kt0 * BK);
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<float4*>(&As[(srow + q * 32) * LDT + sk4]) = ra[q];
        *reinterpret_cast<float4*>(&Ws[(srow + q * 32) * LDT + sk4]) = rw[q];
      }
      __syncthreads();
      if (kt + 1 < kt1) fetch((kt + 1) * BK);
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        const float4 a = *reinterpret_cast<const float4*>(&As[(wm * 32 + r) * LDT + kk + 4 * h]);
        const float4 b = *reinterpret_cast<const float4*>(&Ws[(wn * 32 + r) * LDT + kk + 4 * h]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
      }
      __syncthreads();
    }
    if (g.ksplit == 1 && g.kslices > 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) tot[q] += acc[q];
    }
  }
  if (g.ksplit == 1 && g.kslices > 1) acc = tot;
  // epilogue: D[row = (q&3) + 8*(q>>2) + 4*h][col = r]
  const int n = n0 + wn * 32 + r;
  if (n >= g.N) return;
  if (g.ksplit > 1) {  // raw partial sums; gemm_splitk_epilogue_kernel adds them in slice order and applies the epilogue
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (m < g.M) g.splitk_ws[((long)blockIdx.z * g.M + m) * g.N + n] = acc[q];
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int m = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (m < g.M) g.C[(long)m * g.ldc + n] = gemm_epilogue(g, acc[q], m, n);
  }
}

// ---- 128 x 128 tile for the large problems (TDNN layers: 96k rows x 512..1536; 16k x 16k cosine trial matrices) -------------
// With 64 x 64 tiles every k-tile moves 16 KB of operands for 262 kFLOP: 16 FLOP/B, i.e. 6 TB/s of L2 traffic at the 96 TFLOP/s
// the kernel reached (61 % of the f32 MFMA peak).  Four waves of 64 x 64 (2 x 2 accumulator tiles, 64 registers) halve that.
// Each output element sees the same k-ordered FMA chain as in the 64 x 64 kernel: results are bit-identical.
constexpr int BM2 = 128, BN2 = 128;

template <bool SLICED>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[BM2 * LDT];
  __shared__ __attribute__((aligned(16))) float Ws[BN2 * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * BM2, n0 = blockIdx.y * BN2;
  const int srow = tid >> 3, sk4 = (tid & 7) * 4;
  // operand addresses advance with the k-tile (plain f32 A rows only; a dilated-conv A walks (tap, channel) without a division:
  // k = tap * kc + c, row = m + tap * dil)
  const float* A = reinterpret_cast<const float*>(g.A);
  int kcur = 0, ctap = 0, ccol = sk4;   // this thread's k of the current fetch, split as tap * kc + ccol
  auto seek = [&](int k0) {
    kcur = k0 + sk4;
    if (g.kc) { ctap = kcur / g.kc; ccol = kcur - ctap * g.kc; } else { ctap = 0; ccol = kcur; }
  };
  float4 ra[4], rw[4];
  auto fetch = [&]() {     // loads the k-tile at kcur, then steps to the next one
    const bool kok = kcur < g.K;
    const long tapoff = (long)ctap * g.dil;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = srow + q * 32, m = m0 + row, n = n0 + row;
      const long arow = m + tapoff;
      ra[q] = (kok && m < g.M && arow < g.a_rows) ? *reinterpret_cast<const float4*>(A + arow * g.lda + ccol) : make_float4(0.f, 0.f, 0.f, 0.f);
      rw[q] = (kok && n < g.N) ? *reinterpret_cast<const float4*>(g.W + (long)n * g.ldw + kcur) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    kcur += BK; ccol += BK;
    if (g.kc) while (ccol >= g.kc) { ccol -= g.kc; ++ctap; }
  };
  f32x16 acc[2][2], tot[SLICED ? 2 : 1][SLICED ? 2 : 1];
  if constexpr (SLICED) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) tot[i][j][q] = 0.f;
  }
  const int nk_all = (g.K + BK - 1) / BK;
  const int nz = SLICED ? g.kslices : 1, per = (nk_all + nz - 1) / nz;
  for (int z = 0; z < nz; ++z) {   // same slice order as gemm_kernel
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int kt0 = z * per, kt1 = (kt0 + per < nk_all) ? kt0 + per : nk_all;
    if (kt0 < kt1) { seek(kt0 * BK); fetch(); }
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<float4*>(&As[(srow + q * 32) * LDT + sk4]) = ra[q];
        *reinterpret_cast<float4*>(&Ws[(srow + q * 32) * LDT + sk4]) = rw[q];
      }
      __syncthreads();
      if (kt + 1 < kt1 && !(g.dbg & 1)) fetch();
#pragma unroll
      for (int kk = 0; kk < BK; kk += 8) {
        float4 a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a[i] = *reinterpret_cast<const float4*>(&As[(wm * 64 + i * 32 + r) * LDT + kk + 4 * h]);
          b[i] = *reinterpret_cast<const float4*>(&Ws[(wn * 64 + i * 32 + r) * LDT + kk + 4 * h]);
        }
        // the four accumulators take turns, so consecutive MFMAs are independent; each still sees k in order
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
      __syncthreads();
    }
    if constexpr (SLICED) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 16; ++q) tot[i][j][q] += acc[i][j][q];
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + r;
      if (n >= g.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (m < g.M) g.C[(long)m * g.ldc + n] = gemm_epilogue(g, SLICED ? tot[SLICED ? i : 0][SLICED ? j : 0][q] : acc[i][j][q], m, n);
      }
    }
}

// ---- bf16 form: 32 x 64 tile, two waves, k-tile of 64 bf16, v_mfma_f32_32x32x16_bf16 ---------------------------------
// A rows are bf16 already (trunk output) or f32 converted on the way to LDS; W comes pre-converted.  LDS rows are
// 128 B + 16 B pad (9 slots: conflict-free ds_read_b128).  The problems this serves are skinny (attention.0: 13 056 x 128 x 2 560 at
// B = 256): with 64 x 64 tiles there were 1.6 workgroups per CU, each a chain of 40 k-tiles that waited a full memory latency per tile
// (45 us, 190 TFLOP/s).  Small tiles (816 workgroups) and operands fetched TWO k-tiles ahead keep several latencies in flight per CU;
// every output still sums its k in ascending order, so the numbers are those of the 64 x 64 form.
constexpr int BKH = 64, LDH = BKH * 2 + 16, BMH = 32;

template <bool A_BF16>
__global__ __launch_bounds__(128) void gemm_bf16_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char As[BMH * LDH];
  __shared__ __attribute__((aligned(16))) unsigned char Ws[BN * LDH];
  const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // 1-D grid: the NT column tiles of one row tile sit 8 block ids apart -- the same XCD under round-robin dispatch, started together --
  // so the second read of the A rows (67 MB for attention.0 at B = 256) is an L2 hit instead of a second trip to HBM
  const int nt_ = (g.N + BN - 1) / BN;
  const int n0 = (int)((blockIdx.x >> 3) % nt_) * BN, m0 = (int)(((blockIdx.x >> 3) / nt_) * 8 + (blockIdx.x & 7)) * BMH;
  if (m0 >= g.M) return;
  // staging: 8 chunks of 8 bf16 per row; 32 A rows -> 2 chunks per thread, 64 W rows -> 4.  The loads carry no guards: rows past M / N
  // are clamped to the last row (their products land in accumulator rows / columns that are never stored) and K is a whole number of
  // k-tiles (launch_gemm) -- a guarded load made every wait a vmcnt(0) and the two-tile prefetch below collapsed to one.
  const int srow = tid >> 3, sk8 = (tid & 7) * 8;
  constexpr int NA = A_BF16 ? 1 : 2;   // 16-B loads per A chunk
  const unsigned char* ap[2];
  const uint16_t* wp[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m = m0 + srow + q * 16;
    ap[q] = reinterpret_cast<const unsigned char*>(g.A) + ((long)(m < g.M ? m : g.M - 1) * g.lda + sk8) * (A_BF16 ? 2 : 4);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n0 + srow + q * 16;
    wp[q] = reinterpret_cast<const uint16_t*>(g.W_bf16) + (long)(n < g.N ? n : g.N - 1) * g.ldw + sk8;
  }
  struct Regs { uint4 a[2][NA]; uint4 w[4]; };   // one k-tile's operands of this thread
  auto fetch = [&](int k0, Regs& R) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < NA; ++e) R.a[q][e] = *reinterpret_cast<const uint4*>(ap[q] + (long)k0 * (A_BF16 ? 2 : 4) + e * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) R.w[q] = *reinterpret_cast<const uint4*>(wp[q] + k0);
  };
  f32x16 acc, tot;
#pragma unroll
  for (int q = 0; q < 16; ++q) { acc[q] = 0.f; tot[q] = 0.f; }
  // K is summed in g.kslices slices of `per` k-tiles (a function of K alone, launch_gemm): slice sums are added in slice order, ((0 + s0) + s1) + ...
  // Below 512 rows the slices run as blockIdx.y (round 6: attention.0 of ONE utterance was four workgroups walking forty k-tiles each, 21.7 us of a
  // 0.62-ms forward) and gemm_splitk_epilogue_kernel adds them in that order; above, this workgroup walks all of them and adds them itself --
  // the same additions, so a row's result does not depend on how many rows share its launch.
  const int per = (g.K / BKH) / g.kslices;
  const int kt0 = g.ksplit > 1 ? (int)blockIdx.y * per : 0;      // first k-tile of this workgroup
  const int nk = g.ksplit > 1 ? per : g.K / BKH;
  int in_slice = 0;
  // one k-tile from its registers through LDS to the matrix cores; the registers are refilled with k-tile `next` (always: a loop whose
  // loads sit behind a condition gets vmcnt(0) waits, and the tile fetched two steps ahead would be waited for at once)
  auto step = [&](Regs& R, int next) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      uint4 v;
      if constexpr (A_BF16) v = R.a[q][0];
      else {
        const float4 x = __builtin_bit_cast(float4, R.a[q][0]), y = __builtin_bit_cast(float4, R.a[q][NA - 1]);
        v = make_uint4(pack_bf16x2(x.x, x.y), pack_bf16x2(x.z, x.w), pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
      }
      *reinterpret_cast<uint4*>(As + (srow + q * 16) * LDH + sk8 * 2) = v;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(Ws + (srow + q * 16) * LDH + sk8 * 2) = R.w[q];
    __syncthreads();
    fetch((kt0 + next) * BKH, R);
#pragma unroll
    for (int kk = 0; kk < BKH; kk += 16) {
      const uint4 a = *reinterpret_cast<const uint4*>(As + r * LDH + (kk + 8 * h) * 2);
      const uint4 b = *reinterpret_cast<const uint4*>(Ws + (wn * 32 + r) * LDH + (kk + 8 * h) * 2);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    if (g.kslices > 1 && ++in_slice == per) {   // a slice is complete (wave-uniform): add it to the running total and start the next one from zero
#pragma unroll
      for (int q = 0; q < 16; ++q) { tot[q] += acc[q]; acc[q] = 0.f; }
      in_slice = 0;
    }
    __syncthreads();
  };
  Regs R0, R1;
  int kt = 0;
  if (nk & 1) {            // an odd tile count: the first tile on its own, pairs after it
    fetch(kt0 * BKH, R0);
    step(R0, 0);
    kt = 1;
  }
  const int last = nk - 1;
  if (kt < nk) {
    fetch((kt0 + kt) * BKH, R0);
    fetch((kt0 + kt + 1) * BKH, R1);
    for (; kt < nk; kt += 2) {   // tiles past the end are clamped to the last one: fetched again, never used
      step(R0, kt + 2 < last ? kt + 2 : last);
      step(R1, kt + 3 < last ? kt + 3 : last);
    }
  }
  const int n = n0 + wn * 32 + r;
  if (n >= g.N) return;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int m = m0 + (q & 3) + 8 * (q >> 2) + 4 * h;
    if (m >= g.M) continue;
    if (g.ksplit > 1) g.splitk_ws[((long)blockIdx.y * g.M + m) * g.N + n] = tot[q];      // this slice's sum (0 + s_z); the epilogue kernel adds the slices in order
    else g.C[(long)m * g.ldc + n] = gemm_epilogue(g, g.kslices > 1 ? tot[q] : acc[q], m, n);
  }
}

__global__ __launch_bounds__(256) void gemm_splitk_epilogue_kernel(GemmArgs g) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= (long)g.M * g.N) return;
  const int m = (int)(i / g.N), n = (int)(i % g.N);
  float v = 0.f;
  for (int z = 0; z < g.ksplit; ++z) v += g.splitk_ws[((long)z * g.M + m) * g.N + n];
  g.C[(long)m * g.ldc + n] = gemm_epilogue(g, v, m, n);
}

// the same for one row per workgroup, followed by l2_norm (loss.py:91-100) + F.normalize(eps 1e-12) (xvector.py:903) of the finished row:
// a thread keeps its columns n = tid, tid + 256, ... in registers and squares / sums them in exactly the order l2norm_kernel (pool.hip) reads
// them back from memory, so the x-vector is the same bits as with the two launches.  N <= 1024.
__global__ __launch_bounds__(256) void gemm_splitk_l2norm_kernel(GemmArgs g) {
  __shared__ float red[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  float c[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = tid + i * 256;
    c[i] = 0.f;
    if (n < g.N) {
      float v = 0.f;
      for (int z = 0; z < g.ksplit; ++z) v += g.splitk_ws[((long)z * g.M + m) * g.N + n];
      c[i] = gemm_epilogue(g, v, m, n);
      g.C[(long)m * g.ldc + n] = c[i];
      s = fmaf(c[i], c[i], s);
    }
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float n1 = sqrtf(red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (tid + i * 256 < g.N) { const float v = c[i] / n1; s2 = fmaf(v, v, s2); }
  s2 = wave_sum(s2);
  if ((tid & 63) == 0) red[tid >> 6] = s2;
  __syncthreads();
  const float n2 = fmaxf(sqrtf(red[0] + red[1] + red[2] + red[3]), 1e-12f);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (tid + i * 256 < g.N) g.l2_out[(long)m * g.N + tid + i * 256] = (c[i] / n1) / n2;
}

int launch_gemm(const GemmArgs& g_in, hipStream_t s) {
  GemmArgs g = g_in;
  if (g.l2_done) *g.l2_done = 0;
  g.ksplit = 1; g.kslices = 1;
  g.dbg = SK_AB_ENV_INT("SIDEKIT_AMD_GEMM_DBG", 0);
  SK_CHECK(g.M > 0 && g.N > 0 && g.K > 0, SK_EARG, "gemm: empty problem %dx%dx%d", g.M, g.N, g.K);
  if (g.W_bf16 && g.a_mode == A_PLAIN && g.kc == 0 && g.K % BKH == 0 && g.lda % 8 == 0 && g.ldw % 8 == 0) {   // whole k-tiles on the bf16 matrix cores (the model's two: K = 2560, 128)
    const int nkh = g.K / BKH;
    g.kslices = (nkh >= 16 && nkh % 8 == 0) ? 8 : 1;                               // K alone decides how the sum is grouped ...
    g.ksplit = (g.kslices > 1 && g.M <= 512 && g.splitk_ws) ? g.kslices : 1;      // ... M only where the slices run
    SK_CHECK(g.lda % 4 == 0, SK_EARG, "gemm: lda alignment");
    const dim3 gridh((unsigned)(cdiv(cdiv(g.M, BMH), 8) * 8 * cdiv(g.N, BN)), (unsigned)g.ksplit);
    if (g.a_bf16) hipLaunchKernelGGL(gemm_bf16_kernel<true>, gridh, dim3(128), 0, s, g);
    else hipLaunchKernelGGL(gemm_bf16_kernel<false>, gridh, dim3(128), 0, s, g);
    SK_HIP(hipGetLastError());
    if (g.ksplit > 1) {
      hipLaunchKernelGGL(gemm_splitk_epilogue_kernel, dim3((unsigned)(((long)g.M * g.N + 255) / 256)), dim3(256), 0, s, g);
      SK_HIP(hipGetLastError());
    }
    return SK_OK;
  }
  if (g.splitk_ws && g.K >= 1024) {   // the summation order depends on K alone; M only decides where the slices run
    const int nk = (g.K + BK - 1) / BK;
    g.kslices = nk / 8 < 32 ? (nk / 8 > 1 ? nk / 8 : 1) : 32;   // >= 8 k-tiles per slice, at most 32 slices
    if (g.M <= 512) g.ksplit = g.kslices;
  }
  SK_CHECK(g.K % 4 == 0 && g.ldw % 4 == 0, SK_EARG, "gemm: K=%d / ldw=%ld must be multiples of 4", g.K, g.ldw);
  SK_CHECK(g.a_mode != A_PLAIN || (g.lda % 4 == 0 && g.kc % 4 == 0), SK_EARG, "gemm: lda/kc alignment");
  if (g.ksplit == 1 && g.M >= 2048 && g.N >= 128 && g.a_mode == A_PLAIN && !g.a_bf16 && !SK_AB_GETENV("SIDEKIT_AMD_GEMM64")) {   // large problem: 128 x 128 tiles (same numbers)
    dim3 grid2(cdiv(g.M, BM2), cdiv(g.N, BN2));
    if (g.kslices > 1) hipLaunchKernelGGL(gemm128_kernel<true>, grid2, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(gemm128_kernel<false>, grid2, dim3(256), 0, s, g);
    SK_HIP(hipGetLastError());
    return SK_OK;
  }
  dim3 grid(cdiv(g.M, BM), cdiv(g.N, BN), g.ksplit);
  switch (g.a_mode) {
    case A_PLAIN: hipLaunchKernelGGL(gemm_kernel<LoadPlain>, grid, dim3(256), 0, s, g); break;
    case A_FRAMES: hipLaunchKernelGGL(gemm_kernel<LoadFrames>, grid, dim3(256), 0, s, g); break;
    case A_POWER: hipLaunchKernelGGL(gemm_kernel<LoadPower>, grid, dim3(256), 0, s, g); break;
    default: set_error("gemm: bad a_mode %d", g.a_mode); return SK_EARG;
  }
  SK_HIP(hipGetLastError());
  if (g.ksplit > 1) {
    if (g.l2_out && g.l2_done && g.N <= 1024) {
      hipLaunchKernelGGL(gemm_splitk_l2norm_kernel, dim3((unsigned)g.M), dim3(256), 0, s, g);
      *g.l2_done = 1;
    } else {
      hipLaunchKernelGGL(gemm_splitk_epilogue_kernel, dim3((unsigned)(((long)g.M * g.N + 255) / 256)), dim3(256), 0, s, g);
    }
    SK_HIP(hipGetLastError());
  }
  return SK_OK;
}

}  // namespace sk
