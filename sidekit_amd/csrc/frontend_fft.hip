// Power spectrogram of the log-mel front-end (torchaudio Spectrogram(n_fft=1024, win=400, hop=160,
// center=True reflect, power=2) after PreEmphasis, sidekit/nnet/preprocessor.py:253-261,278-279 and
// augmentation.py:63-74) as a real FFT instead of a 400 x 1032 DFT contraction (32x fewer FLOPs).
//
// One wavefront per frame.  The 1024-point real transform is a 512-point complex transform of
// z[n] = x[2n] + i x[2n+1] (x = windowed frame, non-zero only on [312, 712)) followed by the real-FFT
// split.  512 = 8^3: three Stockham (autosort) radix-8 passes, every lane owning one 8-point butterfly
// per pass; passes exchange data through a 4.6-KB per-wave LDS buffer (index padded by i>>3 so the
// stride-8 scatter of pass 1 is bank-conflict free); no workgroup barrier is ever needed.
// Output: |X[k]|^2, k = 0..512 (+3 zero pad columns), f32, row stride 516 -- or, with the mel projection fused, the
// log-mel row itself (the power spectrum then never leaves LDS: 211 MB less to write and read back at B=256 x 4 s).
#include "kernels.h"

namespace sk {

struct cf { float x, y; };
__device__ inline cf ldc(const float* table, int idx);
__device__ inline cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ inline cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
__device__ inline cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ inline cf mul_mi(cf a) { return {a.y, -a.x}; }  // a * (-i)

// 8-point DFT, natural-order output (decimation in frequency)
__device__ inline void fft8(cf* v) {
  const float s = 0.70710678118654752440f;
  cf a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
  cf c0 = csub(v[0], v[4]), c1 = csub(v[1], v[5]), c2 = csub(v[2], v[6]), c3 = csub(v[3], v[7]);
  c1 = {s * (c1.x + c1.y), s * (c1.y - c1.x)};    // * (1 - i)/sqrt(2)
  c2 = mul_mi(c2);                                // * (-i)
  c3 = {s * (c3.y - c3.x), -s * (c3.x + c3.y)};   // * (-1 - i)/sqrt(2)
  cf e0 = cadd(a0, a2), e1 = cadd(a1, a3), o0 = csub(a0, a2), o1 = mul_mi(csub(a1, a3));
  v[0] = cadd(e0, e1); v[4] = csub(e0, e1); v[2] = cadd(o0, o1); v[6] = csub(o0, o1);
  cf f0 = cadd(c0, c2), f1 = cadd(c1, c3), p0 = csub(c0, c2), p1 = mul_mi(csub(c1, c3));
  v[1] = cadd(f0, f1); v[5] = csub(f0, f1); v[3] = cadd(p0, p1); v[7] = csub(p0, p1);
}

__device__ inline int pad(int i) { return i + (i >> 3); }
// Table / sample reads can be made to bypass the CU's vector L1 (FFT_L1_BYPASS = 1: global_load ... sc1, L2-served) -- a diagnostic
// switch from the round-3 hunt for the two-lane hazard (it was not the cause, see the Makefile note on this file); plain loads by default.
#ifndef FFT_L1_BYPASS
#define FFT_L1_BYPASS 0
#endif
__device__ inline cf ldc(const float* table, int idx) {   // complex table entry idx (interleaved re, im)
  if (FFT_L1_BYPASS) {
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(table) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return cf{__builtin_bit_cast(float, (unsigned)(u & 0xffffffffu)), __builtin_bit_cast(float, (unsigned)(u >> 32))};
  }
  return reinterpret_cast<const cf*>(table)[idx];
}
__device__ inline float ldf(const float* p) {
  if (FFT_L1_BYPASS) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
__device__ inline int ldi(const int* p) {
  if (FFT_L1_BYPASS) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

// One utterance's samples: float32, or 16-bit PCM widened in the load as x / 32768 -- exact in f32, the very numbers the
// reference's soundfile / torchaudio decode hands its model (sidekit/bin/extract_xvectors.py:57-70), so both entry points give
// bit-identical features.  The selector is wave-uniform.
struct SampleRow {
  const void* base; long off; int pcm16;
  __device__ inline float operator[](int i) const {
    if (pcm16) {
      const short* p = reinterpret_cast<const short*>(base) + off + i;
      const short v = FFT_L1_BYPASS ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
      return (float)v * (1.0f / 32768.0f);
    }
    return ldf(reinterpret_cast<const float*>(base) + off + i);
  }
};

constexpr int FFT_WAVES = 4;
constexpr int FFT_BUF = 512 + 64;  // padded complex slots per wave

__global__ __launch_bounds__(FFT_WAVES * 64) void stft_power_fft_kernel(FftArgs a) {
  __shared__ __attribute__((aligned(16))) cf lds[FFT_WAVES * FFT_BUF];
  __shared__ float pws[FFT_WAVES][520];   // fused mel projection: the frame's power spectrum stays in LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * FFT_WAVES + wave;
  if (m >= a.M) return;
  cf* buf = lds + wave * FFT_BUF;
  int b, t;
  if (a.row_b) { b = a.row_b[m]; t = a.row_t[m]; } else { b = m / a.t_max; t = m % a.t_max; }
  const int L = a.nsamples ? a.nsamples[b] : a.nsamples_uniform;
  float* prow = a.mel_w ? nullptr : a.P + (long)m * a.ldp;
  if (t > L / a.hop) {  // frame beyond this utterance: keep the row defined (zero power)
    if (a.mel_w) { for (int j = lane; j < a.n_mels; j += 64) a.logmel[(long)m * a.ldl + j] = logf(1e-6f); }
    else { for (int k = lane; k < a.ldp; k += 64) prow[k] = 0.f; }
    return;
  }
  const SampleRow w{a.wav, (long)b * a.wav_ld, a.pcm16};
  const int i0 = t * a.hop - 200;  // sample index of window tap 0 (centre tap 200 sits on t*hop)
  auto xw = [&](int k) {           // windowed, pre-emphasised sample of window tap k, reflect-padded at the utterance edges
    int i = i0 + k;
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    const int p = (i == 0) ? 1 : i - 1;
    return ldf(a.window + k) * (w[i] - a.preemph * w[p]);
  };
  cf v[8];
  // ---- pass 0 (Ns = 1): inputs straight from the waveform, no twiddles
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int n = lane + 64 * r, k = 2 * n - 312;
    v[r] = (k >= 0 && k < 400) ? cf{xw(k), xw(k + 1)} : cf{0.f, 0.f};
  }
  fft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[pad(8 * lane + r)] = v[r];
  // ---- pass 1 (Ns = 8)
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[pad(lane + 64 * r)];
  {
    const int k = lane & 7;
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], ldc(a.tw512, k * r * 8));
  }
  fft8(v);
  {
    const int j0 = (lane >> 3) * 64 + (lane & 7);
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[pad(j0 + 8 * r)] = v[r];
  }
  // ---- pass 2 (Ns = 64)
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[pad(lane + 64 * r)];
#pragma unroll
  for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], ldc(a.tw512, lane * r));
  fft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[pad(lane + 64 * r)] = v[r];
  // ---- real-FFT split: X[k] = (Z[k] + conj Z[512-k]) / 2 - i W^k (Z[k] - conj Z[512-k]) / 2,  W = exp(-2 pi i / 1024)
  for (int k = lane; k < a.ldp; k += 64) {
    float pw = 0.f;
    if (k <= 512) {
      const cf zk = buf[pad(k & 511)], zc0 = buf[pad((512 - k) & 511)];
      const cf zc = {zc0.x, -zc0.y};
      const cf s = cadd(zk, zc), d = csub(zk, zc);
      const cf wd = cmul(ldc(a.tw1024, k), d);
      const float re = 0.5f * (s.x + wd.y), im = 0.5f * (s.y - wd.x);  // s/2 - i*wd/2
      pw = re * re + im * im;
    }
    if (a.mel_w) pws[wave][k] = pw;
    else prow[k] = pw;
  }
  if (a.mel_w) {  // each lane finishes filters lane and lane + 64: a dot product over the filter's own run of bins
    for (int k = 513 + lane; k < 520; k += 64) pws[wave][k] = 0.f;   // the 8-wide steps below may read past bin 512
      for (int j = lane; j < a.n_mels; j += 64) {
      const int k0 = ldi(a.mel_start + j), n = ldi(a.mel_len + j);
      float acc = 0.f;
      for (int i = 0; i < n; i += 8) {   // eight taps per step, all loads in flight together (mel_w is zero-padded to a multiple of 8 rows)
        float wv[8], pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { wv[q] = ldf(a.mel_w + (i + q) * a.n_mels + j); pv[q] = pws[wave][k0 + i + q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(wv[q], pv[q], acc);
      }
      a.logmel[(long)m * a.ldl + j] = logf(acc + 1e-6f);
    }
  }
}

// ---- n_fft = 2048, win = 1024 (the MFCC front-end of the TDNN x-vector, preprocessor.py:65-76) ----------------------------
// Same plan one size up: 1024-point complex transform of z[n] = x[2n] + i x[2n+1] (x non-zero on [512, 1536) of the frame) as
// Stockham passes of radix 8, 8, 16 -- a lane owns two 8-point butterflies in each of the first two passes and one 16-point
// butterfly (two 8-point ones on the even / odd inputs + one radix-2 step) in the last -- then the real-FFT split.  As a
// (frames x 1024) x (1024 x 2050) DFT contraction on the exact-f32 MFMA this was 56 % of the TDNN forward (8.0 of 14.1 ms at
// 512 x 6 s); the transform needs 1/40 of the FLOPs.
constexpr int FFT2K_BUF = 1024 + 128;   // padded complex slots per wave

__device__ inline void fft16(cf* v) {   // natural order in, natural order out
  cf e[8], o[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) { e[r] = v[2 * r]; o[r] = v[2 * r + 1]; }
  fft8(e);
  fft8(o);
  // w16^k = exp(-2 pi i k / 16), k = 0..7
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
  const cf w[8] = {{1.f, 0.f}, {c1, -s1}, {h, -h}, {s1, -c1}, {0.f, -1.f}, {-s1, -c1}, {-h, -h}, {-c1, -s1}};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const cf t = cmul(w[k], o[k]);
    v[k] = cadd(e[k], t);
    v[k + 8] = csub(e[k], t);
  }
}

__global__ __launch_bounds__(FFT_WAVES * 64) void stft_power_fft2k_kernel(FftArgs a) {
  __shared__ __attribute__((aligned(16))) cf lds[FFT_WAVES * FFT2K_BUF];
  __shared__ float pws[FFT_WAVES][1032];   // fused mel projection: the frame's power spectrum stays in LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * FFT_WAVES + wave;
  if (m >= a.M) return;
  cf* buf = lds + wave * FFT2K_BUF;
  // a.tw512: exp(-2 pi i m / 1024), m = 0..1023;  a.tw1024: exp(-2 pi i k / 2048), k = 0..1024
  int b, t;
  if (a.row_b) { b = a.row_b[m]; t = a.row_t[m]; } else { b = m / a.t_max; t = m % a.t_max; }
  const int L = a.nsamples ? a.nsamples[b] : a.nsamples_uniform;
  float* prow = a.mel_w ? nullptr : a.P + (long)m * a.ldp;
  if (t > L / a.hop) {  // frame beyond this utterance: keep the row defined (zero power)
    if (a.mel_w) { for (int j = lane; j < a.n_mels; j += 64) a.logmel[(long)m * a.ldl + j] = logf(1e-6f); }
    else { for (int k = lane; k < a.ldp; k += 64) prow[k] = 0.f; }
    return;
  }
  const SampleRow w{a.wav, (long)b * a.wav_ld, a.pcm16};
  const int i0 = t * a.hop - 512;  // sample index of window tap 0 (centre tap 512 sits on t*hop)
  auto xw = [&](int k) {           // windowed, pre-emphasised sample of window tap k, reflect-padded at the utterance edges
    int i = i0 + k;
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    const int p = (i == 0) ? 1 : i - 1;
    return ldf(a.window + k) * (w[i] - a.preemph * w[p]);
  };
  cf v[16];
  // ---- pass 0 (radix 8, Ns = 1): butterfly j takes z[j + 128 r]; only r = 2..5 lie inside the window's support
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = lane + 64 * u;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = 2 * (j + 128 * r) - 512;
      v[r] = (r >= 2 && r < 6) ? cf{xw(k), xw(k + 1)} : cf{0.f, 0.f};
    }
    fft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[pad(8 * j + r)] = v[r];
  }
  // ---- pass 1 (radix 8, Ns = 8)
  {
    cf x2[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 8; ++r) x2[u][r] = buf[pad(lane + 64 * u + 128 * r)];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane + 64 * u, k = j & 7;
#pragma unroll
      for (int r = 1; r < 8; ++r) x2[u][r] = cmul(x2[u][r], ldc(a.tw512, k * r * 16));
      fft8(x2[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = lane + 64 * u, j0 = (j >> 3) * 64 + (j & 7);
#pragma unroll
      for (int r = 0; r < 8; ++r) buf[pad(j0 + 8 * r)] = x2[u][r];
    }
  }
  // ---- pass 2 (radix 16, Ns = 64)
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = buf[pad(lane + 64 * r)];
#pragma unroll
  for (int r = 1; r < 16; ++r) v[r] = cmul(v[r], ldc(a.tw512, lane * r));
  fft16(v);
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[pad(lane + 64 * r)] = v[r];
  // ---- real-FFT split: X[k] = (Z[k] + conj Z[1024-k]) / 2 - i W^k (Z[k] - conj Z[1024-k]) / 2,  W = exp(-2 pi i / 2048)
  for (int k = lane; k < a.ldp; k += 64) {
    float pw = 0.f;
    if (k <= 1024) {
      const cf zk = buf[pad(k & 1023)], zc0 = buf[pad((1024 - k) & 1023)];
      const cf zc = {zc0.x, -zc0.y};
      const cf s = cadd(zk, zc), d = csub(zk, zc);
      const cf wd = cmul(ldc(a.tw1024, k), d);
      const float re = 0.5f * (s.x + wd.y), im = 0.5f * (s.y - wd.x);  // s/2 - i*wd/2
      pw = re * re + im * im;
    }
    if (a.mel_w) pws[wave][k] = pw;
    else prow[k] = pw;
  }
  if (a.mel_w) {  // each lane finishes filters lane and lane + 64: a dot product over the filter's own run of bins
    for (int k = a.ldp + lane; k < 1032; k += 64) pws[wave][k] = 0.f;   // the 8-wide steps below may read past the last bin
    for (int j = lane; j < a.n_mels; j += 64) {
      const int k0 = ldi(a.mel_start + j), n = ldi(a.mel_len + j);
      float acc = 0.f;
      for (int i = 0; i < n; i += 8) {
        float wv[8], pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { wv[q] = ldf(a.mel_w + (i + q) * a.n_mels + j); pv[q] = pws[wave][k0 + i + q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(wv[q], pv[q], acc);
      }
      a.logmel[(long)m * a.ldl + j] = logf(acc + 1e-6f);
    }
  }
}

int launch_stft_power_fft(const FftArgs& a, hipStream_t s) {
  SK_CHECK(a.n_fft == 1024 || a.n_fft == 2048, SK_EARG, "stft_power_fft: n_fft = %d (1024 or 2048)", a.n_fft);
  const int nb = a.n_fft / 2 + 1;
  SK_CHECK(a.M > 0 && (a.mel_w ? (a.ldp >= nb && a.ldp <= nb + 7 && a.logmel && a.mel_start && a.mel_len && a.n_mels > 0) : a.ldp >= nb), SK_EARG, "stft_power_fft: bad arguments");
  if (a.n_fft == 1024) hipLaunchKernelGGL(stft_power_fft_kernel, dim3(cdiv(a.M, FFT_WAVES)), dim3(FFT_WAVES * 64), 0, s, a);
  else hipLaunchKernelGGL(stft_power_fft2k_kernel, dim3(cdiv(a.M, FFT_WAVES)), dim3(FFT_WAVES * 64), 0, s, a);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
