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
  const float* w = a.wav + (long)b * a.wav_ld;
  const int i0 = t * a.hop - 200;  // sample index of window tap 0 (centre tap 200 sits on t*hop)
  auto xw = [&](int k) {           // windowed, pre-emphasised sample of window tap k, reflect-padded at the utterance edges
    int i = i0 + k;
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    const int p = (i == 0) ? 1 : i - 1;
    return a.window[k] * (w[i] - a.preemph * w[p]);
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
    for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], reinterpret_cast<const cf*>(a.tw512)[k * r * 8]);
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
  for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], reinterpret_cast<const cf*>(a.tw512)[lane * r]);
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
      const cf wd = cmul(reinterpret_cast<const cf*>(a.tw1024)[k], d);
      const float re = 0.5f * (s.x + wd.y), im = 0.5f * (s.y - wd.x);  // s/2 - i*wd/2
      pw = re * re + im * im;
    }
    if (a.mel_w) pws[wave][k] = pw;
    else prow[k] = pw;
  }
  if (a.mel_w) {  // each lane finishes filters lane and lane + 64: a dot product over the filter's own run of bins
    for (int k = 513 + lane; k < 520; k += 64) pws[wave][k] = 0.f;   // the 8-wide steps below may read past bin 512
    for (int j = lane; j < a.n_mels; j += 64) {
      const int k0 = a.mel_start[j], n = a.mel_len[j];
      float acc = 0.f;
      for (int i = 0; i < n; i += 8) {   // eight taps per step, all loads in flight together (mel_w is zero-padded to a multiple of 8 rows)
        float wv[8], pv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { wv[q] = a.mel_w[(i + q) * a.n_mels + j]; pv[q] = pws[wave][k0 + i + q]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(wv[q], pv[q], acc);
      }
      a.logmel[(long)m * a.ldl + j] = logf(acc + 1e-6f);
    }
  }
}

int launch_stft_power_fft(const FftArgs& a, hipStream_t s) {
  SK_CHECK(a.M > 0 && (a.mel_w ? (a.ldp >= 513 && a.ldp <= 520 && a.logmel && a.mel_start && a.mel_len && a.n_mels > 0) : a.ldp >= 513), SK_EARG, "stft_power_fft: bad arguments");
  hipLaunchKernelGGL(stft_power_fft_kernel, dim3(cdiv(a.M, FFT_WAVES)), dim3(FFT_WAVES * 64), 0, s, a);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
