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
//
// The kernel is bound by VALU issue (a wave64 instruction occupies its 16-lane SIMD for four cycles), so everything around the
// butterflies is kept short (round 3: 1 086 -> 534 VALU instructions per frame, 289 -> 200 us at 256 x 4 s; the MFCC kernel 620 -> 375 us
// at 512 x 6 s):
//  * the wave index is made scalar (readfirstlane): frame, utterance, length and every base address live in SGPRs, loads take the
//    SGPR-base + lane-offset form;
//  * interior frames (all but two at either end of an utterance) take their samples without the reflect logic: per complex input one
//    12-byte sample load and one 8-byte window load; only the four radix-8 inputs that can fall inside the window's support are
//    touched, and pass 0 runs the butterfly that knows the other four are zero;
//  * the real-FFT split works by mirror pairs on the lane's own outputs Z[lane + 64 r] while they are in registers: one LDS read
//    (the partner Z[N/2 - k]) and one twiddle give the powers of bins k AND N/2 - k; the power row overwrites the transform's buffer
//    (a wave's LDS instructions execute in order), which is what lets eight workgroups share a CU;
//  * the mel projection runs on a host-made list of 8-tap chunks (a filter of n bins = ceil(n / 8) chunks, 153 chunks for the 80
//    triangles instead of 2 x 4 rounds in which every lane waits for the widest filter): one chunk per lane and round, chunk sums
//    through LDS, then each filter adds its (consecutive) chunks in order; log through v_log_f32.
// No packed-f32 instructions: the file is built without SLP vectorisation (csrc/Makefile says why) and forms none by hand.
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
__device__ inline cf ldc(const float* table, int idx) { return reinterpret_cast<const cf*>(table)[idx]; }   // complex table entry (interleaved re, im)

constexpr int FFT_WAVES = 4;
constexpr int FFT_BUF = 512 + 64;  // padded complex slots per wave

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

// complex product with explicit FMAs (the file is built with -ffp-contract=off)
__device__ inline cf cmulf(cf a, cf b) { return {fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)}; }
// Real-FFT split of one mirror pair: from Z[k], Z[NH - k] and W^k (W = exp(-2 pi i / n_fft)) the two powers |X[k]|^2 and |X[NH - k]|^2:
//   X[k] = s/2 - i W^k d/2,  X[NH - k] = conj(s/2 + i W^k d/2),  s = Z[k] + conj Z[NH - k],  d = Z[k] - conj Z[NH - k]
// (W^(NH - k) = -conj W^k, and the mirror bin's s and d are conj(s) and -conj(d), exactly)
__device__ inline void split_pair(cf zk, cf zm, cf w, float& pk, float& pm) {
  const cf s = {zk.x + zm.x, zk.y - zm.y}, d = {zk.x - zm.x, zk.y + zm.y};
  const cf wd = cmulf(w, d);
  const float re = 0.5f * (s.x + wd.y), im = 0.5f * (s.y - wd.x);
  const float rm = 0.5f * (s.x - wd.y), jm = 0.5f * (s.y + wd.x);
  pk = fmaf(re, re, im * im);
  pm = fmaf(rm, rm, jm * jm);
}

// 8-point DFT of (0, 0, v2, v3, v4, v5, 0, 0)
__device__ inline void fft8_mid4(cf* v) {
  const float s = 0.70710678118654752440f;
  const cf a0 = v[4], a1 = v[5], a2 = v[2], a3 = v[3];
  const cf c0 = {-v[4].x, -v[4].y};
  cf c1 = {-v[5].x, -v[5].y}, c2 = v[2], c3 = v[3];
  c1 = {s * (c1.x + c1.y), s * (c1.y - c1.x)};
  c2 = mul_mi(c2);
  c3 = {s * (c3.y - c3.x), -s * (c3.x + c3.y)};
  cf e0 = cadd(a0, a2), e1 = cadd(a1, a3), o0 = csub(a0, a2), o1 = mul_mi(csub(a1, a3));
  v[0] = cadd(e0, e1); v[4] = csub(e0, e1); v[2] = cadd(o0, o1); v[6] = csub(o0, o1);
  cf f0 = cadd(c0, c2), f1 = cadd(c1, c3), p0 = csub(c0, c2), p1 = mul_mi(csub(c1, c3));
  v[1] = cadd(f0, f1); v[5] = csub(f0, f1); v[3] = cadd(p0, p1); v[7] = csub(p0, p1);
}

// windowed, pre-emphasised complex input z = (x[k], x[k + 1]) of window tap k (even) of an INTERIOR frame (no reflection can occur):
// three samples and one 8-byte window pair; `row0` points at the sample under tap 0
template <bool PCM16>
__device__ inline cf frame_input(const void* row0, const float* window, int k, bool ok, float pe) {
  const int kk = ok ? k : 0;
  float sm, s0, s1;
  if (PCM16) {
    const short* w = reinterpret_cast<const short*>(row0) + kk;
    sm = (float)w[-1] * (1.0f / 32768.0f); s0 = (float)w[0] * (1.0f / 32768.0f); s1 = (float)w[1] * (1.0f / 32768.0f);
  } else {
    const float* w = reinterpret_cast<const float*>(row0) + kk;
    sm = w[-1]; s0 = w[0]; s1 = w[1];
  }
  const cf wn = *reinterpret_cast<const cf*>(window + kk);
  const float x0 = wn.x * (s0 - pe * sm), x1 = wn.y * (s1 - pe * s0);
  return ok ? cf{x0, x1} : cf{0.f, 0.f};
}
// the same for a frame that overhangs the utterance: indices reflected (the STFT's padding), predecessor of the REFLECTED index
// (PreEmphasis runs before the padding and repeats sample 1 in front of sample 0, augmentation.py:63-74)
template <bool PCM16>
__device__ inline cf frame_input_edge(const void* row, const float* window, int i0, int k, bool ok, int L, float pe) {
  const int kk = ok ? k : 0;
  auto at = [&](int i) { return PCM16 ? (float)reinterpret_cast<const short*>(row)[i] * (1.0f / 32768.0f) : reinterpret_cast<const float*>(row)[i]; };
  auto xw = [&](int kt) {
    int i = i0 + kt;
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    const int p = (i == 0) ? 1 : i - 1;
    return window[kt] * (at(i) - pe * at(p));
  };
  const float x0 = xw(kk), x1 = xw(kk + 1);
  return ok ? cf{x0, x1} : cf{0.f, 0.f};
}

// chunked mel projection of one wave's power row (LDS) -> log-mel row; `part` is per-wave LDS scratch of >= mel_chunks + 8 floats
__device__ inline void mel_chunks_project(const FftArgs& a, const float* pw_row, float* part, int lane, long m) {
  for (int c = lane; c < a.mel_chunks; c += 64) {
    const int k0 = a.mel_ck0[c];
    const float4 w0 = reinterpret_cast<const float4*>(a.mel_cw)[2 * c], w1 = reinterpret_cast<const float4*>(a.mel_cw)[2 * c + 1];
    const float* p = pw_row + k0;
    float acc = w0.x * p[0];
    acc = fmaf(w0.y, p[1], acc); acc = fmaf(w0.z, p[2], acc); acc = fmaf(w0.w, p[3], acc);
    acc = fmaf(w1.x, p[4], acc); acc = fmaf(w1.y, p[5], acc); acc = fmaf(w1.z, p[6], acc); acc = fmaf(w1.w, p[7], acc);
    part[c] = acc;
  }
  for (int j = lane; j < a.n_mels; j += 64) {
    const int fm = a.mel_fmeta[j], cb = fm & 0xffff, nc = fm >> 16;
    const float x0 = part[cb], x1 = part[cb + 1], x2 = part[cb + 2], x3 = part[cb + 3];   // in flight together; `part` is readable 8 floats past the last chunk
    float acc = x0 + (nc > 1 ? x1 : 0.f);
    acc += nc > 2 ? x2 : 0.f;
    acc += nc > 3 ? x3 : 0.f;
    for (int q = 4; q < nc; ++q) acc += part[cb + q];
    a.logmel[m * a.ldl + j] = __builtin_amdgcn_logf(acc + 1e-6f) * 0.69314718055994530942f;   // v_log_f32 (log2, 1 ulp); the argument is >= 1e-6, never denormal
  }
}

template <bool PCM16, bool MEL>
__global__ __launch_bounds__(FFT_WAVES * 64) void stft_power_fft_kernel(FftArgs a) {
  __shared__ __attribute__((aligned(16))) cf lds[FFT_WAVES * FFT_BUF];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // everything per frame below is scalar
  const int m = blockIdx.x * FFT_WAVES + wave;
  if (m >= a.M) return;
  cf* buf = lds + wave * FFT_BUF;
  int b, t;
  if (a.row_b) { b = a.row_b[m]; t = a.row_t[m]; } else { b = m / a.t_max; t = m % a.t_max; }
  const int L = a.nsamples ? a.nsamples[b] : a.nsamples_uniform;
  float* prow = MEL ? nullptr : a.P + (long)m * a.ldp;
  if (t > L / a.hop) {  // frame beyond this utterance: keep the row defined (zero power)
    if (MEL) { for (int j = lane; j < a.n_mels; j += 64) a.logmel[(long)m * a.ldl + j] = logf(1e-6f); }
    else { for (int k = lane; k < a.ldp; k += 64) prow[k] = 0.f; }
    return;
  }
  const void* row = PCM16 ? (const void*)(reinterpret_cast<const short*>(a.wav) + (long)b * a.wav_ld) : (const void*)(reinterpret_cast<const float*>(a.wav) + (long)b * a.wav_ld);
  const int i0 = t * a.hop - 200;                        // sample index of window tap 0 (centre tap 200 sits on t*hop)
  const bool interior = i0 >= 1 && i0 + 400 <= L;        // wave-uniform
  cf v[8];
  // ---- pass 0 (Ns = 1): inputs n = lane + 64 r straight from the waveform; taps k = 2 n - 312 in [0, 400) <=> r in 2..5
  v[0] = v[1] = v[6] = v[7] = cf{0.f, 0.f};
  if (interior) {
    const void* row0 = PCM16 ? (const void*)(reinterpret_cast<const short*>(row) + i0) : (const void*)(reinterpret_cast<const float*>(row) + i0);
#pragma unroll
    for (int r = 2; r < 6; ++r) {
      const int k = 2 * lane + 128 * r - 312;
      v[r] = frame_input<PCM16>(row0, a.window, k, (r == 3 || r == 4) ? true : (unsigned)k < 400u, a.preemph);
    }
  } else {
#pragma unroll
    for (int r = 2; r < 6; ++r) {
      const int k = 2 * lane + 128 * r - 312;
      v[r] = frame_input_edge<PCM16>(row, a.window, i0, k, (unsigned)k < 400u, L, a.preemph);
    }
  }
  fft8_mid4(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[pad(8 * lane + r)] = v[r];
  // ---- pass 1 (Ns = 8)
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = buf[pad(lane + 64 * r)];
  {
    const int k = lane & 7;
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cmulf(v[r], ldc(a.tw512, k * r * 8));
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
  for (int r = 1; r < 8; ++r) v[r] = cmulf(v[r], ldc(a.tw512, lane * r));
  fft8(v);
#pragma unroll
  for (int r = 0; r < 8; ++r) buf[pad(lane + 64 * r)] = v[r];
  // ---- real-FFT split by mirror pairs: the lane's own Z[k], k = lane + 64 r, r = 0..3 (k < 256), against Z[512 - k] from LDS gives the
  //      powers of bins k and 512 - k (k = 0: Z[0] against itself gives bins 0 and 512); lane 0 also owns the self-paired bin 256 (r = 4)
  const cf* tw = reinterpret_cast<const cf*>(a.tw1024) + lane;
  const int pk0 = (512 - lane) & 511;
  cf zm[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) zm[r] = buf[pad((pk0 - 64 * r) & 511)];   // Z[512 - k]; k = 0 pairs with itself and its mirror power is bin 512's
  // every LDS read of the transform has been issued: the power row may overwrite it (a wave's LDS instructions execute in order)
  float* pw_row = MEL ? reinterpret_cast<float*>(buf) : prow;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float p_k, p_m;
    split_pair(v[r], zm[r], tw[64 * r], p_k, p_m);
    pw_row[lane + 64 * r] = p_k;
    pw_row[512 - lane - 64 * r] = p_m;
  }
  if (lane == 0) {
    float p_k, p_m;
    split_pair(v[4], v[4], tw[256], p_k, p_m);     // bin 256 pairs with itself
    pw_row[256] = p_k;
  }
  for (int k = 513 + lane; k < (MEL ? 520 : a.ldp); k += 64) pw_row[k] = 0.f;   // zero pad columns
  if (MEL) mel_chunks_project(a, pw_row, pw_row + 528, lane, m);   // chunk sums behind the power row, still inside this wave's buffer
}

template <bool PCM16, bool MEL>
__global__ __launch_bounds__(FFT_WAVES * 64) void stft_power_fft2k_kernel(FftArgs a) {
  __shared__ __attribute__((aligned(16))) cf lds[FFT_WAVES * FFT2K_BUF];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = blockIdx.x * FFT_WAVES + wave;
  if (m >= a.M) return;
  cf* buf = lds + wave * FFT2K_BUF;
  // a.tw512: exp(-2 pi i m / 1024), m = 0..1023;  a.tw1024: exp(-2 pi i k / 2048), k = 0..1024
  int b, t;
  if (a.row_b) { b = a.row_b[m]; t = a.row_t[m]; } else { b = m / a.t_max; t = m % a.t_max; }
  const int L = a.nsamples ? a.nsamples[b] : a.nsamples_uniform;
  float* prow = MEL ? nullptr : a.P + (long)m * a.ldp;
  if (t > L / a.hop) {  // frame beyond this utterance: keep the row defined (zero power)
    if (MEL) { for (int j = lane; j < a.n_mels; j += 64) a.logmel[(long)m * a.ldl + j] = logf(1e-6f); }
    else { for (int k = lane; k < a.ldp; k += 64) prow[k] = 0.f; }
    return;
  }
  const void* row = PCM16 ? (const void*)(reinterpret_cast<const short*>(a.wav) + (long)b * a.wav_ld) : (const void*)(reinterpret_cast<const float*>(a.wav) + (long)b * a.wav_ld);
  const int i0 = t * a.hop - 512;                        // sample index of window tap 0 (centre tap 512 sits on t*hop)
  const bool interior = i0 >= 1 && i0 + 1024 <= L;       // wave-uniform
  cf v[16];
  // ---- pass 0 (radix 8, Ns = 1): butterfly j takes z[j + 128 r]; only r = 2..5 lie inside the window's support (all of it)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int j = lane + 64 * u;
    v[0] = v[1] = v[6] = v[7] = cf{0.f, 0.f};
    if (interior) {
      const void* row0 = PCM16 ? (const void*)(reinterpret_cast<const short*>(row) + i0) : (const void*)(reinterpret_cast<const float*>(row) + i0);
#pragma unroll
      for (int r = 2; r < 6; ++r) v[r] = frame_input<PCM16>(row0, a.window, 2 * (j + 128 * r) - 512, true, a.preemph);
    } else {
#pragma unroll
      for (int r = 2; r < 6; ++r) v[r] = frame_input_edge<PCM16>(row, a.window, i0, 2 * (j + 128 * r) - 512, true, L, a.preemph);
    }
    fft8_mid4(v);
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
      for (int r = 1; r < 8; ++r) x2[u][r] = cmulf(x2[u][r], ldc(a.tw512, k * r * 16));
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
  for (int r = 1; r < 16; ++r) v[r] = cmulf(v[r], ldc(a.tw512, lane * r));
  fft16(v);
#pragma unroll
  for (int r = 0; r < 16; ++r) buf[pad(lane + 64 * r)] = v[r];
  // ---- real-FFT split by mirror pairs: k = lane + 64 r, r = 0..7 (k < 512) against Z[1024 - k]; lane 0: bins 512 (r = 8) and 0 / 1024
  const cf* tw = reinterpret_cast<const cf*>(a.tw1024) + lane;
  const int pk0 = (1024 - lane) & 1023;
  cf zm[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) zm[r] = buf[pad((pk0 - 64 * r) & 1023)];
  float* pw_row = MEL ? reinterpret_cast<float*>(buf) : prow;   // as in the 1024-point kernel: the power row overwrites the transform
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float p_k, p_m;
    split_pair(v[r], zm[r], tw[64 * r], p_k, p_m);
    pw_row[lane + 64 * r] = p_k;
    pw_row[1024 - lane - 64 * r] = p_m;
  }
  if (lane == 0) {
    float p_k, p_m;
    split_pair(v[8], v[8], tw[512], p_k, p_m);
    pw_row[512] = p_k;
  }
  for (int k = 1025 + lane; k < (MEL ? 1032 : a.ldp); k += 64) pw_row[k] = 0.f;
  if (MEL) mel_chunks_project(a, pw_row, pw_row + 1040, lane, m);
}

int launch_stft_power_fft(const FftArgs& a, hipStream_t s) {
  SK_CHECK(a.n_fft == 1024 || a.n_fft == 2048, SK_EARG, "stft_power_fft: n_fft = %d (1024 or 2048)", a.n_fft);
  const int nb = a.n_fft / 2 + 1;
  SK_CHECK(a.M > 0 && (a.mel_cw ? (a.ldp >= nb && a.ldp <= nb + 7 && a.logmel && a.mel_ck0 && a.mel_fmeta && a.n_mels > 0 && a.mel_chunks > 0 &&
                                   a.mel_chunks % 64 == 0 && a.mel_chunks + 8 <= 2 * (a.n_fft == 1024 ? FFT_BUF : FFT2K_BUF) - (a.n_fft == 1024 ? 528 : 1040))
                                : a.ldp >= nb),
           SK_EARG, "stft_power_fft: bad arguments");
  const dim3 grid(cdiv(a.M, FFT_WAVES)), block(FFT_WAVES * 64);
  {
#define SK_FFT_LAUNCH(K) do { \
    if (a.pcm16) { if (a.mel_cw) hipLaunchKernelGGL((K<true, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((K<true, false>), grid, block, 0, s, a); } \
    else { if (a.mel_cw) hipLaunchKernelGGL((K<false, true>), grid, block, 0, s, a); else hipLaunchKernelGGL((K<false, false>), grid, block, 0, s, a); } } while (0)
    if (a.n_fft == 1024) SK_FFT_LAUNCH(stft_power_fft_kernel); else SK_FFT_LAUNCH(stft_power_fft2k_kernel);
#undef SK_FFT_LAUNCH
  }
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
