// Sample-rate conversion of the extraction driver: the reference resamples a file whose rate differs from the model's with
// torchaudio.transforms.Resample(orig_freq, new_freq) (sidekit/bin/extract_xvectors.py:141-143).  torchaudio is pinned at 0.8.2
// and NOT vendored (install.sh:36): its arithmetic -- compliance.kaldi.resample_waveform, windowed-sinc interpolation with
// lowpass_filter_width 6 and roll-off 0.99, evaluated as a strided convolution with one filter per output phase -- is restated
// here from the published algorithm; PARITY UNPINNED at this boundary, like the mel front-end.
//
//   g = gcd(orig, new); O = orig / g; N = new / g; base = min(O, N) * 0.99; width = ceil(6 * O / base)
//   filter[i][k] = sinc(pi t) * cos^2(pi t / 12) * base / O,  t = clamp((-i / N + (k - width) / O) * base, -6, 6),  k in [0, 2 width + O)
//   out[b * N + i] = sum_k filter[i][k] * x[b * O + k - width]      (x zero outside [0, n)),   n_out = ceil(N * n / O)
//
// One thread per output sample; the N x K filter table (a few hundred KB at most, 44.1 -> 16 kHz: 160 x 475) is built on the host
// in double precision, rounded once, cached per (orig, new, device) and read through L2.
#include <math.h>

#include <map>
#include <mutex>
#include <numeric>
#include <tuple>
#include <vector>

#include "../../include/sidekit_amd.h"
#include "kernels.h"

namespace sk {

struct ResampleTable { float* d = nullptr; int O = 0, N = 0, width = 0, K = 0; };
static std::mutex g_rs_mu;
static std::map<std::tuple<int, int, int>, ResampleTable> g_rs_tables;

static int resample_table(int orig, int nw, ResampleTable* out) {
  int dev = 0;
  SK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_rs_mu);
  auto key = std::make_tuple(orig, nw, dev);
  auto it = g_rs_tables.find(key);
  if (it != g_rs_tables.end()) { *out = it->second; return SK_OK; }
  const int g = std::gcd(orig, nw);
  ResampleTable t;
  t.O = orig / g; t.N = nw / g;
  const double lpw = 6.0;
  const double base = (double)(t.O < t.N ? t.O : t.N) * 0.99;
  t.width = (int)ceil(lpw * t.O / base);
  t.K = 2 * t.width + t.O;
  SK_CHECK((size_t)t.N * t.K <= (size_t)1 << 26, SK_EARG, "sk_resample: %d -> %d Hz needs a %d x %d filter table (rates with a tiny common divisor)", orig, nw, t.N, t.K);
  std::vector<float> h((size_t)t.N * t.K);
  for (int i = 0; i < t.N; ++i)
    for (int k = 0; k < t.K; ++k) {
      double x = (-(double)i / t.N + (double)(k - t.width) / t.O) * base;
      x = x < -lpw ? -lpw : (x > lpw ? lpw : x);
      const double a = x * M_PI;
      const double win = cos(a / lpw / 2.0);
      const double s = a == 0.0 ? 1.0 : sin(a) / a;
      h[(size_t)i * t.K + k] = (float)(s * win * win * (base / t.O));
    }
  SK_HIP(hipMalloc((void**)&t.d, h.size() * 4));
  SK_HIP(hipMemcpy(t.d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  g_rs_tables[key] = t;
  *out = t;
  return SK_OK;
}

template <typename T>
__global__ __launch_bounds__(256) void resample_kernel(const T* __restrict__ x, long n_in, const float* __restrict__ filt, int O, int N, int width, int K,
                                                       float* __restrict__ out, long n_out) {
  const long j = blockIdx.x * 256L + threadIdx.x;
  if (j >= n_out) return;
  const long blk = j / N;
  const int ph = (int)(j - blk * N);
  const float* f = filt + (long)ph * K;
  const long s0 = blk * O - width;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {      // conv1d's accumulation order: taps in ascending k
    const long s = s0 + k;
    float v = 0.f;
    if (s >= 0 && s < n_in) {
      if constexpr (sizeof(T) == 2) v = (float)x[s] * (1.0f / 32768.0f);
      else v = x[s];
    }
    acc = fmaf(f[k], v, acc);
  }
  out[j] = acc;
}

}  // namespace sk

using namespace sk;

extern "C" int sk_resample(const void* d_in, int32_t in_dtype, int64_t n_in, int32_t orig_freq, int32_t new_freq, float* d_out,
                           int64_t out_capacity, int64_t* n_out, void* stream) {
  SK_CHECK(n_out && orig_freq > 0 && new_freq > 0 && n_in >= 0, SK_EARG, "sk_resample: bad arguments");
  SK_CHECK(in_dtype == XT_F32 || in_dtype == XT_I16, SK_EARG, "sk_resample: input must be XT_F32 or XT_I16");
  const int g = std::gcd(orig_freq, new_freq);
  const int64_t O = orig_freq / g, N = new_freq / g;
  *n_out = (N * n_in + O - 1) / O;                      // int(math.ceil(new_freq * length / orig_freq))
  if (!d_out) return SK_OK;                             // size query
  SK_CHECK(d_in || n_in == 0, SK_EARG, "sk_resample: null input");
  SK_CHECK(out_capacity >= *n_out, SK_EARG, "sk_resample: output holds %lld samples, %lld needed", (long long)out_capacity, (long long)*n_out);
  if (*n_out == 0) return SK_OK;
  ResampleTable t;
  SK_TRY(resample_table(orig_freq, new_freq, &t));
  const unsigned grid = (unsigned)((*n_out + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (in_dtype == XT_I16) hipLaunchKernelGGL(resample_kernel<int16_t>, dim3(grid), dim3(256), 0, st, (const int16_t*)d_in, (long)n_in, t.d, t.O, t.N, t.width, t.K, d_out, (long)*n_out);
  else hipLaunchKernelGGL(resample_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)d_in, (long)n_in, t.d, t.O, t.N, t.width, t.K, d_out, (long)*n_out);
  SK_HIP(hipGetLastError());
  return SK_OK;
}
