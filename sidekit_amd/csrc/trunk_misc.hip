// Small trunk kernels around the 3x3 convolutions: stem conv (1->32), squeeze-excite gate,
// gate*x + shortcut + ReLU.  Reference: sidekit/nnet/res_net.py:272-281 (SELayer),
// :309-320 (BasicBlock tail), :509-515,549 (stem).  All HBM-bound: 16-B vector accesses.
#include "kernels.h"

namespace sk {

// ---- stem: relu(bn(conv3x3(1->32, pad 1))) on the logical (B,1,H=T,W=80) image -----------------
// One workgroup = TT time rows x 80 freqs; the (TT+2) x 82 input patch goes through LDS, every
// thread produces one position x 32 channels (288 FMAs) and writes 64 B (bf16) / 128 B (f32).
constexpr int STEM_TT = 16;
constexpr int STEM_W = 80;

template <int EB>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ feats, long sb, long sf, long st,
                                                   const float* __restrict__ w, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, unsigned char* __restrict__ out,
                                                   Lens lens, int T) {
  __shared__ float patch[(STEM_TT + 2) * (STEM_W + 2)];
  __shared__ float ws[32 * 9 + 64];
  const int tid = threadIdx.x;
  const int tiles = (T + STEM_TT - 1) / STEM_TT;
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * STEM_TT;
  const int tb = lens.get(b);
  if (t0 >= tb) return;
  for (int i = tid; i < 32 * 9; i += 256) ws[i] = w[i];
  if (tid < 32) { ws[288 + tid] = scale[tid]; ws[320 + tid] = shift[tid]; }
  for (int i = tid; i < (STEM_TT + 2) * (STEM_W + 2); i += 256) {
    const int row = i / (STEM_W + 2), col = i % (STEM_W + 2);
    const int t = t0 - 1 + row, f = col - 1;
    float v = 0.f;
    if (t >= 0 && t < tb && f >= 0 && f < STEM_W) v = feats[b * sb + f * sf + t * st];
    patch[i] = v;
  }
  __syncthreads();
  for (int p = tid; p < STEM_TT * STEM_W; p += 256) {
    const int tl = p / STEM_W, f = p % STEM_W;
    const int t = t0 + tl;
    if (t >= tb) continue;
    float x[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) x[q] = patch[(tl + q / 3) * (STEM_W + 2) + f + q % 3];
    unsigned char* op = out + (((size_t)b * T + t) * STEM_W + f) * 32 * EB;
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += 8) {
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 9; ++q) s = fmaf(ws[(c0 + c) * 9 + q], x[q], s);
        v[c] = relu_nan(s * ws[288 + c0 + c] + ws[320 + c0 + c]);
      }
      if constexpr (EB == 2) {
        *reinterpret_cast<uint4*>(op + c0 * 2) =
            make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
      } else {
        *reinterpret_cast<float4*>(op + c0 * 4) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + c0 * 4 + 16) = make_float4(v[4], v[5], v[6], v[7]);
      }
    }
  }
}

int launch_stem(const float* feats, long sb, long sf, long st, const float* w, const float* scale, const float* shift,
                void* out, int dtype, Lens lens, int B, int T, hipStream_t s) {
  const int tiles = cdiv(T, STEM_TT);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(stem_kernel<2>, dim3(B * tiles), dim3(256), 0, s, feats, sb, sf, st, w, scale, shift,
                       (unsigned char*)out, lens, T);
  else
    hipLaunchKernelGGL(stem_kernel<4>, dim3(B * tiles), dim3(256), 0, s, feats, sb, sf, st, w, scale, shift,
                       (unsigned char*)out, lens, T);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

// ---- SE gate -----------------------------------------------------------------------------------
// One workgroup per utterance, one thread per channel.  Partial plane sums are added in a fixed
// order (tile, wave) so the result is bitwise reproducible.
__global__ void se_gate_kernel(const float* __restrict__ se_part, int tiles, int wm, int th,
                               const float* __restrict__ w1, const float* __restrict__ w2, float* __restrict__ gate,
                               Lens lens, int halvings_out, int wout, int C) {
  __shared__ float y[256];
  __shared__ float hid[16];
  const int b = blockIdx.x, c = threadIdx.x;
  const int hb = halve(lens.get(b), halvings_out);
  const int nt = (hb + th - 1) / th;
  float s = 0.f;
  for (int t = 0; t < nt; ++t)
    for (int w = 0; w < wm; ++w) s += se_part[(((size_t)b * tiles + t) * wm + w) * C + c];
  y[c] = s / (float)(hb * wout);
  __syncthreads();
  const int R = C / 16;
  if (c < R) {
    float a = 0.f;
    for (int k = 0; k < C; ++k) a = fmaf(w1[c * C + k], y[k], a);
    hid[c] = relu_nan(a);
  }
  __syncthreads();
  float z = 0.f;
  for (int k = 0; k < R; ++k) z = fmaf(w2[c * R + k], hid[k], z);
  gate[(size_t)b * C + c] = 1.f / (1.f + expf(-z));
}

int launch_se_gate(const float* se_part, int tiles, int wm, int th, const float* w1, const float* w2, float* gate,
                   Lens lens, int halvings_out, int wout, int C, int B, hipStream_t s) {
  SK_CHECK(C <= 256 && C % 16 == 0, SK_EARG, "se_gate: C=%d unsupported", C);
  hipLaunchKernelGGL(se_gate_kernel, dim3(B), dim3(C), 0, s, se_part, tiles, wm, th, w1, w2, gate, lens, halvings_out,
                     wout, C);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

// ---- y = relu(o2 * gate + shortcut) ----------------------------------------------------------------
template <int EB>
__global__ __launch_bounds__(256) void residual_kernel(const uint4* __restrict__ o2, const float* __restrict__ gate,
                                                       const uint4* __restrict__ sc, uint4* __restrict__ y, long nvec,
                                                       long vec_per_utt, int C) {
  constexpr int VE = 16 / EB;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += (long)gridDim.x * 256L) {
    const int b = (int)(i / vec_per_utt);
    const int c = (int)((i * VE) % C);
    const float* g = gate + (size_t)b * C + c;
    const uint4 a = o2[i], s = sc[i];
    uint4 r;
    if constexpr (EB == 2) {
      const uint32_t av[4] = {a.x, a.y, a.z, a.w}, sv[4] = {s.x, s.y, s.z, s.w};
      uint32_t rv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float lo = relu_nan(bf16_to_f32(av[q] & 0xffff) * g[2 * q] + bf16_to_f32(sv[q] & 0xffff));
        const float hi = relu_nan(bf16_to_f32(av[q] >> 16) * g[2 * q + 1] + bf16_to_f32(sv[q] >> 16));
        rv[q] = pack_bf16x2(lo, hi);
      }
      r = make_uint4(rv[0], rv[1], rv[2], rv[3]);
    } else {
      const float4 af = __builtin_bit_cast(float4, a), sf = __builtin_bit_cast(float4, s);
      float4 rf;
      rf.x = relu_nan(af.x * g[0] + sf.x);
      rf.y = relu_nan(af.y * g[1] + sf.y);
      rf.z = relu_nan(af.z * g[2] + sf.z);
      rf.w = relu_nan(af.w * g[3] + sf.w);
      r = __builtin_bit_cast(uint4, rf);
    }
    y[i] = r;
  }
}

int launch_residual(const void* o2, const float* gate, const void* sc, void* y, int dtype, int B, long plane, int C,
                    hipStream_t s) {
  const int EB = dtype == DT_BF16 ? 2 : 4;
  const long vec_per_utt = plane * C * EB / 16;
  const long nvec = vec_per_utt * B;
  const int grid = (int)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(residual_kernel<2>, dim3(grid), dim3(256), 0, s, (const uint4*)o2, gate, (const uint4*)sc,
                       (uint4*)y, nvec, vec_per_utt, C);
  else
    hipLaunchKernelGGL(residual_kernel<4>, dim3(grid), dim3(256), 0, s, (const uint4*)o2, gate, (const uint4*)sc,
                       (uint4*)y, nvec, vec_per_utt, C);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
