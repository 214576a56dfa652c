// Small trunk kernels around the 3x3 convolutions: stem conv (1->32) and the squeeze-excite gate.
// Reference: sidekit/nnet/res_net.py:272-281 (SELayer), :509-515,549 (stem).  The block tail
// (gate * out + shortcut, ReLU; res_net.py:316-319) lives in the second convolution's epilogue.
#include "kernels.h"

namespace sk {

// ---- stem: relu(bn(conv3x3(1->32, pad 1))) on the logical (B,1,H=T,W=80) image -----------------
// One workgroup = TT time rows x 80 freqs; the (TT+2) x 82 input patch goes through LDS, every
// thread produces one position x 32 channels (288 FMAs as 144 packed-f32 FMAs, weights as SGPR operands) and writes
// 64 B (bf16) / 128 B (f32).
constexpr int STEM_TT = 16;
constexpr int STEM_W = 80;

template <int EB>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ feats, long sb, long sf, long st,
                                                   const float* __restrict__ w, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, unsigned char* __restrict__ out,
                                                   Lens lens, int T) {
  __shared__ float patch[(STEM_TT + 2) * (STEM_W + 2)];
  __shared__ __attribute__((aligned(16))) unsigned char stage[256 * (32 * EB + 16)];
  const int tid = threadIdx.x;
  const int tiles = (T + STEM_TT - 1) / STEM_TT;
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * STEM_TT;
  const int tb = lens.get(b);
  if (t0 >= tb) return;
  for (int i = tid; i < (STEM_TT + 2) * (STEM_W + 2); i += 256) {
    const int row = i / (STEM_W + 2), col = i % (STEM_W + 2);
    const int t = t0 - 1 + row, f = col - 1;
    float v = 0.f;
    if (t >= 0 && t < tb && f >= 0 && f < STEM_W) v = feats[b * sb + f * sf + t * st];
    patch[i] = v;
  }
  __syncthreads();
  // 1280 positions = 5 rounds of 256; a round's outputs are 256 consecutive NHWC positions = one contiguous 16 / 32 KB
  // run, staged through LDS so that every store instruction writes whole 1-KB lines (a thread storing its own 64 B at a
  // 64-B lane stride ran at 2.9 TB/s)
  constexpr int PB = 32 * EB, PS = PB + 16;   // bytes per position, padded LDS stride
  for (int p0 = 0; p0 < STEM_TT * STEM_W; p0 += 256) {
    const int p = p0 + tid;
    const int tl = p / STEM_W, f = p % STEM_W;
    const int t = t0 + tl;
    if (t < tb) {
      float x[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) x[q] = patch[(tl + q / 3) * (STEM_W + 2) + f + q % 3];
      unsigned char* lp = stage + tid * PS;
#pragma unroll
      for (int c0 = 0; c0 < 32; c0 += 8) {
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; c += 2) {   // two channels per v_pk_fma_f32; w is tap-major [9][32], uniform index -> scalar loads
          f32x2_t s = {0.f, 0.f};
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const f32x2_t wq = {w[q * 32 + c0 + c], w[q * 32 + c0 + c + 1]}, xq = {x[q], x[q]};
            s = __builtin_elementwise_fma(wq, xq, s);
          }
          v[c] = relu_nan(s[0] * scale[c0 + c] + shift[c0 + c]);
          v[c + 1] = relu_nan(s[1] * scale[c0 + c + 1] + shift[c0 + c + 1]);
        }
        if constexpr (EB == 2) {
          *reinterpret_cast<uint4*>(lp + c0 * 2) =
              make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        } else {
          *reinterpret_cast<float4*>(lp + c0 * 4) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(lp + c0 * 4 + 16) = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
    }
    __syncthreads();
    // rows are whole: the round's valid positions are a prefix (positions past the utterance's last row are not written)
    const int nvalid = (tb - t0) * STEM_W - p0;   // valid positions of this round (may exceed 256)
    unsigned char* ob = out + (((size_t)b * T + t0) * STEM_W + p0) * PB;
    constexpr int CPP = PB / 16;                  // 16-B chunks per position
#pragma unroll
    for (int j = 0; j < CPP; ++j) {
      const int id = j * 256 + tid, pos = id / CPP, part = id % CPP;
      if (pos < nvalid) *reinterpret_cast<uint4*>(ob + (size_t)id * 16) = *reinterpret_cast<const uint4*>(stage + pos * PS + part * 16);
    }
    __syncthreads();
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

// ---- SE gate from conv1's output sums -----------------------------------------------------------------
// One workgroup (1024 threads) per utterance.  Partial sums are added in a fixed order (tile, wave) so the result is
// bitwise reproducible.  For the tap shifted by (dh, dw) the sum of the shifted, zero-padded plane is
//   S = T - R(excluded border row) - C(excluded border column) + corner(both excluded).
// Phase 1 (thread = channel x tile group) reduces conv1's per-tile sums to S[9][C]; phase 2 contracts S with conv2's
// weights: a thread owns 16 B of consecutive output channels (8 bf16 / 4 f32) and every KG-th (tap, ci) row, so the
// 9*C*C weights stream as whole 16-B loads (one 2/4-B load per FMA was latency-bound: 55 us at C = 256), partial
// sums meet in LDS in row-group order; phase 3 is the two small FC layers and the sigmoid.
template <typename WT>
__global__ __launch_bounds__(1024) void se_pre_kernel(SeArgs a) {
  constexpr int VEC = 16 / sizeof(WT);
  __shared__ float red[8 * 1024];   // phase 1: 3 x 1024; phase 2: [KG][C] partial sums (KG * C = 1024 * VEC / ... <= 8192)
  __shared__ float S[9 * 256];
  __shared__ float y[256];
  __shared__ float hid[16];
  const int b = blockIdx.x, C = a.C, G = 1024 / C, c = threadIdx.x % C, g = threadIdx.x / C;
  const int hb = halve(a.lens.get(b), a.halvings);
  const int nt = (hb + a.th - 1) / a.th;
  float T = 0.f, C0 = 0.f, CL = 0.f;
  for (int t = g; t < nt; t += G) {
    for (int w = 0; w < a.wm; ++w) T += a.se_part[(((size_t)b * a.tiles + t) * a.wm + w) * C + c];
    C0 += a.col_part[((size_t)b * a.tiles + t) * 2 * C + c];
    CL += a.col_part[((size_t)b * a.tiles + t) * 2 * C + C + c];
  }
  red[threadIdx.x] = T; red[1024 + threadIdx.x] = C0; red[2048 + threadIdx.x] = CL;
  __syncthreads();
  if (g == 0) {
    T = 0.f; C0 = 0.f; CL = 0.f;
    for (int q = 0; q < G; ++q) { T += red[q * C + c]; C0 += red[1024 + q * C + c]; CL += red[2048 + q * C + c]; }
    const float* eg = a.edge + (size_t)b * 6 * C + c;
    const float R0 = eg[0], RL = eg[C], k00 = eg[2 * C], k0L = eg[3 * C], kL0 = eg[4 * C], kLL = eg[5 * C];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        // dh = kh - 1: a tap with dh < 0 never reaches the last row, dh > 0 never the first; same for columns
        const float rex = kh == 0 ? RL : (kh == 2 ? R0 : 0.f);
        const float cex = kw == 0 ? CL : (kw == 2 ? C0 : 0.f);
        const float corner = (kh == 0 && kw == 0) ? kLL : (kh == 0 && kw == 2) ? kL0 : (kh == 2 && kw == 0) ? k0L : (kh == 2 && kw == 2) ? k00 : 0.f;
        S[(kh * 3 + kw) * C + c] = T - rex - cex + corner;
      }
  }
  __syncthreads();
  {
    const int CG = C / VEC, KG = 1024 / CG, cg = threadIdx.x % CG, kg = threadIdx.x / CG;
    float m[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) m[v] = 0.f;
    const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.w2t) + (size_t)cg * 16;
#pragma unroll 8
    for (int k = kg; k < 9 * C; k += KG) {   // k = tap * C + ci; eight 16-B weight loads in flight per thread (the loop is L2-latency bound)
      const uint4 w = *reinterpret_cast<const uint4*>(wp + (size_t)k * C * sizeof(WT));
      const float s = S[k];
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if constexpr (sizeof(WT) == 2) {
          m[2 * q] = fmaf(bf16_to_f32((uint16_t)(ww[q] & 0xffff)), s, m[2 * q]);
          m[2 * q + 1] = fmaf(bf16_to_f32((uint16_t)(ww[q] >> 16)), s, m[2 * q + 1]);
        } else {
          m[q] = fmaf(__builtin_bit_cast(float, ww[q]), s, m[q]);
        }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[kg * C + cg * VEC + v] = m[v];   // KG * C = 1024 * VEC floats
    __syncthreads();
    // the KG row-group partials of a channel meet in two stages (every thread sums KG / G of them, then G partial sums):
    // a single thread walking all KG (256 at C = 32) was a serial chain of LDS reads; the order stays fixed
    {
      float t = 0.f;
      for (int q = g; q < KG; q += G) t += red[q * C + c];
      __syncthreads();
      red[g * C + c] = t;
    }
    __syncthreads();
    if (g == 0) {
      float t = 0.f;
      for (int q = 0; q < G; ++q) t += red[q * C + c];
      y[c] = t / (float)(hb * a.wout) * a.scale2[c] + a.shift2[c];
    }
  }
  __syncthreads();
  const int R = C / 16;
  {  // FC1 (R x C): all threads, thread = (hidden unit r, slice of 16 input channels), slices added in order
    const int NS = C / 16, r1 = threadIdx.x % R, sl = threadIdx.x / R;   // R * NS = C * C / 256 <= 256 threads
    if (sl < NS) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s = fmaf(a.fc1[r1 * C + sl * 16 + k], y[sl * 16 + k], s);
      red[sl * R + r1] = s;
    }
    __syncthreads();
    if (threadIdx.x < R) {
      float s = 0.f;
      for (int q = 0; q < NS; ++q) s += red[q * R + threadIdx.x];
      hid[threadIdx.x] = relu_nan(s);
    }
  }
  __syncthreads();
  if (g == 0) {
    float z = 0.f;
    for (int k = 0; k < R; ++k) z = fmaf(a.fc2[c * R + k], hid[k], z);
    a.gate[(size_t)b * C + c] = 1.f / (1.f + expf(-z));
  }
}

int launch_se_pre(const SeArgs& a, hipStream_t s) {
  SK_CHECK(a.C <= 256 && a.C % 16 == 0 && 1024 % a.C == 0, SK_EARG, "se_pre: C=%d unsupported", a.C);
  if (a.w2t_bf16) hipLaunchKernelGGL(se_pre_kernel<uint16_t>, dim3(a.B), dim3(1024), 0, s, a);
  else hipLaunchKernelGGL(se_pre_kernel<float>, dim3(a.B), dim3(1024), 0, s, a);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
