// Squeeze-excite gate of a BasicBlock, computed BEFORE the block's second convolution runs.  Reference: sidekit/nnet/res_net.py:272-281
// (SELayer); the block tail (gate * out + shortcut, ReLU; res_net.py:316-319) lives in the second convolution's epilogue.
// Built like every other file with -fno-slp-vectorize since round 5: rounds 3-4 built this file alone with SLP for its packed FMAs (13.7 vs
// 18.5 us per launch in a serial forward); inside the pipelined step the difference does not show (profiles/r05_ab_scalar_forms.txt).
#include "kernels.h"

namespace sk {

// ---- SE gate from conv1's output sums -----------------------------------------------------------------
// One workgroup (1024 threads) per utterance.  Partial sums are added in a fixed order (tile, wave) so the result is
// bitwise reproducible.  For the tap shifted by (dh, dw) the sum of the shifted, zero-padded plane is
//   S = T - R(excluded border row) - C(excluded border column) + corner(both excluded).
// Phase 1 (thread = channel x tile group) reduces conv1's per-tile sums to S[9][C]; phase 2 contracts S with conv2's
// weights: a thread owns 16 B of consecutive output channels (8 bf16 / 4 f32) and every KG-th (tap, ci) row, so the
// 9*C*C weights stream as whole 16-B loads (one 2/4-B load per FMA was latency-bound: 55 us at C = 256), partial
// sums meet in LDS in row-group order; phase 3 is the two small FC layers and the sigmoid.
constexpr int se_chunk(int nit) {
  for (int d = 24; d > 1; --d)
    if (nit % d == 0) return d;
  return 1;
}

template <typename WT, int C_>
__global__ __launch_bounds__(1024) void se_pre_kernel(SeArgs a) {
  constexpr int VEC = 16 / sizeof(WT);
  __shared__ float red[8 * 1024];   // phase 1: 3 x 1024; phase 2: [KG][C] partial sums (KG * C = 1024 * VEC / ... <= 8192); phase 3: the two FC weight matrices
  __shared__ float S[9 * 256];      // phase 3: FC1's slice partials
  __shared__ float y[256];
  __shared__ float hid[16];
  constexpr int C = C_, G = 1024 / C, R = C / 16;
  const int b = blockIdx.x, c = threadIdx.x % C, g = threadIdx.x / C;
  // (0) Everything that does not depend on this launch's sums is requested FIRST (round 4): the two FC weight matrices (C * C / 16 floats
  // each, parked in registers until phase 3 and read from LDS there), conv2's BatchNorm constants and the first CH conv2-weight rows of
  // phase 2.  The kernel is a chain of dependent L2 round trips (a batch-1 forward spends 146 us in sixteen of these launches for almost no
  // arithmetic); these four used to sit behind the phases that consume them.  Values and summation order are unchanged: same bits.
  constexpr int NF = C * R, NPF = (NF + 1023) / 1024;
  float pf1[NPF], pf2[NPF];
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int idx = threadIdx.x + i * 1024;
    pf1[i] = idx < NF ? a.fc1[idx] : 0.f;
    pf2[i] = idx < NF ? a.fc2[idx] : 0.f;
  }
  const float psc = a.scale2[c], psh = a.shift2[c];
  constexpr int CG = C / VEC, KG = 1024 / CG;
  const int cg = threadIdx.x % CG, kg = threadIdx.x / CG;
  const unsigned char* wp = reinterpret_cast<const unsigned char*>(a.w2t) + (size_t)cg * 16;
  // k = tap * C + ci, this thread's rows k = kg + i * KG.  The loop is a chain of L2 latencies (72 dependent rounds of one 16-B load at
  // C = 256 were 55 us, eight in flight 20 us): the trip count is a compile-time constant, so up to 24 loads are issued back to back
  // (three rounds at C = 256, one at C <= 128) and the FMAs follow in k order -- the sums are those of the rolled loop.
  constexpr int NIT = (9 * C + KG - 1) / KG, CH = se_chunk(NIT);   // the largest divisor of the trip count up to 24
  constexpr bool EARLY = C <= 128;   // C = 256: 24 rows in flight + the parked FC weights do not fit the 128 registers of a 1024-thread workgroup
  uint4 w[CH];
  if constexpr (EARLY) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int k = kg + i * KG;
      w[i] = k < 9 * C ? *reinterpret_cast<const uint4*>(wp + (size_t)k * C * sizeof(WT)) : make_uint4(0, 0, 0, 0);
    }
  }
  const int hb = halve(a.lens.get(b), a.halvings);
  const int nt = (hb + a.th - 1) / a.th;
  float T = 0.f, C0 = 0.f, CL = 0.f;
  for (int t = g; t < nt; t += G) {
    for (int wv = 0; wv < a.wm; ++wv) T += a.se_part[(((size_t)b * a.tiles + t) * a.wm + wv) * C + c];
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
    float m[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) m[v] = 0.f;
#pragma unroll 1
    for (int i0 = 0; i0 < NIT; i0 += CH) {
      if (i0 || !EARLY) {   // the first chunk was requested at the top of the kernel (C <= 128)
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          const int k = kg + (i0 + i) * KG;
          w[i] = k < 9 * C ? *reinterpret_cast<const uint4*>(wp + (size_t)k * C * sizeof(WT)) : make_uint4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int k = kg + (i0 + i) * KG;
        if (k < 9 * C) {
          const float s = S[k];
          const uint32_t ww[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
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
      y[c] = t / (float)(hb * a.wout) * psc + psh;
    }
  }
  __syncthreads();
  // phase 3: the FC weights parked in registers go to LDS (red is free now): fc1 [R][C] at red[0 ..), fc2 [C][R] at red[4096 ..)
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int idx = threadIdx.x + i * 1024;
    if (idx < NF) { red[idx] = pf1[i]; red[4096 + idx] = pf2[i]; }
  }
  __syncthreads();
  {  // FC1 (R x C): all threads, thread = (hidden unit r, slice of 16 input channels), slices added in order
    const int NS = C / 16, r1 = threadIdx.x % R, sl = threadIdx.x / R;   // R * NS = C * C / 256 <= 256 threads
    if (sl < NS) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s = fmaf(red[r1 * C + sl * 16 + k], y[sl * 16 + k], s);
      S[sl * R + r1] = s;
    }
    __syncthreads();
    if (threadIdx.x < R) {
      float s = 0.f;
      for (int q = 0; q < NS; ++q) s += S[q * R + threadIdx.x];
      hid[threadIdx.x] = relu_nan(s);
    }
  }
  __syncthreads();
  if (g == 0) {
    float z = 0.f;
    for (int k = 0; k < R; ++k) z = fmaf(red[4096 + c * R + k], hid[k], z);
    a.gate[(size_t)b * C + c] = 1.f / (1.f + expf(-z));
  }
}

int launch_se_pre(const SeArgs& a, hipStream_t s) {
  SK_CHECK(a.C == 32 || a.C == 64 || a.C == 128 || a.C == 256, SK_EARG, "se_pre: C=%d unsupported (32, 64, 128, 256)", a.C);
#define SK_SE_LAUNCH(CC) do { \
    if (a.w2t_bf16) hipLaunchKernelGGL((se_pre_kernel<uint16_t, CC>), dim3(a.B), dim3(1024), 0, s, a); \
    else hipLaunchKernelGGL((se_pre_kernel<float, CC>), dim3(a.B), dim3(1024), 0, s, a); } while (0)
  switch (a.C) {
    case 32: SK_SE_LAUNCH(32); break;
    case 64: SK_SE_LAUNCH(64); break;
    case 128: SK_SE_LAUNCH(128); break;
    default: SK_SE_LAUNCH(256); break;
  }
#undef SK_SE_LAUNCH
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
