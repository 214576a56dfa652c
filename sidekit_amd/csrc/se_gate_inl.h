// The squeeze-excite gate of one utterance, computed by ONE workgroup of NT threads with the arithmetic of se_pre_kernel (se_gate.hip):
// the same sums in the same order, so the gate is the same bits whichever kernel produced it.
//
// Why it exists (round 5): the reference driver calls the model one utterance at a time (sidekit/bin/extract_xvectors.py:130-150); at batch 1
// a forward is a chain of 60 dependent kernels of which sixteen are SE gates -- 5.7-7.3 us each for layers 1-3 (the floor of a dependent
// launch) and 16 us for layer 4, 135 us of a 0.67-ms forward (profiles/r05_b1_kernel_stats_before.csv).  The gate is needed only by conv2's
// EPILOGUE, and the kernel boundary between conv1 and conv2 already makes conv1's sums visible.  So for small batches conv2 itself -- a
// separate template instantiation, the batch-256 kernels keep their registers -- lets every workgroup of an utterance compute that
// utterance's gate while its halo tile is landing in LDS: no launch, no fences, no tickets (the round-4 construction, conv1's last workgroup
// behind agent-scope releases, lost for exactly those).  The work is redundant across the utterance's workgroups (9 C^2 MACs and as many
// weight bytes from L2 each), which is why it is selected for small grids only (launch_cfg in conv3x3.hip).
//
// se_pre_kernel runs 1024 threads; here NT (256) threads each walk 1024 / NT "virtual threads" v = tid + j NT of that kernel.  Every
// reduction keeps se_pre_kernel's partition (which virtual thread owns which partial sum) and its order (partials are combined in index
// order through the same LDS layout), every FMA chain its k order.  `red` is 8192 floats, `S` 2304, `y` 256, `hid` 16; `gate_out` receives
// C floats (LDS).  Ends with the gate visible to the whole workgroup (__syncthreads).
#pragma once
#include "kernels.h"

namespace sk {

template <typename WT, int C, int NT>
__device__ __attribute__((always_inline)) inline void se_gate_block(const SeArgs a, int b, int tid, float* __restrict__ red, float* __restrict__ S, float* __restrict__ y,
                                     float* __restrict__ hid, float* __restrict__ gate_out) {
  static_assert(1024 % NT == 0 && NT >= C && NT >= 256, "virtual-thread walk: NT divides 1024 and covers the per-channel stages");
  constexpr int NV = 1024 / NT;
  constexpr int VEC = 16 / sizeof(WT), G = 1024 / C, R = C / 16;
  const int hb = halve(a.lens.get_uniform(b), a.halvings);
  const int nt = (hb + a.th - 1) / a.th;
  // ---- phase 1: conv1's per-tile sums -> S[9][C] (sums of the zero-padded plane shifted by each tap)
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int v = tid + j * NT, c = v % C, g = v / C;
    float T = 0.f, C0 = 0.f, CL = 0.f;
    for (int t = g; t < nt; t += G) {
      for (int wv = 0; wv < a.wm; ++wv) T += a.se_part[(((size_t)b * a.tiles + t) * a.wm + wv) * C + c];
      C0 += a.col_part[((size_t)b * a.tiles + t) * 2 * C + c];
      CL += a.col_part[((size_t)b * a.tiles + t) * 2 * C + C + c];
    }
    red[v] = T; red[1024 + v] = C0; red[2048 + v] = CL;
  }
  __syncthreads();
  if (tid < C) {
    const int c = tid;
    float T = 0.f, C0 = 0.f, CL = 0.f;
    for (int q = 0; q < G; ++q) { T += red[q * C + c]; C0 += red[1024 + q * C + c]; CL += red[2048 + q * C + c]; }
    const float* eg = a.edge + (size_t)b * 6 * C + c;
    const float R0 = eg[0], RL = eg[C], k00 = eg[2 * C], k0L = eg[3 * C], kL0 = eg[4 * C], kLL = eg[5 * C];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float rex = kh == 0 ? RL : (kh == 2 ? R0 : 0.f);
        const float cex = kw == 0 ? CL : (kw == 2 ? C0 : 0.f);
        const float corner = (kh == 0 && kw == 0) ? kLL : (kh == 0 && kw == 2) ? kL0 : (kh == 2 && kw == 0) ? k0L : (kh == 2 && kw == 2) ? k00 : 0.f;
        S[(kh * 3 + kw) * C + c] = T - rex - cex + corner;
      }
  }
  __syncthreads();
  // ---- phase 2: S contracted with conv2's weights; virtual thread (cg, kg) owns 16 B of output channels and the rows k = kg + i KG
  constexpr int CG = C / VEC, KG = 1024 / CG, NIT = (9 * C + KG - 1) / KG;
  constexpr int CH = NIT % 12 == 0 ? 12 : (NIT % 9 == 0 ? 9 : (NIT % 8 == 0 ? 8 : (NIT % 6 == 0 ? 6 : (NIT % 4 == 0 ? 4 : (NIT % 3 == 0 ? 3 : (NIT % 2 == 0 ? 2 : 1))))));   // loads in flight per round
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.w2t);
#pragma unroll 1
  for (int j = 0; j < NV; ++j) {
    const int v = tid + j * NT, cg = v % CG, kg = v / CG;
    const unsigned char* wp = wbase + (size_t)cg * 16;
    float m[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) m[q] = 0.f;
#pragma unroll 1
    for (int i0 = 0; i0 < NIT; i0 += CH) {
      uint4 w[CH];
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int k = kg + (i0 + i) * KG;
        w[i] = k < 9 * C ? *reinterpret_cast<const uint4*>(wp + (size_t)k * C * sizeof(WT)) : make_uint4(0, 0, 0, 0);
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
    for (int q = 0; q < VEC; ++q) red[kg * C + cg * VEC + q] = m[q];
  }
  __syncthreads();
  {  // the KG row-group partials of a channel meet in two stages, as in se_pre_kernel
    float part[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int v = tid + j * NT, c = v % C, g = v / C;
      float t = 0.f;
      for (int q = g; q < KG; q += G) t += red[q * C + c];
      part[j] = t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) red[tid + j * NT] = part[j];     // red[g * C + c] with v = g * C + c
    __syncthreads();
    if (tid < C) {
      float t = 0.f;
      for (int q = 0; q < G; ++q) t += red[q * C + tid];
      y[tid] = t / (float)(hb * a.wout) * a.scale2[tid] + a.shift2[tid];
    }
  }
  __syncthreads();
  // ---- phase 3: FC -> ReLU -> FC -> sigmoid.  fc1 [R][C] at red[0 ..), fc2 [C][R] at red[4096 ..)
  constexpr int NF = C * R;
  for (int idx = tid; idx < NF; idx += NT) { red[idx] = a.fc1[idx]; red[4096 + idx] = a.fc2[idx]; }
  __syncthreads();
  {
    constexpr int NS = C / 16;                       // R * NS = C * C / 256 <= 256 virtual threads: all of them real
    const int r1 = tid % R, sl = tid / R;
    if (sl < NS) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s = fmaf(red[r1 * C + sl * 16 + k], y[sl * 16 + k], s);
      S[sl * R + r1] = s;
    }
    __syncthreads();
    if (tid < R) {
      float s = 0.f;
      for (int q = 0; q < NS; ++q) s += S[q * R + tid];
      hid[tid] = relu_nan(s);
    }
  }
  __syncthreads();
  if (tid < C) {
    float z = 0.f;
    for (int k = 0; k < R; ++k) z = fmaf(red[4096 + tid * R + k], hid[k], z);
    gate_out[tid] = 1.f / (1.f + expf(-z));
  }
  __syncthreads();
}

constexpr int SE_GATE_SCRATCH_FLOATS = 8192 + 9 * 256 + 256 + 16;   // red, S, y, hid

}  // namespace sk
