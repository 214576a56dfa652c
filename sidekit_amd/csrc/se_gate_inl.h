// The squeeze-excite gate of one utterance, computed by ONE workgroup of NT threads with the arithmetic of se_pre_kernel (se_gate.hip):
// the same sums in the same order, so the gate is the same bits whichever kernel produced it.
//
// Why it exists (round 5): the reference driver calls the model one utterance at a time (sidekit/bin/extract_xvectors.py:130-150); at batch 1
// a forward is a chain of 60 dependent kernels of which sixteen are SE gates -- 5.7-7.3 us each for layers 1-3 and 16 us for layer 4, 135 us
// of a 0.67-ms forward (profiles/r05_b1_kernel_stats_before.csv).  The gate is needed only by conv2's EPILOGUE, and the kernel boundary between
// conv1 and conv2 already makes conv1's sums visible.  The round-4 verdict asked for the gate in conv2's prologue -- a separate template
// instantiation (the batch-256 kernels keep their registers) in which every workgroup of an utterance computes that utterance's gate itself:
// no launch, no fences, no tickets (round 4's construction, conv1's last workgroup behind agent-scope releases, lost for exactly those).
// Built, bit-identical, and MEASURED: it does not pay either.  The gate is a chain of dependent L2 round trips (conv1's sums come from other
// XCDs' write-backs; 9 C^2 weight bytes pass one CU's L2 port) and barriers that takes 5-6 us whether it is a kernel or the head of one; on one
// wave beside the k-loop (se_gate_wave below) it takes ~10 us, longer than the convolution it would hide behind (xt_api.hip, xt_handle::
// gate_prologue; profiles/r05_latency_matrix.txt).  The forms stay selectable for A/B and are tested for bit-identity; the product launches
// se_pre_kernel.
//
// se_pre_kernel runs 1024 threads; here NT (256) threads each walk 1024 / NT "virtual threads" v = tid + j NT of that kernel.  Every
// reduction keeps se_pre_kernel's partition (which virtual thread owns which partial sum) and its order (partials are combined in index
// order through the same LDS layout), every FMA chain its k order.  `red` is 8192 floats, `S` 2304, `y` 256, `hid` 16; `gate_out` receives
// C floats (LDS).  Ends with the gate visible to the whole workgroup (__syncthreads).
#pragma once
#include "kernels.h"

namespace sk {

constexpr int se_rows_per_round(int nit) {   // the largest divisor of the trip count up to 24
  for (int d = 24; d > 1; --d)
    if (nit % d == 0) return d;
  return 1;
}

template <typename WT, int C, int NT>
__device__ __attribute__((always_inline)) inline void se_gate_block(const SeArgs a, int b, int tid, float* __restrict__ red, float* __restrict__ S, float* __restrict__ y,
                                     float* __restrict__ hid, float* __restrict__ gate_out) {
  static_assert(1024 % NT == 0 && NT >= C && NT >= 256, "virtual-thread walk: NT divides 1024 and covers the per-channel stages");
  constexpr int NV = 1024 / NT;
  constexpr int VEC = 16 / sizeof(WT), G = 1024 / C, R = C / 16;
  constexpr int CG = C / VEC, KG = 1024 / CG, NIT = (9 * C + KG - 1) / KG;
  // This runs on the critical path of a batch-1 forward with ONE wave per SIMD: it is written for few dependent round trips to L2, not for
  // throughput.  Everything that does not depend on conv1's sums is requested first -- the FC matrices, conv2's BatchNorm constants and as
  // many conv2-weight rows as the register file holds (the kernel owns all of it: one workgroup per CU) --, the sums of the NV virtual
  // threads are requested together, and the weight rows travel in rounds of WCH loads.
  constexpr int WCH = se_rows_per_round(NIT);        // rows per round and virtual thread: bf16 C <= 128 one round (2 / 5 / 18 rows), C = 256 three of 24
  static_assert(NIT % WCH == 0, "weight rounds");
  constexpr int NF = C * R, NPF = (NF + NT - 1) / NT;
  constexpr bool FC_EARLY = NPF <= 4;                // C = 256 (16 + 16 registers beside two 96-register weight buffers) fetches them when phase 3 starts
  float pf1[NPF], pf2[NPF];
  if constexpr (FC_EARLY) {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int idx = tid + i * NT;
      pf1[i] = idx < NF ? a.fc1[idx] : 0.f;
      pf2[i] = idx < NF ? a.fc2[idx] : 0.f;
    }
  }
  const float psc = a.scale2[tid < C ? tid : 0], psh = a.shift2[tid < C ? tid : 0];
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.w2t);
  constexpr bool EXACT = KG * NIT == 9 * C;          // every (virtual thread, row) is a row of the matrix: no bound check
  auto wrow = [&](int j, int i) {                    // row i of virtual thread j: k = kg + i KG, this thread's 16 B of output channels
    const int v = tid + j * NT, cg = v % CG, kg = v / CG, k = kg + i * KG;
    // wave-uniform part (row block i) + 32-bit lane part: one address register per virtual thread instead of a 64-bit pair per row
    const unsigned lane_off = (unsigned)kg * (unsigned)(C * sizeof(WT)) + (unsigned)cg * 16u;
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(i * KG * C * (int)sizeof(WT));
    const uint4 val = *reinterpret_cast<const uint4*>(wbase + soff + ((EXACT || k < 9 * C) ? lane_off : 0u));
    return (EXACT || k < 9 * C) ? val : make_uint4(0, 0, 0, 0);
  };
  constexpr bool EARLY = NV * NIT <= 72 && WCH == NIT;             // every weight row of this thread in flight before the sums are (C <= 128: 8 / 20 / 72 x 16 B)
  uint4 w[EARLY ? NV : 1][WCH];
  if constexpr (EARLY) {
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
      for (int i = 0; i < WCH; ++i) w[j][i] = wrow(j, i);
  }
  const int hb = halve(a.lens.get_uniform(b), a.halvings);
  const int nt = (hb + a.th - 1) / a.th;
  // ---- phase 1: conv1's per-tile sums -> S[9][C] (sums of the zero-padded plane shifted by each tap).  Virtual thread (c, g) adds the tiles
  // t = g, g + G, ... in that order; the NV virtual threads of a lane advance together (a tile past the utterance's last adds +0.0f: the sums
  // start at +0.0f, so that changes no bit)
  {
    float T[NV], C0[NV], CL[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) T[j] = C0[j] = CL[j] = 0.f;
    for (int t0 = 0; t0 < nt; t0 += G) {
      float vT[NV][4], vC0[NV], vCL[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int v = tid + j * NT, c = v % C, t = t0 + v / C;
        const bool ok = t < nt;
        const size_t tt = (size_t)b * a.tiles + (ok ? t : 0);
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) vT[j][wv] = (ok && wv < a.wm) ? a.se_part[(tt * a.wm + (wv < a.wm ? wv : 0)) * C + c] : 0.f;
        vC0[j] = ok ? a.col_part[tt * 2 * C + c] : 0.f;
        vCL[j] = ok ? a.col_part[tt * 2 * C + C + c] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) T[j] += vT[j][wv];      // wm <= 4 wave rows; the rows past wm add +0.0f
        C0[j] += vC0[j];
        CL[j] += vCL[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) { const int v = tid + j * NT; red[v] = T[j]; red[1024 + v] = C0[j]; red[2048 + v] = CL[j]; }
  }
  float eg6[6];
  if (tid < C) {
    const float* eg = a.edge + (size_t)b * 6 * C + tid;
#pragma unroll
    for (int q = 0; q < 6; ++q) eg6[q] = eg[q * C];
  }
  __syncthreads();
  if (tid < C) {
    const int c = tid;
    float T = 0.f, C0 = 0.f, CL = 0.f;
    for (int q = 0; q < G; ++q) { T += red[q * C + c]; C0 += red[1024 + q * C + c]; CL += red[2048 + q * C + c]; }
    const float R0 = eg6[0], RL = eg6[1], k00 = eg6[2], k0L = eg6[3], kL0 = eg6[4], kLL = eg6[5];
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
  // ---- phase 2: S contracted with conv2's weights; virtual thread (cg, kg) owns 16 B of output channels and the rows k = kg + i KG, i ascending
  auto fma_row = [&](float* m, const uint4& wv, int k) {
    if (k < 9 * C) {
      const float s = S[k];
      const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
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
  };
  if constexpr (EARLY) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int v = tid + j * NT, cg = v % CG, kg = v / CG;
      float m[VEC];
#pragma unroll
      for (int q = 0; q < VEC; ++q) m[q] = 0.f;
#pragma unroll
      for (int i = 0; i < WCH; ++i) fma_row(m, w[j][i], kg + i * KG);
#pragma unroll
      for (int q = 0; q < VEC; ++q) red[kg * C + cg * VEC + q] = m[q];
    }
  } else {
    // C = 256: 288 rows of 16 B per thread -- rounds of WCH rows through two register buffers, the next round requested before this one is
    // consumed; rolled (two rounds per trip), the round index decides which virtual thread and which rows
    constexpr int RPJ = NIT / WCH, NRND = NV * RPJ;
    static_assert(NRND % 2 == 0, "two rounds per trip");
    uint4 wa[WCH], wb[WCH];
    float m[VEC];
    auto request = [&](uint4* buf, int rnd) {
      const int j = rnd / RPJ, i0 = (rnd % RPJ) * WCH;
#pragma unroll
      for (int i = 0; i < WCH; ++i) buf[i] = wrow(j, i0 + i);
    };
    auto consume = [&](const uint4* buf, int rnd) {
      const int j = rnd / RPJ, i0 = (rnd % RPJ) * WCH;
      const int v = tid + j * NT, cg = v % CG, kg = v / CG;
      if (i0 == 0) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) m[q] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < WCH; ++i) fma_row(m, buf[i], kg + (i0 + i) * KG);
      if (i0 + WCH == NIT) {
#pragma unroll
        for (int q = 0; q < VEC; ++q) red[kg * C + cg * VEC + q] = m[q];
      }
    };
    request(wa, 0);
#pragma unroll 1
    for (int rnd = 0; rnd < NRND; rnd += 2) {
      request(wb, rnd + 1);
      consume(wa, rnd);
      if (rnd + 2 < NRND) request(wa, rnd + 2);
      consume(wb, rnd + 1);
    }
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
      y[tid] = t / (float)(hb * a.wout) * psc + psh;
    }
  }
  __syncthreads();
  // ---- phase 3: FC -> ReLU -> FC -> sigmoid.  fc1 [R][C] at red[0 ..), fc2 [C][R] at red[4096 ..)
#pragma unroll
  for (int i = 0; i < NPF; ++i) {
    const int idx = tid + i * NT;
    if constexpr (!FC_EARLY) { pf1[i] = idx < NF ? a.fc1[idx] : 0.f; pf2[i] = idx < NF ? a.fc2[idx] : 0.f; }
    if (idx < NF) { red[idx] = pf1[i]; red[4096 + idx] = pf2[i]; }
  }
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

// The same gate on ONE wave (64 lanes walking 16 virtual threads each), for a wave that has nothing else to do: conv2's gate wave (conv3x3.hip,
// GATEPRO == 2) computes it BESIDE the four waves that stage the tile and run the k-loop -- the chain of dependent L2 round trips that makes
// the gate cost 5-6 us wherever it runs alone (profiles/r05_latency_matrix.txt) then costs nothing, because nobody waits for it before the
// epilogue.  A single wave needs no barrier: LDS operations of one wave execute in order, so a write is visible to the wave's next read once
// the compiler is kept from reordering them (wave_barrier) and the data has been returned (lgkmcnt).  Only C = 32 / 64 (layers 1-2): one
// wave's VALU is enough for 9 C^2 = 9 216 / 36 864 MACs, not for layer 3's 147 456.
__device__ inline void se_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <typename WT, int C>
__device__ __attribute__((always_inline)) inline void se_gate_wave(const SeArgs a, int b, int lane, float* __restrict__ red, float* __restrict__ S, float* __restrict__ y,
                                    float* __restrict__ hid, float* __restrict__ gate_out) {
  static_assert(C == 32 || C == 64, "one wave: layers 1-2");
  constexpr int NT = 64, NV = 1024 / NT;
  constexpr int VEC = 16 / sizeof(WT), G = 1024 / C, R = C / 16;
  constexpr int CG = C / VEC, KG = 1024 / CG, NIT = (9 * C + KG - 1) / KG;
  constexpr int NROW = NV * NIT;                     // weight rows of this lane, virtual-thread-major: bf16 32 (C = 32) / 80 (C = 64)
  constexpr int CHK = NROW % 20 == 0 ? 20 : 16;      // loads in flight: five waves on four SIMDs leave a wave 256 registers, and phase 1 holds 144 of them
  static_assert(NROW % CHK == 0, "weight chunks");
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.w2t);
  auto wrow = [&](int r) {                           // row r = (virtual thread j, trip i): k = kg + i KG, 16 B of output channels
    const int j = r / NIT, i = r % NIT;
    const int v = lane + j * NT, cg = v % CG, k = v / CG + i * KG;
    const unsigned off = (unsigned)k * (unsigned)(C * sizeof(WT)) + (unsigned)cg * 16u;
    const uint4 val = *reinterpret_cast<const uint4*>(wbase + (k < 9 * C ? off : 0u));
    return k < 9 * C ? val : make_uint4(0, 0, 0, 0);
  };
  uint4 w[CHK];
#pragma unroll
  for (int u = 0; u < CHK; ++u) w[u] = wrow(u);       // the first chunk travels while the sums are reduced
  const int hb = halve(a.lens.get_uniform(b), a.halvings);
  const int nt = (hb + a.th - 1) / a.th;
#pragma unroll
  for (int jh = 0; jh < NV; jh += NV / 2) {           // the virtual threads in two halves: 8 x 9 registers of sums and operands at a time
    constexpr int NH = NV / 2;
    float T[NH], C0[NH], CL[NH];
#pragma unroll
    for (int j = 0; j < NH; ++j) T[j] = C0[j] = CL[j] = 0.f;
    for (int t0 = 0; t0 < nt; t0 += G) {
      float vT[NH][4], vC0[NH], vCL[NH];
#pragma unroll
      for (int j = 0; j < NH; ++j) {
        const int v = lane + (jh + j) * NT, c = v % C, t = t0 + v / C;
        const bool ok = t < nt;
        const size_t tt = (size_t)b * a.tiles + (ok ? t : 0);
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) vT[j][wv] = (ok && wv < a.wm) ? a.se_part[(tt * a.wm + (wv < a.wm ? wv : 0)) * C + c] : 0.f;
        vC0[j] = ok ? a.col_part[tt * 2 * C + c] : 0.f;
        vCL[j] = ok ? a.col_part[tt * 2 * C + C + c] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < NH; ++j) {
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) T[j] += vT[j][wv];
        C0[j] += vC0[j];
        CL[j] += vCL[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NH; ++j) { const int v = lane + (jh + j) * NT; red[v] = T[j]; red[1024 + v] = C0[j]; red[2048 + v] = CL[j]; }
  }
  se_wave_sync();
  for (int c = lane; c < C; c += NT) {
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
  se_wave_sync();
  {  // phase 2, fully unrolled: rows in (virtual thread, trip) order, chunks of CHK loads; one accumulator set live at a time
    float m[VEC];
#pragma unroll
    for (int r0 = 0; r0 < NROW; r0 += CHK) {
#pragma unroll
      for (int u = 0; u < CHK; ++u) {
        const int r = r0 + u, j = r / NIT, i = r % NIT;
        const int v = lane + j * NT, cg = v % CG, kg = v / CG, k = kg + i * KG;
        if (i == 0) {
#pragma unroll
          for (int q = 0; q < VEC; ++q) m[q] = 0.f;
        }
        if (k < 9 * C) {
          const float s = S[k];
          const uint32_t ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
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
        if (i == NIT - 1) {
#pragma unroll
          for (int q = 0; q < VEC; ++q) red[kg * C + cg * VEC + q] = m[q];
        }
        if (r + CHK < NROW) w[u] = wrow(r + CHK);     // the slot is free: the next chunk's row
      }
    }
  }
  se_wave_sync();
  {
    float part[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int v = lane + j * NT, c = v % C, g = v / C;
      float t = 0.f;
      for (int q = g; q < KG; q += G) t += red[q * C + c];
      part[j] = t;
    }
    se_wave_sync();
#pragma unroll
    for (int j = 0; j < NV; ++j) red[lane + j * NT] = part[j];
    se_wave_sync();
    for (int c = lane; c < C; c += NT) {
      float t = 0.f;
      for (int q = 0; q < G; ++q) t += red[q * C + c];
      y[c] = t / (float)(hb * a.wout) * a.scale2[c] + a.shift2[c];
    }
  }
  se_wave_sync();
  constexpr int NF = C * R;
  for (int idx = lane; idx < NF; idx += NT) { red[idx] = a.fc1[idx]; red[4096 + idx] = a.fc2[idx]; }
  se_wave_sync();
  {
    constexpr int NS = C / 16;
    for (int v = lane; v < R * NS; v += NT) {
      const int r1 = v % R, sl = v / R;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s = fmaf(red[r1 * C + sl * 16 + k], y[sl * 16 + k], s);
      S[sl * R + r1] = s;
    }
    se_wave_sync();
    if (lane < R) {
      float s = 0.f;
      for (int q = 0; q < NS; ++q) s += S[q * R + lane];
      hid[lane] = relu_nan(s);
    }
  }
  se_wave_sync();
  for (int c = lane; c < C; c += NT) {
    float z = 0.f;
    for (int k = 0; k < R; ++k) z = fmaf(red[4096 + c * R + k], hid[k], z);
    gate_out[c] = 1.f / (1.f + expf(-z));
  }
  se_wave_sync();
}

constexpr int SE_GATE_SCRATCH_FLOATS = 8192 + 9 * 256 + 256 + 16;   // red, S, y, hid

}  // namespace sk
