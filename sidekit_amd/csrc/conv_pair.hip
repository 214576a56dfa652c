// A/B BUILD ONLY (csrc/Makefile `make ab`; SIDEKIT_AMD_PAIR=1): built in round 6 as the round-5 verdict specified, bit-identical to the two launches it
// replaces, and measured SLOWER than them alone and in the forward (profiles/r06_conv_pair_L1.txt has the phase stamps: the fused workgroup's chain of
// barrier-separated memory phases is twice as long while a CU still holds only two workgroups, so the pass of HBM traffic it saves is paid for twice over
// in exposed latency).  Kept with its test (tests/test_gpu_halfresnet.py::test_layer1_pair_kernel_gives_the_bits_of_the_two_launches) as the record.
//
// conv2 of BasicBlock k and conv1 of BasicBlock k + 1 in ONE kernel, layer 1 of the HalfResNet34 trunk (bf16 path): the block output Y_k
// goes from conv2's epilogue to conv1's k-loop through LDS, so HBM sees it once (the write) instead of twice.  Reference arithmetic:
// BasicBlock.forward, sidekit/nnet/res_net.py:309-320, two consecutive blocks of `layer1` (res_net.py:518).
//
// Why this pair and no other: the block dataflow is conv1 R X, W O1; [SE gate from conv1's sums]; conv2 R O1, R X, W Y
// -- 5 activation passes -- and the next block's conv1 reads Y back.  conv1 -> conv2 of ONE block cannot go through LDS (the gate needs the
// sums of the whole plane of O1 before conv2's epilogue may run); conv2(k) -> conv1(k + 1) can: gate_k is known before the launch, and
// conv1(k + 1)'s statistics epilogue leaves the sums gate_{k+1} is derived from, as the stand-alone kernel does.  Per pair the HBM traffic falls
// from R O1, R X, W Y | R Y, W O1' (5 passes) to R O1, R X, W Y, W O1' (4): layer 1 moves 13 instead of 15 passes per forward.
//
// One work item = (utterance, 8 output rows of conv1(k + 1)) = exactly the stand-alone statistics kernel's tile (conv3x3.hip, B_L1: TH = 8, four
// waves x five 32-position MFMA tiles, linear lanes), so the per-tile sums, their order and therefore the SE gate are the stand-alone kernel's
// BITS.  Phase A computes the ten rows of Y that tile needs (eight + one halo row either side: 1.25 x conv2's MFMAs; 25 MFMA tiles over four
// waves = seven per wave) from twelve staged rows of O1, applies bn2 * gate + shortcut + ReLU exactly as the residual epilogues of
// conv3x3_kernel do (same operations per element in the same order: the intermediate bn2 * gate value is rounded to bf16 before the shortcut is
// added, as there), writes the eight interior rows to HBM and leaves all ten in the swizzled LDS image conv1's taps read -- rows outside the
// utterance as zeros, which is conv1's zero padding.  Phase B is the stand-alone statistics-form convolution on that image.
//
// Resources: 62 208 B halo image (twelve rows; Y and then conv1's transposed output tile reuse it) + 18 432 B of conv1's weight fragments in LDS
// (phase A keeps conv2's 18 fragments in 72 registers beside 112 accumulator registers; a second resident set does not fit the 256 registers
// of two workgroups per CU) = 63 LDS granules: two persistent workgroups per CU, as the stand-alone kernel has.
#include "kernels.h"

namespace sk {

namespace {

constexpr int PC = 32, PW = 80, PTH = 8;            // channels, width, output rows of phase B per item
constexpr int PRY = PTH + 2, PRIN = PTH + 4;        // rows of Y per item; staged rows of O1
constexpr int PCB = PC * 2;                         // bytes per position
constexpr int PRS = (PW + 1) * PCB;                 // LDS row stride: 80 positions + the zero position that serves as column -1 and column 80
constexpr int PSPP = PCB / 16, PPPR = (PW * PSPP + 63) / 64;   // 16-B slots per position, 1-KiB DMA pieces per row
constexpr int PIMG = PRIN * PRS;                    // 62 208
constexpr int PNK = 18, PKS = 2;                    // k-steps (9 taps x 2 x 16 channels)
constexpr int PMWA = 7, PMTA = PRY * PW;            // phase A: 800 positions = 25 tiles of 32 over 4 waves
constexpr int PMWB = 5, PMTB = PTH * PW;            // phase B: 640 positions = 20 tiles
constexpr int POPS = PC * 2 + 16;                   // phase B's transposed output tile: position stride (conv3x3.hip OPS)
constexpr int PWLDS = PNK * 1024;                   // conv1's weight fragments
static_assert(PMTB * POPS <= PIMG && PPPR * 1024 <= PRS, "the output tile and a row's DMA pieces fit the image");
static_assert((PIMG + PWLDS) % 1280 == 0 && 2 * (PIMG + PWLDS) <= 160 * 1024, "two workgroups per CU");

__device__ inline int swz(int col) { return (col >> 2) & 3; }   // conv3x3.hip ConvCfg::swz_key for 64-B positions, linear lanes
// k-step order of the swizzled linear-lane image (conv3x3.hip step_tap / step_ks / kord): (dw, ks, dh) nested, dh innermost
__device__ constexpr int k_tap(int kk) { return (kk % 3) * 3 + kk / (3 * PKS); }
__device__ constexpr int k_ks(int kk) { return (kk / 3) % PKS; }
__device__ constexpr int k_ord(int kk) { return k_tap(kk) * PKS + k_ks(kk); }

__device__ inline f32x16 mma(f32x16 acc, const uint4& w, const uint4& x) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

}  // namespace

// RSC: block k is the first block of the layer -- its shortcut is bn_s(conv1x1(block input)) evaluated in phase A's epilogue on the matrix
// cores (conv3x3.hip FORM_RESID_SC) instead of a stored tensor.
template <bool RSC>
__global__ __launch_bounds__(256, 2) void conv_pair_l1_kernel(ConvPairArgs a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[PIMG];
  __shared__ __attribute__((aligned(1024))) unsigned char wlds[PWLDS];
  const int tid0 = threadIdx.x;
  const int wm = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  const int tiles = (a.H + PTH - 1) / PTH;
  // XCD-aware persistent walk, as conv3x3_kernel's weight-resident shapes: every XCD owns a contiguous run of (utterance, row tile) items
  const int bidx = (int)blockIdx.x;
  const int nwork = a.B * tiles, q8 = nwork >> 3, r8 = nwork & 7, xcd = bidx & 7;
  const int wfirst = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8, wcount = q8 + (xcd < r8 ? 1 : 0);
  const int wstride = (int)(gridDim.x >> 3) + (xcd < (int)(gridDim.x & 7) ? 1 : 0);

  const unsigned lane16 = (unsigned)(tid0 & 63) * 16u;
  // conv2's fragments: resident in registers, in k-step order
  uint4 w2[PNK];
  {
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w2pack);
#pragma unroll
    for (int kk = 0; kk < PNK; ++kk) w2[kk] = *reinterpret_cast<const uint4*>(wb + (unsigned)(k_ord(kk) * 1024) + lane16);
  }
  // conv1's fragments: LDS, in k-step order (every wave reads the same 1 KB per k-step, 16 B per lane: conflict-free)
  {
    const unsigned char* wb = reinterpret_cast<const unsigned char*>(a.w1pack);
    for (int s = tid0; s < PNK * 64; s += 256) {
      const int kk = s >> 6, ln = s & 63;
      *reinterpret_cast<uint4*>(wlds + kk * 1024 + ln * 16) = *reinterpret_cast<const uint4*>(wb + k_ord(kk) * 1024 + ln * 16);
    }
  }
  const unsigned char* in = reinterpret_cast<const unsigned char*>(a.in);
  unsigned char* yout = reinterpret_cast<unsigned char*>(a.y_out);
  unsigned char* oout = reinterpret_cast<unsigned char*>(a.o_out);

  auto stamp = [&](int k) {   // diagnostic path only (a.stamps == nullptr in the product)
    if (a.stamps && tid0 == 0) {
      unsigned long long* sp = a.stamps + (size_t)bidx * 16;
      sp[k] = __builtin_amdgcn_s_memtime();
      if (k == 0) sp[15] = __builtin_amdgcn_s_memrealtime();
      if (k == 9) sp[15] = __builtin_amdgcn_s_memrealtime() - sp[15];
    }
  };
  auto do_item = [&](int work, bool first_item) -> bool {
    int tid = tid0;
    asm volatile("" : "+v"(tid));   // keep per-item address arithmetic out of the persistent loop's invariants (registers)
    const int lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int b = work / tiles, tile = work % tiles;
    const int ho0 = tile * PTH;
    const int hb = a.lens.get_uniform(b);        // rows of this utterance (layer 1: no halving)
    if (ho0 >= hb) return false;
    if (!first_item) __syncthreads();            // the previous item's copy-out has left the LDS
    stamp(0);
    // ---- stage rows ho0 - 2 .. ho0 + 9 of O1
    for (int s = tid; s < PRIN * PSPP; s += 256)   // the zero position after each row
      *reinterpret_cast<uint4*>(smem + (s / PSPP) * PRS + PW * PCB + (s % PSPP) * 16) = make_uint4(0, 0, 0, 0);
    for (int it = wm; it < PRIN * PPPR; it += 4) {
      const int row = it / PPPR, q = it % PPPR;
      const int hi = ho0 - 2 + row;
      const bool rowok = hi >= 0 && hi < hb;
      const int slot = q * 64 + lane, col = slot / PSPP, cs = slot % PSPP;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(a.zeros);
      if (rowok) src = in + (((size_t)b * a.H + hi) * PW + col) * PCB + ((cs ^ swz(col)) << 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(smem + row * PRS + q * 1024), 16, 0, 0);
    }
    // conv2's epilogue constants for this utterance, requested while the tile lands (a pass's epilogue otherwise waits one L2 round trip per channel
    // group): residual form k1 = scale2 * gate, k0 = shift2 * gate; first-block form k0 = shift2 * gate + shift_s -- the products conv3x3.hip forms
    // per value, formed once per channel here (the same two roundings)
    f32x4 k1[4], k0[4];
    {
      const float* gate_b = a.gate + (size_t)b * PC;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale2 + 8 * g + 4 * h), sh = *reinterpret_cast<const f32x4*>(a.shift2 + 8 * g + 4 * h);
        const f32x4 gt = *reinterpret_cast<const f32x4*>(gate_b + 8 * g + 4 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) { k1[g][q] = sc[q] * gt[q]; k0[g][q] = sh[q] * gt[q]; }
        if constexpr (RSC) {
          const f32x4 h2 = *reinterpret_cast<const f32x4*>(a.sc_shift + 8 * g + 4 * h);
#pragma unroll
          for (int q = 0; q < 4; ++q) k0[g][q] = k0[g][q] + h2[q];
        }
      }
    }
    stamp(1);
    __syncthreads();
    stamp(2);
    __builtin_amdgcn_s_setprio(0);

    // ================= phase A: Y rows ho0 - 1 .. ho0 + 8 =================
    // A wave's seven tiles in two passes over the k-loop (tiles 0-3, then 4-6): 112 live accumulator registers beside conv2's 72 resident weight
    // registers left the compiler 25-35 spilled registers; a pass's accumulators go through the epilogue arithmetic at once and wait as packed
    // bf16 (8 registers per tile) until every wave is done with the O1 image.  Same MFMAs, same k order per tile: same bits.
    uint2 park[PMWA][4];
    {
      auto pos = [&](int i) { return (wm * PMWA + i) * 32 + r; };   // position in the ten-row Y tile; >= 800: none (wave 3's last tiles)
      auto pass = [&](auto i0_tag, auto n_tag) {
        constexpr int I0 = decltype(i0_tag)::value, N = decltype(n_tag)::value;
        f32x16 acc[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
        int base[N][3];
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const int m = pos(I0 + i), mc = m < PMTA ? m : PMTA - 1;
          const int row = mc / PW, col = mc % PW;
#pragma unroll
          for (int dw = 0; dw < 3; ++dw) {
            int cx = col + dw - 1;
            if (cx < 0) cx = PW;
            base[i][dw] = row * PRS + cx * PCB + ((h ^ swz(cx)) << 4);
          }
        }
        auto xa = [&](int i, int kk) {
          const int t = k_tap(kk), ks = k_ks(kk);
          return smem + (base[i][t % 3] ^ (ks << 5)) + (t / 3) * PRS;
        };
        uint4 xc[N], xn[N];
#pragma unroll
        for (int i = 0; i < N; ++i) xc[i] = *reinterpret_cast<const uint4*>(xa(i, 0));
#pragma unroll
        for (int kk = 0; kk < PNK; ++kk) {
          if (kk + 1 < PNK) {
#pragma unroll
            for (int i = 0; i < N; ++i) xn[i] = *reinterpret_cast<const uint4*>(xa(i, kk + 1));
          }
          const uint4 wf = w2[kk];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < N; ++i) acc[i] = mma(acc[i], wf, xc[i]);
          if (kk + 1 < PNK) {
#pragma unroll
            for (int i = 0; i < N; ++i) xc[i] = xn[i];
          }
        }
        if constexpr (!RSC) {
          // bn2 * gate, rounded to bf16 (conv3x3.hip `cell`, residual form, LEAN walk: per channel group, fmaf with scale * gate and shift * gate)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
              float v[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = fmaf(acc[i][4 * g + q], k1[g][q], k0[g][q]);
              uint32_t p0 = pack_bf16x2(v[0], v[1]), p1 = pack_bf16x2(v[2], v[3]);
              asm volatile("" : "+v"(p0), "+v"(p1));   // computed HERE: the compiler otherwise sinks the arithmetic below the barrier and keeps the f32 accumulators alive through the second pass
              park[I0 + i][g] = make_uint2(p0, p1);
            }
          }
        } else {
          // first block of the layer: x = k1 * acc + conv1x1(x_in) [bn_s scale folded into the weights] + k0, k1 = scale2 * gate, k0 = shift2 * gate + shift_s
          // (conv3x3.hip FORM_RESID_SC, 32x32 MFMA branch: the same operations per element in the same order)
          const unsigned char* xin = reinterpret_cast<const unsigned char*>(a.sc_in);
          const unsigned char* scb = reinterpret_cast<const unsigned char*>(a.sc_wpack);
          uint4 wx[PKS];
#pragma unroll
          for (int ks = 0; ks < PKS; ++ks) wx[ks] = *reinterpret_cast<const uint4*>(scb + (unsigned)(ks * 1024) + lane16);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < N; ++i)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[i][4 * g + q] *= k1[g][q];
          auto xload = [&](int i, uint4* dst) {
            const int m = pos(I0 + i);
            const int row = m / PW, col = m % PW, grow = ho0 - 1 + row;
            const bool ok = m < PMTA && grow >= 0 && grow < hb;
            const unsigned char* xp = xin + (((size_t)b * a.H + (ok ? grow : 0)) * PW + (ok ? col : 0)) * PCB + h * 16;
#pragma unroll
            for (int ks = 0; ks < PKS; ++ks) dst[ks] = ok ? *reinterpret_cast<const uint4*>(xp + ks * 32) : make_uint4(0, 0, 0, 0);
          };
          uint4 xf[N][PKS];     // every tile's block-input fragments in flight at once: one L2 / HBM round trip per pass instead of one per tile
#pragma unroll
          for (int i = 0; i < N; ++i) xload(i, xf[i]);
#pragma unroll
          for (int i = 0; i < N; ++i)
#pragma unroll
            for (int ks = 0; ks < PKS; ++ks) acc[i] = mma(acc[i], wx[ks], xf[i][ks]);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
              const int grow = ho0 - 1 + pos(I0 + i) / PW;
              const bool ok = grow >= 0 && grow < hb;     // rows outside the utterance: conv1's zero padding
              float v[4];
#pragma unroll
              for (int q = 0; q < 4; ++q) v[q] = relu_nan(acc[i][4 * g + q] + k0[g][q]);
              uint32_t p0 = ok ? pack_bf16x2(v[0], v[1]) : 0u, p1 = ok ? pack_bf16x2(v[2], v[3]) : 0u;
              asm volatile("" : "+v"(p0), "+v"(p1));
              park[I0 + i][g] = make_uint2(p0, p1);
            }
          }
        }
      };
      pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
      pass(std::integral_constant<int, 4>{}, std::integral_constant<int, 3>{});
      __builtin_amdgcn_s_setprio(3);
      stamp(3);
    }
    // ---- Y: (+ shortcut, ReLU,) interior rows to HBM, all ten rows stay in the image
    {
      constexpr int CPR = PSPP, NCH = PMTA * CPR, NIT = (NCH + 255) / 256;     // 3200 16-B chunks, 13 rounds
      const unsigned char* scut = reinterpret_cast<const unsigned char*>(a.shortcut);
      // chunk idx of the ten-row tile sits (idx - 80 * CPR) * 16 bytes from the first interior row in both tensors
      const long run0 = (((long)b * a.H + ho0) * PW) * PCB - (long)PW * PCB;
      uint4 sreg[RSC ? 1 : NIT];
      if constexpr (!RSC) {     // the accumulators are dead: all thirteen shortcut chunks travel while the waves meet and the image is rewritten
#pragma unroll
        for (int q = 0; q < NIT; ++q) {
          const int idx = tid + q * 256, grow = ho0 - 1 + idx / (PW * CPR);
          sreg[q] = (idx < NCH && grow >= 0 && grow < hb) ? *reinterpret_cast<const uint4*>(scut + run0 + (long)idx * 16) : make_uint4(0, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      auto pos = [&](int i) { return (wm * PMWA + i) * 32 + r; };
      __syncthreads();   // every wave is done with the O1 image: Y may overwrite it
#pragma unroll
      for (int i = 0; i < PMWA; ++i) {
        const int m = pos(i);
        if (m < PMTA) {
          const int row = m / PW, col = m % PW;
          unsigned char* lp = smem + row * PRS + col * PCB + h * 8;
#pragma unroll
          for (int g = 0; g < 4; ++g) *reinterpret_cast<uint2*>(lp + ((g ^ swz(col)) << 4)) = park[i][g];
        }
      }
      stamp(4);
      __syncthreads();   // the image holds bn2 * gate (or, RSC, the finished Y)
#pragma unroll
      for (int q = 0; q < NIT; ++q) {
        const int idx = tid + q * 256;
        if (idx >= NCH) break;
        const int p = idx / CPR, cc = idx % CPR, row = p / PW, col = p % PW, grow = ho0 - 1 + row;
        const bool ok = grow >= 0 && grow < hb;
        unsigned char* lp = smem + row * PRS + col * PCB + ((cc ^ swz(col)) << 4);
        uint4 v = *reinterpret_cast<const uint4*>(lp);
        if constexpr (!RSC) {
          const uint4 sv = sreg[q];
          const uint32_t vv[4] = {v.x, v.y, v.z, v.w}, ss[4] = {sv.x, sv.y, sv.z, sv.w};
          uint32_t rr[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            rr[e] = pack_bf16x2(relu_nan(__builtin_bit_cast(float, vv[e] << 16) + __builtin_bit_cast(float, ss[e] << 16)),
                                relu_nan(__builtin_bit_cast(float, vv[e] & 0xffff0000u) + __builtin_bit_cast(float, ss[e] & 0xffff0000u)));
          v = ok ? make_uint4(rr[0], rr[1], rr[2], rr[3]) : make_uint4(0, 0, 0, 0);
          *reinterpret_cast<uint4*>(lp) = v;
        }
        if (ok && row >= 1 && row <= PTH) *reinterpret_cast<uint4*>(yout + run0 + (long)idx * 16) = v;
      }
    }
    stamp(5);
    __syncthreads();   // Y complete in the image
    stamp(6);
    __builtin_amdgcn_s_setprio(0);

    // ================= phase B: conv1 of the next block on the Y image, statistics form (conv3x3.hip B_L1, FORM_STATS) =================
    f32x16 acc[PMWB];
#pragma unroll
    for (int i = 0; i < PMWB; ++i)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    {
      int base[PMWB][3];
#pragma unroll
      for (int i = 0; i < PMWB; ++i) {
        const int m = (wm * PMWB + i) * 32 + r;
        const int row = m / PW, col = m % PW;
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
          int cx = col + dw - 1;
          if (cx < 0) cx = PW;
          base[i][dw] = row * PRS + cx * PCB + ((h ^ swz(cx)) << 4);
        }
      }
      auto xb = [&](int i, int kk) {
        const int t = k_tap(kk), ks = k_ks(kk);
        return smem + (base[i][t % 3] ^ (ks << 5)) + (t / 3) * PRS;
      };
      uint4 xc[PMWB], xn[PMWB], wc, wn;
#pragma unroll
      for (int i = 0; i < PMWB; ++i) xc[i] = *reinterpret_cast<const uint4*>(xb(i, 0));
      wc = *reinterpret_cast<const uint4*>(wlds + lane * 16);
#pragma unroll
      for (int kk = 0; kk < PNK; ++kk) {
        if (kk + 1 < PNK) {
#pragma unroll
          for (int i = 0; i < PMWB; ++i) xn[i] = *reinterpret_cast<const uint4*>(xb(i, kk + 1));
          wn = *reinterpret_cast<const uint4*>(wlds + (kk + 1) * 1024 + lane * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PMWB; ++i) acc[i] = mma(acc[i], wc, xc[i]);
        if (kk + 1 < PNK) {
#pragma unroll
          for (int i = 0; i < PMWB; ++i) xc[i] = xn[i];
          wc = wn;
        }
      }
    }
    __builtin_amdgcn_s_setprio(3);
    stamp(7);
    const int mvalid = (hb - ho0) * PW < PMTB ? (hb - ho0) * PW : PMTB;
    const size_t gpos0 = ((size_t)b * a.H + ho0) * PW;
    __syncthreads();   // every wave is done with the Y image
    {
      float ssum[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) ssum[q] = 0.f;
      const bool full = mvalid == PMTB;   // wave-uniform
      f32x4 sc_n = *reinterpret_cast<const f32x4*>(a.scale1 + 4 * h), sh_n = *reinterpret_cast<const f32x4*>(a.shift1 + 4 * h);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 sc = sc_n, sh = sh_n;
        if (g + 1 < 4) {
          sc_n = *reinterpret_cast<const f32x4*>(a.scale1 + 8 * (g + 1) + 4 * h);
          sh_n = *reinterpret_cast<const f32x4*>(a.shift1 + 8 * (g + 1) + 4 * h);
        }
#pragma unroll
        for (int i = 0; i < PMWB; ++i) {
          const int m = (wm * PMWB + i) * 32 + r;
          const bool valid = full || m < mvalid;
          float v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = relu_nan(fmaf(acc[i][4 * g + q], sc[q], sh[q]));
          const uint32_t u01 = pack_bf16x2(v[0], v[1]), u23 = pack_bf16x2(v[2], v[3]);
          const float rv[4] = {__builtin_bit_cast(float, u01 << 16), __builtin_bit_cast(float, u01 & 0xffff0000u),
                               __builtin_bit_cast(float, u23 << 16), __builtin_bit_cast(float, u23 & 0xffff0000u)};
#pragma unroll
          for (int q = 0; q < 4; ++q) ssum[4 * g + q] += valid ? rv[q] : 0.f;
          *reinterpret_cast<uint2*>(smem + m * POPS + (8 * g + 4 * h) * 2) = make_uint2(u01, u23);
        }
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) ssum[q] = half_sum_upper_row(ssum[q]);
      if (r == 16) {
        float* sp = a.se_part + (((size_t)b * tiles + tile) * 4 + wm) * PC;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(sp + 4 * h + 8 * g) = make_float4(ssum[4 * g], ssum[4 * g + 1], ssum[4 * g + 2], ssum[4 * g + 3]);
      }
    }
    __syncthreads();   // output tile complete
    stamp(8);
    auto lds_elem = [&](int m, int c) { return bf16_to_f32(*reinterpret_cast<const uint16_t*>(smem + m * POPS + c * 2)); };
    if (tid < 2 * PC) {   // edge sums for the zero padding of the conv that follows (conv3x3.hip, statistics form: same chains, same order)
      const int side = tid >= PC ? 1 : 0;
      const int c = tid - side * PC, rows_valid = mvalid / PW, hl = hb - 1;
      const int wcol = side ? PW - 1 : 0;
      float cs = 0.f;
      for (int hr = 0; hr < rows_valid; ++hr) cs += lds_elem(hr * PW + wcol, c);
      a.col_part[((size_t)b * tiles + tile) * 2 * PC + side * PC + c] = cs;
      float* eg = a.edge + (size_t)b * 6 * PC + c;
      if (tile == 0) {
        if (side == 0) {
          float s = 0.f;
          for (int wo = 0; wo < PW; ++wo) s += lds_elem(wo, c);
          eg[0] = s;
          eg[2 * PC] = lds_elem(0, c);
        } else {
          eg[3 * PC] = lds_elem(PW - 1, c);
        }
      }
      if (tile == hl / PTH) {
        const int m0 = (hl - ho0) * PW;
        if (side == 1) {
          float s = 0.f;
          for (int wo = 0; wo < PW; ++wo) s += lds_elem(m0 + wo, c);
          eg[1 * PC] = s;
          eg[5 * PC] = lds_elem(m0 + PW - 1, c);
        } else {
          eg[4 * PC] = lds_elem(m0, c);
        }
      }
    }
    {
      constexpr int CPR = PC * 2 / 16, NIT = PMTB * CPR / 256;   // 4 chunks per position, 10 rounds
      const size_t run0 = gpos0 * PC * 2;
      const unsigned lane_lds = ((unsigned)tid / CPR) * POPS + ((unsigned)tid % CPR) * 16;
#pragma unroll
      for (int q = 0; q < NIT; ++q) {
        const int idx = tid + q * 256;
        if (idx >= mvalid * CPR) break;
        *reinterpret_cast<uint4*>(oout + run0 + (unsigned)idx * 16u) = *reinterpret_cast<const uint4*>(smem + lane_lds + q * (256 / CPR) * POPS);
      }
    }
    stamp(9);
    return true;
  };
  __syncthreads();   // conv1's weights are in LDS
  bool first_item = true;
  for (int wi = bidx >> 3; wi < wcount; wi += wstride)
    if (do_item(wfirst + wi, first_item)) first_item = false;
}

static int pair_cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
    else n = 256;
  }
  return n;
}

int launch_conv_pair(const ConvPairArgs& a, hipStream_t st) {
  SK_CHECK(a.C == PC && a.W == PW, SK_EARG, "conv pair: only the layer-1 geometry (32 channels x 80) is built");
  SK_CHECK(a.in && a.w2pack && a.scale2 && a.shift2 && a.gate && a.y_out && a.w1pack && a.scale1 && a.shift1 && a.o_out && a.se_part && a.col_part && a.edge && a.zeros,
           SK_EARG, "conv pair: missing argument");
  SK_CHECK((a.shortcut != nullptr) != (a.sc_in != nullptr), SK_EARG, "conv pair: either a shortcut tensor or the first block's in-place shortcut");
  SK_CHECK(!a.sc_in || (a.sc_wpack && a.sc_shift), SK_EARG, "conv pair: in-place shortcut needs its folded weights and shift");
  SK_CHECK(a.B > 0 && a.H > 0, SK_EARG, "conv pair: empty problem");
  const int tiles = cdiv(a.H, PTH), nwork = a.B * tiles;
  const int per_cu = (a.persist_cap > 0 && a.persist_cap < 2) ? a.persist_cap : 2;
  const int resident = pair_cu_count() * per_cu;
  const dim3 grid((unsigned)(nwork > resident ? resident : nwork)), block(256);
  if (a.sc_in) hipLaunchKernelGGL(conv_pair_l1_kernel<true>, grid, block, 0, st, a);
  else hipLaunchKernelGGL(conv_pair_l1_kernel<false>, grid, block, 0, st, a);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
