// Stem convolution (1 -> 32) of the HalfResNet34 trunk.  Reference: sidekit/nnet/res_net.py:509-515,549.  (The squeeze-excite gate
// kernel lives in se_gate.hip.)
#include "kernels.h"

namespace sk {

// ---- stem: relu(bn(conv3x3(1->32, pad 1))) on the logical (B,1,H=T,W=80) image -----------------
// One workgroup = TT time rows x 80 freqs; the (TT+2) x 82 input patch goes through LDS, every
// thread produces one position x 32 channels (288 scalar FMAs with the weights as SGPR operands; rounds 3-4 issued them as 144 explicit
// v_pk_fma_f32 -- no faster inside the pipelined step, and packed f32 is gone from the library since round 5, csrc/Makefile) and writes
// 64 B (bf16) / 128 B (f32).
constexpr int STEM_TT = 16;
constexpr int STEM_W = 80;

template <int EB>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ feats, long sb, long sf, long st,
                                                   const float* __restrict__ w, const float* __restrict__ shift,
                                                   unsigned char* __restrict__ out,
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
        for (int c = 0; c < 8; ++c) {      // one v_fma_f32 per (channel, tap) with the weight as an SGPR operand; w is tap-major [9][32], uniform index -> scalar loads
          float s = shift[c0 + c];         // w carries the BatchNorm scale (xt_api.hip): bn(conv(x)) = shift + sum w' x
#pragma unroll
          for (int q = 0; q < 9; ++q) s = fmaf(w[q * 32 + c0 + c], x[q], s);
          v[c] = relu_nan(s);
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

int launch_stem(const float* feats, long sb, long sf, long st, const float* w, const float* shift,
                void* out, int dtype, Lens lens, int B, int T, hipStream_t s) {
  const int tiles = cdiv(T, STEM_TT);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(stem_kernel<2>, dim3(B * tiles), dim3(256), 0, s, feats, sb, sf, st, w, shift,
                       (unsigned char*)out, lens, T);
  else
    hipLaunchKernelGGL(stem_kernel<4>, dim3(B * tiles), dim3(256), 0, s, feats, sb, sf, st, w, shift,
                       (unsigned char*)out, lens, T);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

}  // namespace sk
