// Internal kernel-launcher interface of libsidekit_amd (not part of the C ABI).
#pragma once
#include "common.h"

namespace sk {

enum { DT_F32 = 0, DT_BF16 = 1, DT_F64 = 2, DT_I64 = 3 };

// ---- conv3x3.hip ------------------------------------------------------------------------
enum ConvShape {
  CONV_L1 = 0, CONV_L1S, CONV_L2A, CONV_L2S, CONV_L2, CONV_L3A, CONV_L3S, CONV_L3, CONV_L4A, CONV_L4S, CONV_L4,
  CONV_NSHAPES,
  // small-grid forms of CONV_L3 / CONV_L4 (conv3x3.hip: 3- / 5-row tiles, residual forms only; ids 11-41 and 44-46 are A/B alternatives)
  CONV_L3T = 42, CONV_L4T = 43, CONV_L1G = 47   // L1G: layer 1's residual forms beside a gate wave (weights through a ring instead of resident)
};

struct SeArgs {
  const float* se_part; const float* col_part; const float* edge;   // as written by the statistics-mode conv
  int tiles, wm, th;
  const void* w2t;       // conv2 weights as consumed by the MFMA, [tap][ci][co]: bf16 in the bf16 path (w2t_bf16), else f32
  int w2t_bf16;
  const float* scale2; const float* shift2;
  const float* fc1; const float* fc2;   // se.fc.0 [C/16][C], se.fc.2 [C][C/16]
  float* gate;           // [B][C]
  Lens lens; int halvings; int wout; int C; int B;
};

struct ConvArgs {
  const void* in;      // [B][Hin][WIN][CIN]   (bf16 or f32)
  const void* wpack;   // fragment-ordered weights
  const float* scale;  // [COUT] folded BatchNorm scale
  const float* shift;  // [COUT] folded BatchNorm shift
  void* out;           // [B][Hout][WOUT][COUT]
  // -- statistics mode (conv1 of a BasicBlock; se_part != nullptr): sums of the stored (rounded) output that determine
  //    the plane mean of the 3x3 convolution that FOLLOWS (SELayer input, res_net.py:278-279) by linearity
  float* se_part;      // [B][tiles][WM][COUT] per-workgroup totals
  float* col_part;     // [B][tiles][2][COUT]  first / last column sums over the tile's valid rows
  float* edge;         // [B][6][COUT]         first-row sum, last-row sum, corners (0,0) (0,W-1) (last,0) (last,W-1)
  // -- residual mode (conv2 of a BasicBlock; gate != nullptr): out = relu(bn(conv) * gate[b][c] + shortcut)  (res_net.py:316-319)
  // -- fused 1x1 shortcut (first block of a layer; sc_wpack != nullptr): bn(conv1x1_stride(x)) from the centre tap of
  //    the halo tile this convolution has staged anyway (res_net.py:301-307), written to sc_out
  const void* sc_wpack; const float* sc_scale; const float* sc_shift; void* sc_out;
  const float* gate;   // [B][COUT]
  const void* shortcut;  // NHWC, same shape and type as out
  // -- residual mode of the FIRST block of a layer (gate != nullptr && sc_in != nullptr): the shortcut is not a stored
  //    tensor but bn(conv1x1_stride(x)) of the block input x (res_net.py:301-307), computed in this epilogue from x's
  //    own rows: sc_in = x [B][sc_hin][WOUT*s][cin_x], weights sc_wpack / sc_scale / sc_shift
  const void* sc_in; int sc_hin;
  const void* zeros;   // >= 16 zero bytes in device memory (source of the conv zero padding)
  Lens lens;           // feature frames per utterance
  int halvings_in;     // stride-2 stages between the features and this conv's input
  int B, Hin, Hout;    // allocated rows of in / out
  int relu;
  int persist_cap;     // > 0: at most this many workgroups per CU for the persistent (weight-resident) shapes, see launch_cfg
  // -- residual mode with the SE gate computed in this kernel's prologue (gate_pro != 0; small grids only, se_gate_inl.h): `gate` is not read,
  //    every workgroup derives its utterance's gate from conv1's sums (`se`, what launch_se_pre would have been handed)
  int gate_pro;
  SeArgs se;
  unsigned long long* stamps;  // diagnostics only: per-workgroup s_memtime stamps at the phase boundaries (8 per block), or nullptr
  int dbg;             // diagnostics only (sk_bench_conv): bit0 skip stores, bit1 skip MFMA loop, bit2 skip staging
};

struct ConvGeom { int cin, cout, stride, win, th, wm, ck, taps, ks, eb, nw; int m16; };   // m16: weights packed for v_mfma_f32_16x16x32_bf16 (16 x 32 fragments)

int conv_geom(int shape, int dtype, ConvGeom* g);
int launch_conv(int shape, int dtype, const ConvArgs& a, hipStream_t st);
size_t conv_pack_bytes(const ConvGeom& g);
void conv_pack_weights(const ConvGeom& g, const float* w, int kh_kw, void* dst);

// ---- conv_pair.hip (A/B builds only) ----------------------------------------------------------
// conv2 of BasicBlock k + conv1 of BasicBlock k + 1 of layer 1 in one kernel (bf16): the block output Y_k passes from conv2's epilogue to conv1's
// k-loop through LDS and is written to HBM once, never read back (res_net.py:309-320, two consecutive blocks).  Same numbers, bit for bit, as
// launch_conv(residual form) followed by launch_conv(statistics form).
struct ConvPairArgs {
  int C, W;              // geometry: 32 channels x 80 columns (layer 1) is what is built
  // -- conv2 of block k (residual form)
  const void* in;        // O1_k  [B][H][W][C] bf16: conv1_k's output
  const void* w2pack;    // conv2_k weights, fragment order (conv_pack_weights, the layer's stride-1 shape)
  const float* scale2; const float* shift2;   // bn2_k folded
  const float* gate;     // [B][C] SE gate of block k (launch_se_pre)
  const void* shortcut;  // X_k [B][H][W][C] bf16: the block input (identity shortcut) -- or nullptr with the in-place 1x1 shortcut below
  const void* sc_in; const void* sc_wpack; const float* sc_shift;   // first block of the layer: x_in, folded 1x1 weights, shortcut BatchNorm shift
  void* y_out;           // Y_k [B][H][W][C] bf16
  // -- conv1 of block k + 1 (statistics form)
  const void* w1pack; const float* scale1; const float* shift1;
  void* o_out;           // O1_{k+1}
  float* se_part; float* col_part; float* edge;   // as ConvArgs (tiles of 8 rows, 4 wave rows)
  const void* zeros;
  Lens lens;             // rows per utterance (layer 1: the feature frames)
  int B, H;              // utterances, allocated rows
  int persist_cap;
  unsigned long long* stamps;   // diagnostics only (sk_bench_conv shape 48): 16 s_memtime stamps per workgroup (its last item), or nullptr
};
int launch_conv_pair(const ConvPairArgs& a, hipStream_t st);

// ---- trunk_misc.hip -----------------------------------------------------------------------
// stem: features (strides sf, st in elements) -> relu(bn(conv3x3 1->32)) NHWC [B][T][80][32]; w = tap-major [9][32] weights with the
// BatchNorm scale folded in, shift = the BatchNorm shift
int launch_stem(const float* feats, long sb, long sf, long st, const float* w, const float* shift, void* out, int dtype, Lens lens, int B,
                int T, hipStream_t s);
// SE gate of a BasicBlock computed BEFORE its second convolution runs: the plane mean of bn2(conv2(o1)) is linear in
// o1, so it follows from the sums conv1 left behind (total, first/last row and column, corners) and conv2's weights:
//   mean[co] = scale2[co] / (H W) * sum_{ci,dh,dw} W2[co][ci][dh][dw] * S[ci][dh][dw] + shift2[co]
// then FC -> ReLU -> FC -> sigmoid (res_net.py:262-281).
int launch_se_pre(const SeArgs& a, hipStream_t s);

// ---- gemm.hip ---------------------------------------------------------------------------
// C[m][n] = epi( sum_k A(m,k) * W[n][k] ), fp32 MFMA (exact f32 FMA chain), W row-major [N][K].
// A(m,k) comes from a loader:
//   A_PLAIN  : A[(m + (k / kc) * dil) * lda + k % kc]  (kc == 0: A[m*lda + k]); f32 or bf16 rows
//              -- the (kc, dil) form is a dilated "valid" conv1d over channel-contiguous rows (TDNN)
//   A_FRAMES : window[k] * preemph(wav_b)[reflect(t*hop - win/2 + k)]   (STFT framing, row m = (b,t))
//   A_POWER  : S[m][k]^2 + S[m][k + kc]^2                               (|DFT|^2 from [re | im] rows)
enum { A_PLAIN = 0, A_FRAMES = 1, A_POWER = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU02 = 2, ACT_RELU_BN_TANH = 3, ACT_LOG_EPS = 4 };
struct GemmArgs {
  int a_mode;
  const void* A;        // f32 or bf16 rows (A_PLAIN); f32 wav [B][wav_ld] (A_FRAMES); f32 S (A_POWER)
  int a_bf16;           // A element type (A_PLAIN only)
  long lda;             // elements between consecutive rows of A
  long a_rows;          // rows that exist in A (reads beyond return 0)
  int kc, dil;          // see loader description
  // A_FRAMES only
  const float* window;  // [win]
  const int* nsamples;  // [B] device, or null -> nsamples_uniform
  int nsamples_uniform;
  long wav_ld;
  int hop, t_max;       // frames per utterance slot: row m -> (b = m / t_max, t = m % t_max)
  const int* row_b;     // optional ragged map: row -> utterance, row -> frame
  const int* row_t;
  float preemph;
  const float* W;       // [N][K]
  long ldw;
  float* C;             // [M][ldc]
  long ldc;
  int M, N, K;
  const float* bias;    // [N] or null (added before the activation)
  const float* rowbias; // [M / rows_per_group][N] or null: per-group bias (attention context term)
  int rows_per_group;
  int act;
  const float* scale;   // [N] or null: applied after the activation (BatchNorm that follows it)
  const float* shift;   // [N]
  float alpha;          // final multiplier
  // split-K for skinny problems (M <= 512, K large: 16 workgroups would otherwise walk thousands of k-steps serially):
  const void* W_bf16;   // optional bf16 copy of W ([N][K], ldw elements): with a_mode == A_PLAIN and K % 8 == 0 the product runs on the
                        // bf16 MFMA (operands rounded to bf16, f32 accumulate) -- used by the bf16 compute path only
  float* splitk_ws;     // [ksplit][M][N] partial sums (M <= 512), or null: plain k order.  Non-null selects the sliced summation order for every M
  int ksplit;           // set by launch_gemm: slices run as blockIdx.z (1: inside the workgroup)
  int kslices;          // set by launch_gemm: slices the K range is summed in (a function of K alone)
  int dbg;              // diagnostics only (SIDEKIT_AMD_GEMM_DBG, scripts/gemm_ablate.py): bit0 no operand prefetch after the first k-tile
  // optional fused tail of the embedding GEMM: when the slices run as blockIdx.z (small M) the kernel that adds them also L2-normalises
  // each finished row into l2_out [M][N] (the arithmetic of l2norm_kernel on the same values: same bits) and sets *l2_done = 1; otherwise
  // *l2_done = 0 and the caller launches the normalisation itself
  float* l2_out;
  int* l2_done;         // host
};
GemmArgs gemm_args();   // zero-initialised, alpha = 1
int launch_gemm(const GemmArgs& g, hipStream_t s);

// ---- frontend_fft.hip ------------------------------------------------------------------------
// |rFFT_1024(window * preemph(frame))|^2 for the log-mel front-end (n_fft 1024, win 400), one wave per frame
struct FftArgs {
  const void* wav; long wav_ld;                // [B][wav_ld] float32 samples, or int16 PCM (pcm16 != 0: widened as x / 32768 in the load)
  int pcm16;
  const int* nsamples; int nsamples_uniform;   // per-utterance sample counts (device) or one value for all
  int n_fft;                                   // 1024 (window 400, log-mel front-end) or 2048 (window 1024, MFCC front-end)
  const float* window;                         // [400] / [1024]
  const float* tw512;                          // exp(-2 pi i m / (n_fft/2)), m = 0..n_fft/2-1, interleaved (re, im): the complex transform's twiddles
  const float* tw1024;                         // exp(-2 pi i k / n_fft),     k = 0..n_fft/2: the real-FFT split's
  float* P; long ldp;                          // [M][ldp] power spectrum, ldp >= n_fft/2 + 1 (extra columns zeroed)
  int M, t_max, hop;
  const int* row_b; const int* row_t;          // optional ragged row map
  float preemph;
  // fused mel projection (mel_cw != nullptr): instead of P the kernel writes logmel[m][j] = log(sum_k fb[k][j] P[k] + 1e-6).
  // Every filter is a short run of bins (triangles: <= 31 of 513) cut into chunks of 8 taps: chunk c = bins [mel_ck0[c], + 8) with
  // weights mel_cw[c][0..8) (zero beyond the filter's run); filter j owns the consecutive chunks [cb, cb + nc), mel_fmeta[j] = cb | nc << 16;
  // mel_chunks is padded to a multiple of 64 with all-zero chunks
  const float* mel_cw; const int* mel_ck0; const int* mel_fmeta; int mel_chunks; int n_mels;
  float* logmel; long ldl;
};
int launch_stft_power_fft(const FftArgs& a, hipStream_t s);

// ---- pool.hip ---------------------------------------------------------------------------
// rows layout: x[(row0[b] + t) * ld + d], t < count[b]
struct RowSpan {
  const int* offsets;  // [B] device row offset per utterance, or null -> b * stride
  int stride;
  Lens lens;           // feature frames per utterance
  int halvings;        // count = halve(frames, halvings) - shrink
  int shrink;
  __device__ inline int row0(int b) const { return offsets ? offsets[b] : b * stride; }
  __device__ inline int count(int b) const { return halve(lens.get(b), halvings) - shrink; }
};
// out[b][0:D] = mean_t, out[b][D:2D] = unbiased std_t  (pooling.py:55-70)
int launch_mean_std(const void* x, int x_bf16, long ld, int D, RowSpan rs, float* out, int B, hipStream_t s);
// attentive statistics: w = softmax_t(e); mu = sum x w; rh = sqrt(clamp(sum x^2 w - mu^2, 1e-9)) (pooling.py:165-168)
int launch_att_stats(const void* x, int x_bf16, const float* e, long ld, int D, RowSpan rs, float* out, int B, hipStream_t s);
// bf16 path: attention.4 (h [rows][128] f32 x W2 [D][128] bf16 + b2), softmax over time and the weighted statistics in one kernel:
// the (rows x D) score matrix e is never written
int launch_att_fused(const void* x_bf16, const float* h, const void* w2_bf16, const float* b2, long ld, int D, RowSpan rs, float* out, int B,
                     hipStream_t s);
// per-(b, column) CMVN over rows: (x - mean) / sqrt(var_biased + eps), in place (InstanceNorm1d)
int launch_cmvn(float* x, long ld, int D, RowSpan rs, float eps, int B, hipStream_t s);
// x / ||x||_2 (loss.py:91-100) followed by F.normalize(eps=1e-12) (xvector.py:903)
int launch_l2norm(const float* x, float* out, int D, int B, hipStream_t s);
// x / max(||x||_2, eps) per row (F.normalize)
int launch_normalize_rows(const float* x, float* out, int D, int B, float eps, hipStream_t s);

}  // namespace sk
