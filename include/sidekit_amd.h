/* sidekit_amd C ABI -- MI355X (gfx950) x-vector extraction and trial scoring.
 *
 * The reference (deep-privacy/sidekit) is pure Python and has no FFI layer: the drop-in boundary
 * is its Python API.  This header is the C ABI that sits underneath the Python
 * mirror in `sidekit_amd/` -- plain pointers and sizes, no torch types.  Each entry point names
 * the reference interface it replaces (paths relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, a negative SK_E* class otherwise; the message is
 *     available from xt_last_error() (thread-local).  No exception crosses the ABI.
 *   - `d_` pointers are device (HIP) pointers owned by the caller, `h_` pointers are host memory.
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous on that stream.
 *   - a handle is bound to the device current at xt_create(); handles on different devices are independent (one process per GPU).
 *   - streams: consecutive calls on one handle may arrive on DIFFERENT streams.  The handle's workspaces are reused from call to call;
 *     the library orders that reuse itself: every call that touches a workspace (xt_forward*, xt_forward_begin, xt_forward_features,
 *     xt_features) records an event on the stream it was given, and a call that arrives on another stream first makes its stream wait for
 *     that event and for every batch still in flight on the handle's own streams.  (The caller still orders its own buffers: d_wav must be
 *     ready, and d_emb is complete, in the order of the stream passed.)
 *   - threads: one host thread at a time per handle.  A second thread that enters a handle while another is inside gets SK_ESTATE and
 *     nothing is enqueued; use one handle per driving thread (the reference, too, drives a model from one thread: SURVEY 8b).
 */
#ifndef SIDEKIT_AMD_H
#define SIDEKIT_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SK_OK 0
#define SK_EARG (-1)       /* bad argument            -> AssertionError / ValueError in the shim */
#define SK_ESHAPE (-2)     /* shape / key mismatch    -> RuntimeError (as torch load_state_dict) */
#define SK_EHIP (-3)       /* HIP runtime error       -> RuntimeError                            */
#define SK_EWORKSPACE (-4) /* batch exceeds xt_reserve -> RuntimeError                           */
#define SK_ESTATE (-5)     /* call order (e.g. forward before finalize), concurrent entry -> RuntimeError */

enum { XT_ARCH_HALFRESNET34 = 0, XT_ARCH_TDNN = 1 };
enum { XT_F32 = 0, XT_BF16 = 1, XT_F64 = 2, XT_I64 = 3, XT_I16 = 4 };
enum { XT_LOSS_AAM = 0, XT_LOSS_CCE = 1 };

typedef struct xt_handle xt_handle;

/* Mirrors the arguments of sidekit.nnet.xvector.Xtractor.__init__ (sidekit/nnet/xvector.py:424-431)
 * that matter at inference, plus the compute dtype of the trunk. */
typedef struct xt_config {
  int32_t arch;     /* XT_ARCH_*: model_archi "halfresnet34" (xvector.py:569-599) | "xvector" (:453-513) */
  int32_t dtype;    /* XT_F32 (parity path, exact-f32 MFMA) | XT_BF16 (bf16 MFMA, f32 accumulate)     */
  int32_t loss;     /* XT_LOSS_AAM | XT_LOSS_CCE                                                      */
  int32_t n_spk;    /* speaker_number                                                                 */
  int32_t emb_dim;  /* embedding_size (256)                                                           */
  float aam_s;      /* ArcMarginProduct scale: 30 (halfresnet34) / 64 (xvector)                       */
} xt_config;

/* Xtractor(...) : build an empty model on the current device. */
int xt_create(const xt_config* cfg, xt_handle** out);
int xt_destroy(xt_handle* h);

/* Xtractor.load_state_dict (sidekit/bin/extract_xvectors.py:86, strict=True): hand over one
 * checkpoint tensor by its reference key name (host memory, contiguous, row-major `shape`).
 * dtype: XT_F32, or XT_I64 for the `num_batches_tracked` counters.  Unknown keys and wrong shapes
 * fail with SK_ESHAPE. */
int xt_set_tensor(xt_handle* h, const char* key, const void* h_data, const int64_t* shape, int32_t ndim, int32_t dtype);
/* Number of keys / i-th key name the architecture expects (for strict checking in the shim). */
int xt_num_keys(xt_handle* h);
const char* xt_key_name(xt_handle* h, int32_t i);
/* Fold BatchNorm, repack to kernel layouts, upload.  Fails with SK_ESHAPE naming the first missing key. */
int xt_finalize(xt_handle* h);

/* Size the device workspace for a batch shape of max_batch utterances x max_samples samples each (TDNN: max_batch x
 * max_samples bounds the total).  Buffers only grow; a later forward runs when ONE reserved shape covers its batch in
 * both dimensions, so reserving (256, 64000) and (1, 20000000) sizes the workspace for those two products, not for
 * 256 x 20000000. */
int xt_reserve(xt_handle* h, int32_t max_batch, int64_t max_samples);

/* Xtractor.forward(x, is_eval=True) (sidekit/nnet/xvector.py:876-907).
 *   d_wav       float32 [B][wav_ld] waveform, utterance b uses its first h_nsamples[b] samples
 *   h_nsamples  NULL = every utterance has L samples
 *   d_emb       float32 [B][emb_dim]  L2-normalised x-vectors (tuple slot 1 of the reference)
 *   d_logits    float32 [B][n_spk] s*cos logits (tuple slot 0), or NULL to skip them */
int xt_forward(xt_handle* h, const float* d_wav, int64_t wav_ld, const int32_t* h_nsamples, int32_t B, int64_t L,
               float* d_emb, float* d_logits, void* stream);

/* Same, for 16-bit PCM as the files hold it: the decode of the reference driver (sidekit/bin/extract_xvectors.py:57-70,
 * soundfile.read -> float32 = int16 / 32768) happens inside the front-end kernel's load, so a wav file's payload goes
 * disk -> pinned memory -> device -> STFT without a conversion pass on either side.  x-vectors are bit-identical to
 * xt_forward on the widened samples.  d_pcm int16 [B][pcm_ld]. */
int xt_forward_pcm16(xt_handle* h, const int16_t* d_pcm, int64_t pcm_ld, const int32_t* h_nsamples, int32_t B, int64_t L,
                     float* d_emb, float* d_logits, void* stream);

/* Pipelined forwards: two WHOLE batches in flight instead of the two halves of one.  The reference driver's loop
 * (sidekit/bin/extract_xvectors.py:130-150) is one forward at a time; a corpus is many independent batches, and two of them half a step
 * apart use the chip better than one alone (one batch's HBM-bound first layer beside the other's MFMA-bound deep layers: 5.67 vs 5.87 ms
 * per batch of 256 on MI355X).  xt_reserve_slots sizes `slots` (<= 4; 2 is what pays) full workspaces, each with a stream the handle
 * owns.  xt_forward_begin queues the whole forward of a batch on slot `slot`'s stream -- behind everything queued on `stream` so far --
 * and returns without joining; in_dtype XT_F32 (d_wav float32) or XT_I16 (16-bit PCM as in xt_forward_pcm16).  xt_forward_end makes
 * `stream` wait for that slot's last forward: d_emb / d_logits are complete behind it.  A slot's forwards run in the order they were
 * begun; reusing a slot before its previous batch was ended is allowed only if the caller no longer needs that batch's outputs.
 * x-vectors are the bits xt_forward gives (tests/test_gpu_fullsize.py::test_pipelined_forwards_are_bit_identical). */
int xt_reserve_slots(xt_handle* h, int32_t slots, int32_t max_batch, int64_t max_samples);
int xt_forward_begin(xt_handle* h, int32_t slot, const void* d_wav, int32_t in_dtype, int64_t wav_ld, const int32_t* h_nsamples, int32_t B,
                     int64_t L, float* d_emb, float* d_logits, void* stream);
int xt_forward_end(xt_handle* h, int32_t slot, void* stream);

/* Same, entered after the front-end (everything after xvector.py:885): the features->embedding
 * seam the parity fixtures are cut at.  d_feats float32 (B, 80, T) as MelSpecFrontEnd / MfccFrontEnd
 * return it; h_frames NULL = all T. */
int xt_forward_features(xt_handle* h, const float* d_feats, const int32_t* h_frames, int32_t B, int32_t T,
                        float* d_emb, float* d_logits, void* stream);

/* Front-end only: MelSpecFrontEnd.forward(is_eval=True) (sidekit/nnet/preprocessor.py:267-285) or
 * MfccFrontEnd.forward (:113-124).  d_feats_out float32 (B, 80, T), T = 1 + L / hop. */
int xt_features(xt_handle* h, const float* d_wav, int64_t wav_ld, const int32_t* h_nsamples, int32_t B, int64_t L,
                float* d_feats_out, void* stream);

/* forward(..., norm_embedding=False) (xvector.py:876,893-898): only observable for loss='cce', whose
 * eval output is then the un-normalised linear6 output; the 'aam' branch always normalises (:903). */
int xt_set_norm_embedding(xt_handle* h, int32_t on);

/* Split forward.  With lanes = n (default 2, at most 4; SIDEKIT_AMD_LANES in the environment or xt_set_lanes) a HalfResNet34 batch
 * of >= 128 utterances is forwarded as up to n parts of at least 64 utterances on n HIP streams (the caller's and n - 1 the handle
 * owns): one part's latency-bound kernels run under another part's convolutions.  Results do not change (every kernel is batch-size
 * invariant).  lanes = 1 serialises the forward again, for profiles in which one kernel's duration has to mean something.  Measured on MI355X: two and three
 * lanes are 1-4 % faster than one, FOUR are slower (a constant +2.3 ms per forward: the fourth lane's stream shares a hardware queue).
 * The reference has no counterpart: its forward is one
 * stream of cuDNN calls (sidekit/nnet/xvector.py:876-907). */
int xt_set_lanes(xt_handle* h, int32_t lanes);
int xt_get_lanes(xt_handle* h);

/* Diagnostics for stage-wise parity tests: keep a device copy of intermediate activations of the
 * next forward ("feats", "stem", "layer1".."layer4", "pooled", "pre_norm", TDNN: "conv1".."conv5").
 * xt_debug_tap copies one to host (raw element type of the trunk: f32, or bf16 for XT_BF16 trunk
 * activations) and returns its byte size in *bytes. */
int xt_set_debug(xt_handle* h, int32_t on);
int xt_debug_tap(xt_handle* h, const char* name, void* h_dst, size_t capacity, size_t* bytes);

/* Measurement: when on, every kernel launch of the forward is bracketed by HIP events recorded on
 * the launch stream.  xt_get_profile synchronises and returns, per slot, the summed device time (ms)
 * and the number of launches since the last reset.  Slots 0..10 are the 3x3 / 1x1 trunk convolution
 * shapes in the order L1, L1-shortcut, L2a, L2-shortcut, L2, L3a, L3-shortcut, L3, L4a, L4-shortcut, L4.
 * `on`: 0 = off, 1 = every slot, otherwise a mask with bit (slot + 1) set for each slot to bracket -- an event pair
 * costs about 2 us of stream time, so a timed run brackets only the class it reports (bench.py: the dominant one). */
#define XT_PROF_FRONTEND 11
#define XT_PROF_STEM 12
#define XT_PROF_SE_RES 13
#define XT_PROF_POOL_TAIL 14
#define XT_PROF_TDNN 15
#define XT_PROF_PAIR_L1 16   /* conv2 of block k + conv1 of block k + 1 of layer 1 in one kernel (round 6; csrc/conv_pair.hip) */
#define XT_PROF_SLOTS 17
int xt_set_profile(xt_handle* h, int32_t on);
int xt_get_profile(xt_handle* h, double* ms /*[XT_PROF_SLOTS]*/, int64_t* launches /*[XT_PROF_SLOTS]*/, int32_t reset);

/* Tuning harness (diagnostic): mean device ms of `iters` launches of trunk convolution `shape` (slot order of
 * xt_get_profile) on a B x T batch; variant bit0 = no stores, bit1 = no MFMA loop, bit2 = no staging, bit3 = statistics epilogue, bit4 = residual epilogue,
 * bit5 = one workgroup per tile even for the persistent shapes, bit6 = print the runtime's occupancy for the shape; shapes 11..47 are the
 * A/B alternatives listed next to the product configurations in conv3x3.hip (A/B build of the library only); shape 48 = the layer-1 pair kernel
 * (csrc/conv_pair.hip: conv2 of block k + conv1 of block k + 1; variant bit0 = the first block's in-place shortcut form). */
int sk_bench_conv(int32_t shape, int32_t dtype, int32_t B, int32_t T, int32_t iters, int32_t variant, float* ms_out,
                  double* phase_cycles /* [8] mean shader cycles per kernel phase, or NULL */);

const char* xt_last_error(void);

/* Sample-rate conversion of one utterance on the device: `torchaudio.transforms.Resample(orig_freq, new_freq)` as
 * sidekit/bin/extract_xvectors.py:141-143 applies it to a file whose rate differs from the model's (torchaudio 0.8.2, un-vendored:
 * windowed-sinc interpolation, lowpass_filter_width 6, roll-off 0.99 -- restated from the published algorithm, parity unpinned).
 * d_in: n_in samples, XT_F32 or XT_I16 (widened as x / 32768); *n_out = ceil(new * n_in / orig) after reducing the rates by their
 * gcd; d_out = NULL only queries *n_out. */
int sk_resample(const void* d_in, int32_t in_dtype, int64_t n_in, int32_t orig_freq, int32_t new_freq, float* d_out,
                int64_t out_capacity, int64_t* n_out, void* stream);

/* ---- trial scoring ------------------------------------------------------------------------- */

/* sidekit.iv_scoring.cosine_scoring, the einsum of sidekit/iv_scoring.py:108-109: rows already
 * L2-normalised by the caller (StatServer.norm_stat1).  d_out[i][j] = <E_i, T_j>, float32. */
int sc_cosine(const float* d_E, int32_t Ne, const float* d_T, int32_t Nt, int32_t D, float* d_out, void* stream);

/* sidekit.iv_scoring.fast_PLDA_scoring, sidekit/iv_scoring.py:448-462:
 *   out[i][j] = scaling * ( 0.5 e_i' Phi e_i + 0.5 t_j' Phi t_j + cst + e_i' Psi t_j ),  float64.
 * Phi, Psi (D x D, row-major) and cst come from the 256x256 float64 algebra that stays on the
 * host (iv_scoring.py:428-446).  E, T are the centred (and optionally Vtrans-rotated) vectors. */
int sc_plda_fast(const double* d_E, int32_t Ne, const double* d_T, int32_t Nt, int32_t D, const double* d_Phi,
                 const double* d_Psi, double cst, double scaling, double* d_out, void* stream);

/* torch.nn.functional.normalize(x, dim=1) (eps 1e-12) on the device: what the reference applies to cohort and test x-vectors before
 * cosine scoring (sidekit/score_normalization.py:128, sidekit/nnet/xvector.py:243,258-259).  d_out may alias d_X. */
int sc_normalize_rows(const float* d_X, int32_t N, int32_t D, float* d_out, void* stream);

/* sc_plda_fast keeps its intermediate buffer (E.Psi and the quadratic-form partials) cached per (device, stream) so that a call
 * allocates nothing; this frees every cached buffer (after a device synchronise).  The Python shim calls it at interpreter exit;
 * a long-lived host that creates and destroys many streams may call it whenever no sc_plda_fast call is in flight.  The reference
 * has no counterpart (its temporaries are numpy arrays, sidekit/iv_scoring.py:449-460). */
int sc_release_workspace(void);

/* All-pairs cosine scoring without the score matrix (100k x 100k trials = 40 GB of float32): the scores of
 * sc_cosine are classified target (labels_e[i] == labels_t[j]) / non-target and counted into two histograms of `nbins`
 * (= 8192) equal bins over [lo, hi) (out-of-range scores land in the end bins); self_offset >= 0 drops the trials
 * j == i + self_offset, i.e. the self-trials when E is rows [self_offset, self_offset + Ne) of T (a set, or one rank's row
 * shard of it, scored against itself: the trials sidekit/nnet/xvector.py:240-262 masks out through its Ndx); < 0 keeps all.
 * The EER of the binned scores follows on the host (sidekit_amd.bosaris.detplot.eer_from_histograms). */
int sc_cosine_hist(const float* d_E, int32_t Ne, const float* d_T, int32_t Nt, int32_t D, const int32_t* d_labels_e,
                   const int32_t* d_labels_t, int32_t self_offset, float lo, float hi, int32_t nbins, uint64_t* d_hist_tar,
                   uint64_t* d_hist_non, void* stream);

/* Speaker-mean enrolment + cosine over a listed trial set, sidekit/bin/compute_spk_cosine.py:18-26:
 * out[k] = <E[enr_idx[k]], T[tst_idx[k]]> / (|E| |T|), float32 in, float64 maths. */
int sc_cosine_trials(const float* d_E, const float* d_T, int32_t D, const int32_t* d_enr_idx, const int32_t* d_tst_idx,
                     int64_t n_trials, double* d_out, void* stream);

/* Adaptive symmetric score normalisation, sidekit.score_normalization.asnorm (sidekit/score_normalization.py:120-140).
 * sc_topk_stats: per row of a (n_rows x n_cols) float32 score matrix, mean and unbiased std of its k largest values
 * (the reference's `topk(200, dim=1)` + mean/std).  sc_snorm_apply: S[i][j] <- 0.5 ((S - mean_e[i]) / std_e[i] +
 * (S - mean_t[j]) / std_t[j]) in place. */
int sc_topk_stats(const float* d_scores, int32_t n_rows, int32_t n_cols, int32_t k, float* d_mean, float* d_std, void* stream);
int sc_snorm_apply(float* d_S, int32_t Ne, int32_t Nt, const float* d_mean_e, const float* d_std_e, const float* d_mean_t,
                   const float* d_std_t, void* stream);

/* ---- EER support (host code, no GPU needed) --------------------------------------------------- */

/* sidekit.bosaris.detplot.pavx (sidekit/bosaris/detplot.py:289-351): isotonic (non-decreasing) fit of y.
 * width / height need room for n entries; *nbins receives the number of bins.  ghat_out (n) may be NULL. */
int sk_pavx(const double* y, int64_t n, double* ghat_out, int64_t* width, double* height, int64_t* nbins);

/* Vertex walk of sidekit.bosaris.detplot.rocch (detplot.py:414-434): pideal is the 1/0 target indicator
 * ordered by ascending score (stable sort), width the PAV bins; pmiss / pfa receive nbins + 1 vertices. */
int sk_rocch_vertices(const double* pideal, int64_t n, int64_t n_tar, int64_t n_non, const int64_t* width, int64_t nbins,
                      double* pmiss, double* pfa);

/* ---- wav staging for the streaming extractor (host code, no GPU needed) ---------------------------
 * Replaces the per-file `torchaudio.load` of sidekit/bin/extract_xvectors.py:57-70 for canonical files: a pool of `threads`
 * host threads (no interpreter lock) walks the RIFF headers / reads the samples.
 * sk_wav_probe: kind[i] = 1 for RIFF/WAVE PCM 16-bit mono (nsamples / rate / data_offset filled), 0 for any other wav (the
 *   caller decodes it), -1 if the file cannot be opened.
 * sk_wav_read_pcm16: file i's samples -> dst[row[i] * ld .. + nsamples[i]) (dst: a pinned int16 staging buffer of n_rows rows, ld in
 *   elements); status[i] = 0 on success, -1 on a short read / open failure / nsamples[i] > ld / row[i] outside [0, n_rows). */
int sk_wav_probe(const char* const* paths, int32_t n, int32_t threads, int32_t* nsamples, int32_t* rate, int64_t* data_offset,
                 int32_t* kind);
int sk_wav_read_pcm16(const char* const* paths, const int64_t* data_offset, const int32_t* nsamples, const int32_t* row, int32_t n,
                      int32_t threads, int16_t* dst, int64_t ld, int32_t n_rows, int32_t* status);

#ifdef __cplusplus
}
#endif
#endif /* SIDEKIT_AMD_H */
