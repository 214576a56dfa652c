// C ABI of the extractor: model handle, checkpoint ingestion (BatchNorm folding + repacking to the
// kernels' layouts), workspace, and the launch sequence of Xtractor.forward(is_eval=True)
// (sidekit/nnet/xvector.py:876-907) for the two architectures in scope.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "../../include/sidekit_amd.h"
#include <stdlib.h>
#include "kernels.h"

namespace sk {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* last_error() { return g_err; }

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool set = false;
  size_t numel() const { size_t n = 1; for (auto s : shape) n *= (size_t)s; return n; }
};

struct FrontCfg { int n_fft, win, hop, n_mels, n_out; double f_min, f_max; };
static const FrontCfg MELSPEC = {1024, 400, 160, 80, 80, 90.0, 7600.0};      // xvector.py:570-573, preprocessor.py:216-226
static const FrontCfg MFCCCFG = {2048, 1024, 512, 100, 80, 133.333, 6855.4976};  // preprocessor.py:65-76

struct ConvLayer {
  int shape;
  void* wpack = nullptr;
  float* scale = nullptr;
  float* shift = nullptr;
  ConvGeom g;
};
struct Block {
  ConvLayer c1, c2, sc;
  bool has_sc = false;
  float* se_w1 = nullptr;
  float* se_w2 = nullptr;
  void* w2t = nullptr;   // conv2 weights as the MFMA consumes them, [tap][ci][co] bf16 / f32 (SE gate pre-computation)
  void* sc_wfold = nullptr;   // shortcut 1x1 weights with the shortcut BN scale folded in (row co times scale[co]), fragment order
  int C, li;
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t n) {
    if (n <= bytes) return SK_OK;
    // forwards are asynchronous: a batch queued earlier may still be reading this buffer when a longer batch makes it grow
    if (p) { SK_HIP(hipDeviceSynchronize()); SK_HIP(hipFree(p)); }
    p = nullptr; bytes = 0;
    SK_HIP(hipMalloc(&p, n));
    bytes = n;
    return SK_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};


// One lane of the forward: its own activation / statistics workspace and pinned staging ring for per-utterance integers.
struct Lane {
  DevBuf ws_S, ws_feat, ws_act[4], ws_se, ws_col, ws_edge, ws_splitk, ws_gate, ws_ctx, ws_rb, ws_h, ws_e, ws_pooled, ws_pre, ws_int;
  static constexpr int RING = 4;
  int* ring_host[RING] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ring_ev[RING];
  bool ring_used[RING] = {false, false, false, false};
  size_t ring_bytes = 0;
  int ring_cur = 0;
  hipStream_t stream = nullptr;        // lanes 1 .. n-1 only: the stream that part of the batch runs on
  hipEvent_t fork = nullptr, join = nullptr;
  int persist_cap = 0;                 // set by xt_forward_begin: persistent convolution grids leave room for the other batch in flight (conv3x3.hip, launch_cfg)
  std::vector<std::pair<int, int64_t>> reserved;   // side lanes: the (utterances, samples) shapes this lane's workspace has been sized for
  bool covers(int B, int64_t L) const {
    for (auto& r : reserved) if (r.first >= B && r.second >= L) return true;
    return false;
  }
};

static int lanes_from_env() {
  const char* e = getenv("SIDEKIT_AMD_LANES");
  const int n = e ? atoi(e) : 2;
  return n < 1 ? 1 : (n > 4 ? 4 : n);
}

// parts a batch of B utterances is forwarded in: at most `lanes`, each of at least LANE_MIN utterances
static int lane_parts(int lanes, int B, int lane_min) {
  int n = B / lane_min;
  n = n < 1 ? 1 : n;
  return n < lanes ? n : lanes;
}

}  // namespace sk

using namespace sk;

struct xt_handle {
  bool shortcut_tensor = SK_AB_GETENV("SIDEKIT_AMD_SHORTCUT_TENSOR") != nullptr;   // A/B builds only (common.h), see half_from_feats
  bool mel_gemm = SK_AB_GETENV("SIDEKIT_AMD_MEL_GEMM") != nullptr;                 // A/B switch: mel projection as a separate GEMM
  bool mfcc_dft_gemm = SK_AB_GETENV("SIDEKIT_AMD_MFCC_DFT_GEMM") != nullptr;       // A/B switch: MFCC spectrum as a DFT contraction (round-1 form) instead of the FFT
  // SE gate computed inside conv2 instead of by a launch of its own (se_gate_inl.h).  Two forms were built in round 5; both give se_pre_kernel's
  // bits (tests/test_gpu_halfresnet.py) and NEITHER pays (profiles/r05_latency_matrix.txt), so the default is 0 = the launch:
  //  - prologue (3 = small grids, 4 = always; the round-4 verdict's specification: a separate instantiation in which every workgroup of an
  //    utterance reduces the sums in se_pre_kernel's order before its k-loop): the gate is a chain of dependent L2 round trips and barriers
  //    that takes 5-6 us whether it runs as a kernel (5.7-7.3 us) or at the head of conv2 (+5.4-6.3 us per conv2, +23 us for layer 4):
  //    0.690 vs 0.687 ms per 4-s utterance on one box, 0.663 vs 0.618 on another;
  //  - gate wave (1 = small grids, 2 = always; layers 1-2): a fifth wave computes the gate WHILE the four convolution waves stage the tile and
  //    run the k-loop, joining their barriers.  Concurrent, no launch -- but ONE wave walks the chain in ~10 us where the 16-wave kernel
  //    needs 1-2, longer than the convolution it hides behind: 0.636 vs 0.618 ms.
  int gate_prologue = SK_AB_ENV_INT("SIDEKIT_AMD_GATE_PROLOGUE", 0);   // A/B builds only: the product has no in-convolution gate form
  // small-grid tilings for conv2 of layers 3-4 (conv3x3.hip, "Small-grid forms"): 1 (default) at most SMALL_GRID_MAX_B utterances, 0 never, 2 always
  static constexpr int SMALL_GRID_MAX_B = 12;      // crossover with the product tilings between 12 and 16 utterances (profiles/r05_small_grid_sweep.txt)
  static constexpr int GATE_AB_MAX_B = 8;          // A/B builds: the in-convolution gate forms keep their own bound (SIDEKIT_AMD_GATE_PROLOGUE = 1 / 3)
  int small_grid = getenv("SIDEKIT_AMD_SMALL_GRID") ? atoi(getenv("SIDEKIT_AMD_SMALL_GRID")) : 1;
  xt_config cfg;
  int device = 0;
  bool finalized = false;
  std::vector<std::string> keys;               // expected checkpoint keys, reference order
  std::map<std::string, HostTensor> tensors;   // host copies until finalize
  std::vector<void*> dev_allocs;               // weight blobs

  FrontCfg fc;
  int nbp = 0;             // DFT bins padded to a multiple of 4
  float* d_window = nullptr;
  float* d_basis = nullptr;  // [2*nbp][win]
  float* d_fbT = nullptr;    // [n_mels][nbp]
  float* d_mel_cw = nullptr; int* d_mel_ck0 = nullptr; int* d_mel_fmeta = nullptr; int mel_chunks = 0; bool mel_fused = false;   // bank as 8-tap chunks (frontend_fft.hip)
  float* d_dctT = nullptr;   // [n_out][n_mels] (MFCC)
  float* d_tw512 = nullptr;  // FFT twiddles: the n_fft/2-point complex transform's ...
  float* d_tw1024 = nullptr; // ... and the real-FFT split's

  // halfresnet34
  void* d_zeros = nullptr;
  float* stem_w = nullptr; float* stem_shift = nullptr;   // stem weights carry the BatchNorm scale
  std::vector<Block> blocks;
  float *att_w1x = nullptr, *att_w1c = nullptr, *att_b1 = nullptr, *att_bn_scale = nullptr, *att_bn_shift = nullptr;
  float *att_w2 = nullptr, *att_b2 = nullptr;
  void *att_w1x_bf16 = nullptr, *att_w2_bf16 = nullptr;   // bf16 copies for the bf16 compute path
  float *emb_w = nullptr, *emb_scale = nullptr, *emb_shift = nullptr, *emb_bias = nullptr;
  float* head_wn = nullptr;  // row-normalised ArcMargin weight
  // tdnn
  struct TdnnLayer { float *w, *bias, *scale, *shift; int cin, cout, k, dil; };
  std::vector<TdnnLayer> tdnn;

  // workspace: lanes.  Lane 0 runs on the caller's stream and is sized for the whole batch; lanes 1 .. n-1 (own streams, sized for a
  // part of a batch) exist only while the split forward is on: a batch of >= 2 LANE_MIN utterances is then forwarded as n parts on n
  // HIP streams, so that one part's latency-bound kernels (SE gates, pooling, front-end: 0.8 ms of a 6.5-ms step) run under another
  // part's convolutions.  Utterances are independent and every kernel is batch-size invariant, so the x-vectors are the same bits.
  std::vector<std::pair<int, int64_t>> reserved;   // (batch, samples) shapes xt_reserve has sized the workspace for: a batch runs when one of them covers it in BOTH dimensions
  static constexpr int MAX_LANES = 4;
  Lane lane[MAX_LANES];
  int lanes = lanes_from_env();                    // 1: serial (profiling: per-kernel durations mean something), 2 (default) .. 4: that many parts of a batch side by side
  static constexpr int LANE_MIN = 64;              // utterances per lane below which a batch is not split further
  bool norm_embedding = true;
  // The stream contract, enforced (round 5).  Workspaces are ordered by STREAM order: an unsplit forward runs on the caller's stream in lane
  // 0's workspace, a split or pipelined one on streams the handle owns behind an event recorded on the caller's stream.  All of that assumes
  // consecutive calls arrive on ONE stream; round 4's soak test found the first hole in it (a plain forward racing a pipelined batch in slot
  // 0).  Now every call that touches a workspace ends by recording `tail` on the stream it was given, and a call that arrives on ANOTHER
  // stream first makes that stream wait for `tail` and for every lane's completion event -- the library orders itself, whatever stream a
  // call comes from.  Two host threads inside one handle at once are refused (SK_ESTATE): the host-side bookkeeping (pinned rings, lane
  // state, reserved shapes) has no lock and is not meant to have one (one handle, one driving thread).
  std::atomic<bool> busy{false};
  hipStream_t last_stream = nullptr;
  bool has_last = false;
  hipEvent_t tail = nullptr;
  // per-kernel-class HIP-event profile (xt_set_profile)
  uint32_t profile = 0;   // bit (slot + 1) per bracketed slot; 1 = all
  struct ProfRec { int slot; hipEvent_t a, b; };
  std::vector<ProfRec> prof_recs;
  std::vector<hipEvent_t> prof_pool;
  double prof_ms[XT_PROF_SLOTS] = {0};
  int64_t prof_n[XT_PROF_SLOTS] = {0};
  // debug taps
  bool debug = false;
  struct Tap { DevBuf buf; size_t bytes = 0; };
  std::map<std::string, Tap> taps;
};

namespace sk {

// One host thread at a time per handle; see xt_handle::busy.
struct EntryGuard {
  xt_handle* h; bool ok;
  explicit EntryGuard(xt_handle* h_) : h(h_), ok(h_ && !h_->busy.exchange(true, std::memory_order_acquire)) {}
  ~EntryGuard() { if (ok) h->busy.store(false, std::memory_order_release); }
};
#define SK_ENTER(h)                                                                                                   \
  SK_CHECK((h) != nullptr, SK_EARG, "null handle");                                                                   \
  sk::EntryGuard guard_(h);                                                                                           \
  SK_CHECK(guard_.ok, SK_ESTATE, "concurrent entry: another host thread is inside this handle (a handle is driven by one thread at a time; use one handle per thread)")

// A call that touches a workspace and arrives on another stream than the previous one: order it behind the previous stream's tail and behind
// whatever the handle's own streams still run (pipelined batches, xt_forward_begin).
static int enter_stream(xt_handle* h, hipStream_t st) {
  if (h->has_last && h->last_stream != st) {
    SK_HIP(hipStreamWaitEvent(st, h->tail, 0));
    for (int k = 0; k < xt_handle::MAX_LANES; ++k)
      if (h->lane[k].stream) SK_HIP(hipStreamWaitEvent(st, h->lane[k].join, 0));
  }
  return SK_OK;
}
// ... and leaves its mark on the stream it ran on (also after an error: whatever was queued before the error is what the next stream must wait for).
static void leave_stream(xt_handle* h, hipStream_t st) {
  if (!h->tail && hipEventCreateWithFlags(&h->tail, hipEventDisableTiming) != hipSuccess) { h->tail = nullptr; return; }
  if (hipEventRecord(h->tail, st) == hipSuccess) { h->last_stream = st; h->has_last = true; }
}
struct StreamScope {
  xt_handle* h; hipStream_t st;
  StreamScope(xt_handle* h_, hipStream_t st_) : h(h_), st(st_) {}
  ~StreamScope() { leave_stream(h, st); }
};

// ---- expected checkpoint keys ------------------------------------------------------------------
static void add_bn_keys(std::vector<std::pair<std::string, std::vector<int64_t>>>& k, const std::string& p, int64_t c) {
  k.push_back({p + ".weight", {c}});
  k.push_back({p + ".bias", {c}});
  k.push_back({p + ".running_mean", {c}});
  k.push_back({p + ".running_var", {c}});
  k.push_back({p + ".num_batches_tracked", {}});
}

static const int HALF_PLANES[4] = {32, 64, 128, 256};
static const int HALF_BLOCKS[4] = {3, 4, 6, 3};

typedef std::vector<std::pair<std::string, std::vector<int64_t>>> KeyList;

static KeyList expected_keys(const xt_config& c) {
  KeyList k;
  const int64_t E = c.emb_dim, S = c.n_spk;
  if (c.arch == XT_ARCH_HALFRESNET34) {
    k.push_back({"preprocessor.PreEmphasis.flipped_filter", {1, 1, 2}});
    k.push_back({"preprocessor.MelSpec.spectrogram.window", {400}});
    k.push_back({"preprocessor.MelSpec.mel_scale.fb", {513, 80}});
    const std::string sn = "sequence_network";
    k.push_back({sn + ".conv1.weight", {32, 1, 3, 3}});
    add_bn_keys(k, sn + ".bn1", 32);
    int in_planes = 32;
    for (int li = 0; li < 4; ++li) {
      const int64_t pl = HALF_PLANES[li];
      for (int bi = 0; bi < HALF_BLOCKS[li]; ++bi) {
        const std::string p = sn + ".layer" + std::to_string(li + 1) + "." + std::to_string(bi);
        k.push_back({p + ".conv1.weight", {pl, in_planes, 3, 3}});
        add_bn_keys(k, p + ".bn1", pl);
        k.push_back({p + ".conv2.weight", {pl, pl, 3, 3}});
        add_bn_keys(k, p + ".bn2", pl);
        k.push_back({p + ".se.fc.0.weight", {pl / 16, pl}});
        k.push_back({p + ".se.fc.2.weight", {pl, pl / 16}});
        if (bi == 0) {  // res_net.py:302 with the tuple stride of :518: every layer's first block
          k.push_back({p + ".shortcut.0.weight", {pl, in_planes, 1, 1}});
          add_bn_keys(k, p + ".shortcut.1", pl);
        }
        in_planes = (int)pl;
      }
    }
    k.push_back({"before_speaker_embedding.lin_be.weight", {E, 5120}});
    add_bn_keys(k, "before_speaker_embedding.bn_be", E);
    k.push_back({"stat_pooling.attention.0.weight", {128, 7680, 1}});
    k.push_back({"stat_pooling.attention.0.bias", {128}});
    add_bn_keys(k, "stat_pooling.attention.2", 128);
    k.push_back({"stat_pooling.attention.4.weight", {2560, 128, 1}});
    k.push_back({"stat_pooling.attention.4.bias", {2560}});
    k.push_back({"after_speaker_embedding.weight", {S, E}});
  } else {
    k.push_back({"preprocessor.PreEmphasis.flipped_filter", {1, 1, 2}});
    k.push_back({"preprocessor.MFCC.dct_mat", {100, 80}});
    k.push_back({"preprocessor.MFCC.MelSpectrogram.spectrogram.window", {1024}});
    k.push_back({"preprocessor.MFCC.MelSpectrogram.mel_scale.fb", {1025, 100}});
    const int cin[5] = {80, 512, 512, 512, 512}, cout[5] = {512, 512, 512, 512, 1536}, ks[5] = {5, 3, 3, 1, 1};
    for (int i = 0; i < 5; ++i) {
      const std::string n = std::to_string(i + 1);
      k.push_back({"sequence_network.conv" + n + ".weight", {cout[i], cin[i], ks[i]}});
      k.push_back({"sequence_network.conv" + n + ".bias", {cout[i]}});
      add_bn_keys(k, "sequence_network.batch_norm" + n, cout[i]);
    }
    k.push_back({"before_speaker_embedding.linear6.weight", {E, 3072}});
    k.push_back({"before_speaker_embedding.linear6.bias", {E}});
    if (c.loss == XT_LOSS_AAM) {
      k.push_back({"after_speaker_embedding.weight", {S, E}});
    } else {  // xvector.py:499-507 (training-only head; held, never run at eval)
      add_bn_keys(k, "after_speaker_embedding.batch_norm6", 512);
      k.push_back({"after_speaker_embedding.linear7.weight", {512, 512}});
      k.push_back({"after_speaker_embedding.linear7.bias", {512}});
      add_bn_keys(k, "after_speaker_embedding.batch_norm7", 512);
      k.push_back({"after_speaker_embedding.linear8.weight", {S, 512}});
      k.push_back({"after_speaker_embedding.linear8.bias", {S}});
    }
  }
  return k;
}

// ---- uploads ----------------------------------------------------------------------------------
static int upload(xt_handle* h, const void* src, size_t bytes, void** dst) {
  void* p = nullptr;
  SK_HIP(hipMalloc(&p, bytes ? bytes : 16));
  if (bytes) SK_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
  h->dev_allocs.push_back(p);
  *dst = p;
  return SK_OK;
}
static int upload_f(xt_handle* h, const std::vector<float>& v, float** dst) { return upload(h, v.data(), v.size() * 4, (void**)dst); }

static const std::vector<float>& T(xt_handle* h, const std::string& key) { return h->tensors[key].data; }

static void fold_bn(xt_handle* h, const std::string& p, std::vector<float>& scale, std::vector<float>& shift) {
  const auto &g = T(h, p + ".weight"), &b = T(h, p + ".bias"), &m = T(h, p + ".running_mean"), &v = T(h, p + ".running_var");
  const size_t n = g.size();
  scale.resize(n); shift.resize(n);
  for (size_t i = 0; i < n; ++i) {
    const float s = g[i] / sqrtf(v[i] + 1e-5f);  // eval-mode BatchNorm, eps 1e-5
    scale[i] = s;
    shift[i] = b[i] - m[i] * s;
  }
}

static int make_conv(xt_handle* h, ConvLayer& L, int shape, const std::string& wkey, const std::string& bnkey) {
  L.shape = shape;
  SK_TRY(conv_geom(shape, h->cfg.dtype == XT_BF16 ? DT_BF16 : DT_F32, &L.g));
  const HostTensor& w = h->tensors[wkey];
  const int khkw = (int)(w.shape[2] * w.shape[3]);
  SK_CHECK(w.shape[0] == L.g.cout && w.shape[1] == L.g.cin && khkw == L.g.taps, SK_ESHAPE, "conv %s: weight/shape table mismatch", wkey.c_str());
  std::vector<unsigned char> packed(conv_pack_bytes(L.g));
  conv_pack_weights(L.g, w.data.data(), khkw, packed.data());
  SK_TRY(upload(h, packed.data(), packed.size(), &L.wpack));
  std::vector<float> sc, sh;
  fold_bn(h, bnkey, sc, sh);
  SK_TRY(upload_f(h, sc, &L.scale));
  SK_TRY(upload_f(h, sh, &L.shift));
  return SK_OK;
}

static int build_frontend(xt_handle* h) {
  const FrontCfg& f = h->fc;
  const int nb = f.n_fft / 2 + 1;
  h->nbp = (nb + 3) / 4 * 4;
  const bool mel = h->cfg.arch == XT_ARCH_HALFRESNET34;
  const std::string wk = mel ? "preprocessor.MelSpec.spectrogram.window" : "preprocessor.MFCC.MelSpectrogram.spectrogram.window";
  const std::string fk = mel ? "preprocessor.MelSpec.mel_scale.fb" : "preprocessor.MFCC.MelSpectrogram.mel_scale.fb";
  SK_TRY(upload_f(h, T(h, wk), &h->d_window));
  // real-DFT basis restricted to the window support: frame sample k sits at n = left + k of the n_fft frame
  const int left = (f.n_fft - f.win) / 2;
  std::vector<float> basis((size_t)2 * h->nbp * f.win, 0.f);
  for (int j = 0; j < nb; ++j)
    for (int k = 0; k < f.win; ++k) {
      const long ph = ((long)j * (left + k)) % f.n_fft;
      const double ang = 2.0 * M_PI * (double)ph / (double)f.n_fft;
      basis[(size_t)j * f.win + k] = (float)cos(ang);
      basis[((size_t)h->nbp + j) * f.win + k] = (float)(-sin(ang));
    }
  SK_TRY(upload_f(h, basis, &h->d_basis));
  const auto& fb = T(h, fk);  // [nb][n_mels]
  std::vector<float> fbT((size_t)f.n_mels * h->nbp, 0.f);
  for (int j = 0; j < nb; ++j)
    for (int m = 0; m < f.n_mels; ++m) fbT[(size_t)m * h->nbp + j] = fb[(size_t)j * f.n_mels + m];
  SK_TRY(upload_f(h, fbT, &h->d_fbT));
  {  // the bank for the projection fused into the FFT kernel: filter m = bins [st, st + ln) cut into 8-tap chunks (kernels.h, FftArgs)
    std::vector<int> st(f.n_mels, 0), ln(f.n_mels, 0);
    int maxlen = 0;
    for (int m = 0; m < f.n_mels; ++m) {
      int lo = nb, hi = -1;
      for (int j = 0; j < nb; ++j)
        if (fb[(size_t)j * f.n_mels + m] != 0.f) { lo = j < lo ? j : lo; hi = j; }
      if (hi >= lo) { st[m] = lo; ln[m] = hi - lo + 1; }
      maxlen = ln[m] > maxlen ? ln[m] : maxlen;
    }
    if (maxlen > 0 && maxlen <= 64) {   // a dense or very wide bank stays on the GEMM path
      std::vector<float> cw; std::vector<int> ck0, fmeta(f.n_mels, 0);
      for (int m = 0; m < f.n_mels; ++m) {
        const int nc = ln[m] > 0 ? (ln[m] + 7) / 8 : 1, cb = (int)ck0.size();
        fmeta[m] = cb | (nc << 16);
        for (int q = 0; q < nc; ++q) {
          ck0.push_back(ln[m] > 0 ? st[m] + 8 * q : 0);
          for (int i = 8 * q; i < 8 * q + 8; ++i) cw.push_back(i < ln[m] ? fb[(size_t)(st[m] + i) * f.n_mels + m] : 0.f);
        }
      }
      while (ck0.size() % 64) { ck0.push_back(0); cw.insert(cw.end(), 8, 0.f); }
      h->mel_chunks = (int)ck0.size();
      // the power row (n_fft / 2 + 8 floats, rounded up) and the chunk sums live in the transform's LDS buffer (2 * (n_fft / 2 + n_fft / 16) floats)
      h->mel_fused = h->mel_chunks + 8 <= 2 * (f.n_fft / 2 + f.n_fft / 16) - (f.n_fft / 2 + 16) && h->mel_chunks < 65536;
      if (h->mel_fused) {
        SK_TRY(upload_f(h, cw, &h->d_mel_cw));
        SK_TRY(upload(h, ck0.data(), ck0.size() * sizeof(int), (void**)&h->d_mel_ck0));
        SK_TRY(upload(h, fmeta.data(), fmeta.size() * sizeof(int), (void**)&h->d_mel_fmeta));
      }
    }
  }
  {  // twiddles of the n_fft-point real FFT (frontend_fft.hip), rounded once from double: the n_fft/2-point complex transform's
     // and the real-FFT split's
    SK_CHECK((f.n_fft == 1024 && f.win == 400) || (f.n_fft == 2048 && f.win == 1024), SK_EARG, "front-end FFT: n_fft %d / window %d unsupported", f.n_fft, f.win);
    const int nh = f.n_fft / 2;
    std::vector<float> tc(2 * (size_t)nh), ts(2 * (size_t)(nh + 1));
    for (int m = 0; m < nh; ++m) { tc[2 * m] = (float)cos(2.0 * M_PI * m / nh); tc[2 * m + 1] = (float)(-sin(2.0 * M_PI * m / nh)); }
    for (int k = 0; k <= nh; ++k) { ts[2 * k] = (float)cos(2.0 * M_PI * k / f.n_fft); ts[2 * k + 1] = (float)(-sin(2.0 * M_PI * k / f.n_fft)); }
    SK_TRY(upload_f(h, tc, &h->d_tw512));
    SK_TRY(upload_f(h, ts, &h->d_tw1024));
  }
  if (!mel) {
    const auto& dct = T(h, "preprocessor.MFCC.dct_mat");  // [n_mels][n_out]
    std::vector<float> dT((size_t)f.n_out * f.n_mels);
    for (int m = 0; m < f.n_mels; ++m)
      for (int o = 0; o < f.n_out; ++o) dT[(size_t)o * f.n_mels + m] = dct[(size_t)m * f.n_out + o];
    SK_TRY(upload_f(h, dT, &h->d_dctT));
  }
  return SK_OK;
}

static std::vector<float> normalize_rows(const std::vector<float>& w, int rows, int cols) {
  std::vector<float> o(w.size());
  for (int r = 0; r < rows; ++r) {
    double s = 0;
    for (int c = 0; c < cols; ++c) s += (double)w[(size_t)r * cols + c] * w[(size_t)r * cols + c];
    const float n = fmaxf((float)sqrt(s), 1e-12f);  // F.normalize eps
    for (int c = 0; c < cols; ++c) o[(size_t)r * cols + c] = w[(size_t)r * cols + c] / n;
  }
  return o;
}

static int finalize_half(xt_handle* h) {
  const std::string sn = "sequence_network";
  std::vector<float> zero64(64, 0.f);  // the zero page the conv staging reads its zero padding from
  SK_TRY(upload(h, zero64.data(), zero64.size() * 4, &h->d_zeros));
  std::vector<float> sc, sh;
  fold_bn(h, sn + ".bn1", sc, sh);
  {  // stem weights tap-major [9][32]: channel pairs are adjacent, which is what the packed-f32 FMAs of stem_kernel want.  The BatchNorm
     // scale is folded into them (f32 weights, f32 arithmetic: one more rounding per weight) and the shift starts the FMA chain, so the
     // kernel has no separate scale / shift step (64 of its 280 VALU instructions per position)
    const auto& w = T(h, sn + ".conv1.weight");   // [32][1][3][3]
    std::vector<float> wt(9 * 32);
    for (int c = 0; c < 32; ++c)
      for (int q = 0; q < 9; ++q) wt[q * 32 + c] = w[c * 9 + q] * sc[c];
    SK_TRY(upload_f(h, wt, &h->stem_w));
  }
  SK_TRY(upload_f(h, sh, &h->stem_shift));
  static const int first_shape[4] = {CONV_L1, CONV_L2A, CONV_L3A, CONV_L4A};
  static const int rest_shape[4] = {CONV_L1, CONV_L2, CONV_L3, CONV_L4};
  static const int sc_shape[4] = {CONV_L1S, CONV_L2S, CONV_L3S, CONV_L4S};
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < HALF_BLOCKS[li]; ++bi) {
      const std::string p = sn + ".layer" + std::to_string(li + 1) + "." + std::to_string(bi);
      Block b;
      b.C = HALF_PLANES[li]; b.li = li;
      SK_TRY(make_conv(h, b.c1, bi == 0 ? first_shape[li] : rest_shape[li], p + ".conv1.weight", p + ".bn1"));
      SK_TRY(make_conv(h, b.c2, rest_shape[li], p + ".conv2.weight", p + ".bn2"));
      b.has_sc = bi == 0;
      if (b.has_sc) {
        SK_TRY(make_conv(h, b.sc, sc_shape[li], p + ".shortcut.0.weight", p + ".shortcut.1"));
        // in-place shortcut (conv2's epilogue): bn_s(conv1x1(x)) = (scale_s * W) x + shift_s accumulates straight into
        // conv2's (already gate- and BN-scaled) accumulators, so the BN scale of the shortcut goes into its weights
        const HostTensor& w = h->tensors[p + ".shortcut.0.weight"];
        std::vector<float> scs, shs;
        fold_bn(h, p + ".shortcut.1", scs, shs);
        std::vector<float> wf(w.data.size());
        const size_t per = wf.size() / b.sc.g.cout;
        for (size_t i = 0; i < wf.size(); ++i) wf[i] = w.data[i] * scs[i / per];
        ConvGeom gs = b.sc.g;           // the fragments conv2's epilogue multiplies with: conv2's MFMA shape, not the stand-alone 1x1 kernel's
        gs.m16 = b.c2.g.m16;
        std::vector<unsigned char> packed(conv_pack_bytes(gs));
        conv_pack_weights(gs, wf.data(), (int)(w.shape[2] * w.shape[3]), packed.data());
        SK_TRY(upload(h, packed.data(), packed.size(), &b.sc_wfold));
      }
      {
        const auto& w2 = T(h, p + ".conv2.weight");  // [co][ci][3][3]
        const int Cb = b.C;
        std::vector<float> w2t((size_t)9 * Cb * Cb);
        for (int co = 0; co < Cb; ++co)
          for (int ci = 0; ci < Cb; ++ci)
            for (int t = 0; t < 9; ++t) w2t[((size_t)t * Cb + ci) * Cb + co] = w2[((size_t)co * Cb + ci) * 9 + t];
        if (h->cfg.dtype == XT_BF16) {  // exactly the rounded weights the conv multiplies with
          std::vector<uint16_t> hb(w2t.size());
          for (size_t i = 0; i < w2t.size(); ++i) hb[i] = f32_to_bf16(w2t[i]);
          SK_TRY(upload(h, hb.data(), hb.size() * 2, &b.w2t));
        } else {
          float* d = nullptr;
          SK_TRY(upload_f(h, w2t, &d));
          b.w2t = d;
        }
      }
      SK_TRY(upload_f(h, T(h, p + ".se.fc.0.weight"), &b.se_w1));
      SK_TRY(upload_f(h, T(h, p + ".se.fc.2.weight"), &b.se_w2));
      h->blocks.push_back(b);
    }
  // attentive pooling: channel index of the reference is d = c*10 + f (pooling.py:156-160); the
  // trunk's NHWC rows are d' = f*256 + c.  Permute the weights once instead of the activations.
  const int C4 = 256, F4 = 10, D = C4 * F4;
  auto perm = [&](int dp) { const int f = dp / C4, c = dp % C4; return c * F4 + f; };
  const auto& w1 = T(h, "stat_pooling.attention.0.weight");  // [128][7680]
  std::vector<float> w1x((size_t)128 * D), w1c((size_t)128 * 2 * D);
  for (int o = 0; o < 128; ++o)
    for (int dp = 0; dp < D; ++dp) {
      const int d = perm(dp);
      w1x[(size_t)o * D + dp] = w1[(size_t)o * 3 * D + d];
      w1c[(size_t)o * 2 * D + dp] = w1[(size_t)o * 3 * D + D + d];
      w1c[(size_t)o * 2 * D + D + dp] = w1[(size_t)o * 3 * D + 2 * D + d];
    }
  SK_TRY(upload_f(h, w1x, &h->att_w1x));
  SK_TRY(upload_f(h, w1c, &h->att_w1c));
  SK_TRY(upload_f(h, T(h, "stat_pooling.attention.0.bias"), &h->att_b1));
  fold_bn(h, "stat_pooling.attention.2", sc, sh);
  SK_TRY(upload_f(h, sc, &h->att_bn_scale));
  SK_TRY(upload_f(h, sh, &h->att_bn_shift));
  const auto& w2 = T(h, "stat_pooling.attention.4.weight");  // [2560][128]
  const auto& b2 = T(h, "stat_pooling.attention.4.bias");
  std::vector<float> w2p((size_t)D * 128), b2p(D);
  for (int dp = 0; dp < D; ++dp) {
    const int d = perm(dp);
    memcpy(&w2p[(size_t)dp * 128], &w2[(size_t)d * 128], 128 * sizeof(float));
    b2p[dp] = b2[d];
  }
  SK_TRY(upload_f(h, w2p, &h->att_w2));
  if (h->cfg.dtype == XT_BF16) {
    std::vector<uint16_t> hb(w1x.size());
    for (size_t i = 0; i < w1x.size(); ++i) hb[i] = f32_to_bf16(w1x[i]);
    SK_TRY(upload(h, hb.data(), hb.size() * 2, &h->att_w1x_bf16));
    hb.resize(w2p.size());
    for (size_t i = 0; i < w2p.size(); ++i) hb[i] = f32_to_bf16(w2p[i]);
    SK_TRY(upload(h, hb.data(), hb.size() * 2, &h->att_w2_bf16));
  }
  SK_TRY(upload_f(h, b2p, &h->att_b2));
  const int E = h->cfg.emb_dim;
  const auto& lw = T(h, "before_speaker_embedding.lin_be.weight");  // [E][5120] = [mu(d) | rh(d)]
  std::vector<float> lwp((size_t)E * 2 * D);
  for (int o = 0; o < E; ++o)
    for (int dp = 0; dp < D; ++dp) {
      const int d = perm(dp);
      lwp[(size_t)o * 2 * D + dp] = lw[(size_t)o * 2 * D + d];
      lwp[(size_t)o * 2 * D + D + dp] = lw[(size_t)o * 2 * D + D + d];
    }
  SK_TRY(upload_f(h, lwp, &h->emb_w));
  fold_bn(h, "before_speaker_embedding.bn_be", sc, sh);
  SK_TRY(upload_f(h, sc, &h->emb_scale));
  SK_TRY(upload_f(h, sh, &h->emb_shift));
  SK_TRY(upload_f(h, normalize_rows(T(h, "after_speaker_embedding.weight"), h->cfg.n_spk, E), &h->head_wn));
  return SK_OK;
}

static int finalize_tdnn(xt_handle* h) {
  const int cin[5] = {80, 512, 512, 512, 512}, cout[5] = {512, 512, 512, 512, 1536}, ks[5] = {5, 3, 3, 1, 1}, dil[5] = {1, 2, 3, 1, 1};
  for (int i = 0; i < 5; ++i) {
    const std::string n = std::to_string(i + 1);
    const auto& w = T(h, "sequence_network.conv" + n + ".weight");  // [cout][cin][k]
    std::vector<float> wg((size_t)cout[i] * cin[i] * ks[i]);
    for (int o = 0; o < cout[i]; ++o)
      for (int c = 0; c < cin[i]; ++c)
        for (int j = 0; j < ks[i]; ++j) wg[((size_t)o * ks[i] + j) * cin[i] + c] = w[((size_t)o * cin[i] + c) * ks[i] + j];
    xt_handle::TdnnLayer L;
    L.cin = cin[i]; L.cout = cout[i]; L.k = ks[i]; L.dil = dil[i];
    SK_TRY(upload_f(h, wg, &L.w));
    SK_TRY(upload_f(h, T(h, "sequence_network.conv" + n + ".bias"), &L.bias));
    std::vector<float> sc, sh;
    fold_bn(h, "sequence_network.batch_norm" + n, sc, sh);
    SK_TRY(upload_f(h, sc, &L.scale));
    SK_TRY(upload_f(h, sh, &L.shift));
    h->tdnn.push_back(L);
  }
  SK_TRY(upload_f(h, T(h, "before_speaker_embedding.linear6.weight"), &h->emb_w));
  SK_TRY(upload_f(h, T(h, "before_speaker_embedding.linear6.bias"), &h->emb_bias));
  if (h->cfg.loss == XT_LOSS_AAM)
    SK_TRY(upload_f(h, normalize_rows(T(h, "after_speaker_embedding.weight"), h->cfg.n_spk, h->cfg.emb_dim), &h->head_wn));
  return SK_OK;
}

// ---- small layout kernels -----------------------------------------------------------------------
// feats (B, F, T) reference layout <-> row-major rows [row0(b) + t][F]
__global__ void bft_to_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int F, int T, RowSpan rs) {
  const int b = blockIdx.x, n = rs.count(b);
  const long r0 = rs.row0(b);
  for (int i = threadIdx.x; i < n * F; i += blockDim.x) {
    const int t = i / F, f = i % F;
    dst[(r0 + t) * F + f] = src[((long)b * F + f) * T + t];
  }
}
__global__ void rows_to_bft_kernel(const float* __restrict__ src, float* __restrict__ dst, int F, int T, RowSpan rs) {
  const int b = blockIdx.x, n = rs.count(b);
  const long r0 = rs.row0(b);
  for (int i = threadIdx.x; i < T * F; i += blockDim.x) {
    const int f = i / T, t = i % T;
    dst[((long)b * F + f) * T + t] = t < n ? src[(r0 + t) * F + f] : 0.f;
  }
}

// Brackets one launch with HIP events on the launch stream when profiling is on.
struct ProfScope {
  xt_handle* h; hipStream_t st; int slot; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(xt_handle* h_, int slot_, hipStream_t st_) : h(h_), st(st_), slot(slot_) {
    if (!(h->profile == 1u || (h->profile >> (slot_ + 1)) & 1u)) return;
    auto get = [&]() { hipEvent_t e; if (!h->prof_pool.empty()) { e = h->prof_pool.back(); h->prof_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    a = get(); b = get();
    (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (!a) return;
    (void)hipEventRecord(b, st);
    h->prof_recs.push_back({slot, a, b});
  }
};

static int tap(xt_handle* h, const char* name, const void* src, size_t bytes, hipStream_t st) {
  if (!h->debug) return SK_OK;
  auto& t = h->taps[name];
  SK_TRY(t.buf.ensure(bytes));
  t.bytes = bytes;
  SK_HIP(hipMemcpyAsync(t.buf.p, src, bytes, hipMemcpyDeviceToDevice, st));
  return SK_OK;
}

struct BatchMeta {
  int B = 0;
  int T = 0;               // max feature frames
  Lens lens{nullptr, 0};   // feature frames per utterance
  const int* d_nsamples = nullptr;
  int nsamples_uniform = 0;
  // ragged (TDNN)
  int R = 0;               // total rows
  const int* d_offsets = nullptr;
  const int* d_row_b = nullptr;
  const int* d_row_t = nullptr;
};

// Per-utterance integers travel host -> device through a small ring of pinned staging slots, each
// guarded by an event, so the call stays asynchronous and the source never goes out of scope.
static int ring_begin(Lane& ln) {
  ln.ring_cur = (ln.ring_cur + 1) % Lane::RING;
  if (ln.ring_used[ln.ring_cur]) SK_HIP(hipEventSynchronize(ln.ring_ev[ln.ring_cur]));
  return SK_OK;
}
static int push_ints(Lane& ln, const std::vector<int>& v, size_t slot_off, const int** dptr, hipStream_t st) {
  int* base = (int*)ln.ws_int.p;
  SK_CHECK((slot_off + v.size()) * 4 <= ln.ws_int.bytes, SK_EWORKSPACE, "integer workspace too small");
  int* stage = ln.ring_host[ln.ring_cur] + slot_off;
  memcpy(stage, v.data(), v.size() * 4);
  SK_HIP(hipMemcpyAsync(base + slot_off, stage, v.size() * 4, hipMemcpyHostToDevice, st));
  *dptr = base + slot_off;
  return SK_OK;
}
static int ring_end(Lane& ln, hipStream_t st) {
  SK_HIP(hipEventRecord(ln.ring_ev[ln.ring_cur], st));
  ln.ring_used[ln.ring_cur] = true;
  return SK_OK;
}

static int frontend_rows(xt_handle* h, Lane& ln, const void* d_wav, int pcm16, int64_t wav_ld, const BatchMeta& m, float* d_feat_rows, hipStream_t st) {
  const FrontCfg& f = h->fc;
  const int M = m.R ? m.R : m.B * m.T;
  const bool mfcc = h->cfg.arch == XT_ARCH_TDNN;
  float* logmel = mfcc ? (float*)ln.ws_act[3].p : d_feat_rows;   // MFCC: the DCT follows
  GemmArgs p = gemm_args();
  bool have_logmel = false;
  if (!(mfcc && h->mfcc_dft_gemm)) {
    // 1) |rFFT(window * preemph(frame))|^2, one wavefront per frame (frontend_fft.hip)
    FftArgs fa;
    fa.n_fft = f.n_fft;
    fa.wav = d_wav; fa.pcm16 = pcm16; fa.wav_ld = wav_ld; fa.nsamples = m.d_nsamples; fa.nsamples_uniform = m.nsamples_uniform; fa.window = h->d_window;
    fa.tw512 = h->d_tw512; fa.tw1024 = h->d_tw1024; fa.P = (float*)ln.ws_S.p; fa.ldp = h->nbp; fa.M = M; fa.t_max = m.T; fa.hop = f.hop;
    fa.row_b = m.d_row_b; fa.row_t = m.d_row_t; fa.preemph = 0.97f;
    SK_CHECK((size_t)M * h->nbp * 4 <= ln.ws_S.bytes, SK_EWORKSPACE, "spectrum workspace too small (xt_reserve)");
    fa.mel_cw = nullptr; fa.mel_ck0 = nullptr; fa.mel_fmeta = nullptr; fa.mel_chunks = 0; fa.n_mels = 0; fa.logmel = nullptr; fa.ldl = 0;
    if (h->mel_fused && !h->mel_gemm) {  // power spectrum stays in LDS, the kernel writes log-mel rows
      fa.mel_cw = h->d_mel_cw; fa.mel_ck0 = h->d_mel_ck0; fa.mel_fmeta = h->d_mel_fmeta; fa.mel_chunks = h->mel_chunks; fa.n_mels = f.n_mels;
      fa.logmel = logmel; fa.ldl = f.n_mels;
      have_logmel = true;
    }
    { ProfScope ps(h, XT_PROF_FRONTEND, st); SK_TRY(launch_stft_power_fft(fa, st)); }
    // 2) power x mel filterbank, log(. + 1e-6)
    p.a_mode = A_PLAIN; p.A = ln.ws_S.p; p.lda = h->nbp; p.a_rows = M;
  } else {
    // 1) frames x DFT basis -> [re | im]   (A/B form of the MFCC front-end: n_fft 2048, win 1024)
    SK_CHECK(!pcm16, SK_EARG, "the DFT-GEMM A/B form of the MFCC front-end takes float32 samples only");
    GemmArgs g = gemm_args();
    g.a_mode = A_FRAMES; g.A = d_wav; g.wav_ld = wav_ld; g.window = h->d_window; g.nsamples = m.d_nsamples;
    g.nsamples_uniform = m.nsamples_uniform; g.hop = f.hop; g.t_max = m.T; g.row_b = m.d_row_b; g.row_t = m.d_row_t;
    g.preemph = 0.97f;
    g.W = h->d_basis; g.ldw = f.win; g.C = (float*)ln.ws_S.p; g.ldc = 2 * h->nbp; g.M = M; g.N = 2 * h->nbp; g.K = f.win;
    SK_CHECK((size_t)M * 2 * h->nbp * 4 <= ln.ws_S.bytes, SK_EWORKSPACE, "spectrum workspace too small (xt_reserve)");
    { ProfScope ps(h, XT_PROF_FRONTEND, st); SK_TRY(launch_gemm(g, st)); }
    // 2) |.|^2 x mel filterbank, log(. + 1e-6)
    p.a_mode = A_POWER; p.A = ln.ws_S.p; p.lda = 2 * h->nbp; p.kc = h->nbp;
  }
  if (!have_logmel) {
    p.W = h->d_fbT; p.ldw = h->nbp; p.M = M; p.N = f.n_mels; p.K = h->nbp; p.act = ACT_LOG_EPS;
    p.C = logmel; p.ldc = f.n_mels;
    { ProfScope ps(h, XT_PROF_FRONTEND, st); SK_TRY(launch_gemm(p, st)); }
  }
  if (mfcc) {  // 3) DCT-II (ortho) 100 -> 80
    GemmArgs d = gemm_args();
    d.a_mode = A_PLAIN; d.A = logmel; d.lda = f.n_mels; d.a_rows = M; d.W = h->d_dctT; d.ldw = f.n_mels;
    d.C = d_feat_rows; d.ldc = f.n_out; d.M = M; d.N = f.n_out; d.K = f.n_mels;
    { ProfScope ps(h, XT_PROF_FRONTEND, st); SK_TRY(launch_gemm(d, st)); }
  }
  // 4) CMVN over the utterance's own frames
  RowSpan rs{m.d_offsets, m.T, m.lens, 0, 0};
  { ProfScope ps(h, XT_PROF_FRONTEND, st); SK_TRY(launch_cmvn(d_feat_rows, f.n_out, f.n_out, rs, 1e-5f, m.B, st)); }
  return SK_OK;
}

static int tail(xt_handle* h, Lane& ln, int B, float* d_emb, float* d_logits, hipStream_t st, bool normalised = false) {
  const int E = h->cfg.emb_dim;
  SK_TRY(tap(h, "pre_norm", ln.ws_pre.p, (size_t)B * E * 4, st));
  if (!h->norm_embedding && h->cfg.loss == XT_LOSS_CCE) {  // xvector.py:893-898: cce + is_eval returns x as is
    SK_HIP(hipMemcpyAsync(d_emb, ln.ws_pre.p, (size_t)B * E * 4, hipMemcpyDeviceToDevice, st));
    return SK_OK;
  }
  if (!normalised) SK_TRY(launch_l2norm((const float*)ln.ws_pre.p, d_emb, E, B, st));   // else: done by the embedding GEMM's slice-adding kernel
  if (d_logits) {
    SK_CHECK(h->head_wn != nullptr, SK_ESTATE, "logits requested but the model has no cosine head (loss='cce' returns embeddings only)");
    GemmArgs g = gemm_args();
    g.A = d_emb; g.lda = E; g.a_rows = B; g.W = h->head_wn; g.ldw = E; g.C = d_logits; g.ldc = h->cfg.n_spk;
    g.M = B; g.N = h->cfg.n_spk; g.K = E; g.alpha = h->cfg.aam_s;
    SK_TRY(launch_gemm(g, st));
  }
  return SK_OK;
}

// HalfResNet34 from CMVN'ed features with element strides (sb, sf, st)
static int half_from_feats(xt_handle* h, Lane& ln, const float* feats, long sb, long sf, long stt, const BatchMeta& m, float* d_emb,
                           float* d_logits, hipStream_t st) {
  const int dt = h->cfg.dtype == XT_BF16 ? DT_BF16 : DT_F32;
  const int EB = dt == DT_BF16 ? 2 : 4;
  const int B = m.B, T = m.T;
  int Hl[4];
  for (int l = 0; l < 4; ++l) Hl[l] = halve(T, l);
  const size_t act_bytes = (size_t)B * T * 80 * 32 * EB;
  for (int i = 0; i < 4; ++i) SK_CHECK(act_bytes <= ln.ws_act[i].bytes, SK_EWORKSPACE, "activation workspace too small: call xt_reserve(%d, >= %d frames)", B, T);
  void *X = ln.ws_act[0].p, *O1 = ln.ws_act[1].p, *O2 = ln.ws_act[2].p, *SC = ln.ws_act[3].p;
  { ProfScope ps(h, XT_PROF_STEM, st); SK_TRY(launch_stem(feats, sb, sf, stt, h->stem_w, h->stem_shift, X, dt, m.lens, B, T, st)); }
  SK_TRY(tap(h, "stem", X, act_bytes, st));
  int prev_li = 0;
  // A/B builds, SIDEKIT_AMD_PAIR=1 (round 6): layer 1, bf16: conv2 of block k and conv1 of block k + 1 as ONE kernel (conv_pair.hip) -- Y_k reaches
  // conv1 through LDS and HBM sees 13 instead of 15 activation passes for the layer.  Same bits as the two launches, and slower (1.84 vs 1.58 ms per
  // step for the layer: the fused workgroup's chain of barrier-separated memory phases is twice as long and a CU still holds only two of them,
  // profiles/r06_conv_pair_L1.txt): not in the product library.
  const bool use_pair = dt == DT_BF16 && !h->shortcut_tensor && !h->gate_prologue && SK_AB_GETENV("SIDEKIT_AMD_PAIR") != nullptr;
  bool o1_ready = false;   // this block's conv1 has already run (second half of the previous block's pair kernel)
  for (size_t bi = 0; bi < h->blocks.size(); ++bi) {
    Block& b = h->blocks[bi];
    const int li = b.li;
    const bool first = b.has_sc;
    const int lin = first ? (li == 0 ? 0 : li - 1) : li;  // layer index of the block input
    const int wout = 80 >> li;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.lens = m.lens; a.B = B; a.zeros = h->d_zeros; a.persist_cap = ln.persist_cap;
    // conv1 + bn1 + relu -> O1, leaving the sums the block's SE gate is derived from
    a.in = X; a.wpack = b.c1.wpack; a.scale = b.c1.scale; a.shift = b.c1.shift; a.out = O1;
    a.se_part = (float*)ln.ws_se.p; a.col_part = (float*)ln.ws_col.p; a.edge = (float*)ln.ws_edge.p;
    a.halvings_in = lin; a.Hin = Hl[lin]; a.Hout = Hl[li]; a.relu = 1;
    // shortcut rides on conv1's centre tap where that costs no occupancy (the stride-2 shapes already run one workgroup per CU)
    // first block of a layer: by default conv2's epilogue evaluates the 1x1 shortcut conv itself from the block input
    // (no shortcut tensor at all); the older forms -- riding on conv1's centre tap, or a separate 1x1 launch -- remain
    // in A/B builds (make ab; SIDEKIT_AMD_SHORTCUT_TENSOR=1)
    const bool inplace_sc = first && !h->shortcut_tensor;
    const bool fuse_sc = first && !inplace_sc && b.c1.g.stride == 2 && b.c1.g.nw == 1 && b.sc.g.ck == b.c1.g.ck && !b.c1.g.m16;   // (the 16x16x32 k-loop has no fused-shortcut form)
    if (fuse_sc) { a.sc_wpack = b.sc.wpack; a.sc_scale = b.sc.scale; a.sc_shift = b.sc.shift; a.sc_out = SC; }
    {
      const size_t tiles1 = (size_t)cdiv(Hl[li], b.c1.g.th);
      SK_CHECK((size_t)B * tiles1 * b.c1.g.wm * b.C * 4 <= ln.ws_se.bytes && (size_t)B * tiles1 * 2 * b.C * 4 <= ln.ws_col.bytes &&
               (size_t)B * 6 * b.C * 4 <= ln.ws_edge.bytes && (size_t)B * b.C * 4 <= ln.ws_gate.bytes, SK_EWORKSPACE,
               "SE statistics workspace too small for %d x %d frames (xt_reserve)", B, T);
    }
    // SE gate, known before conv2 runs (linearity of the plane mean in O1): its own launch, one workgroup per utterance.  (Round 4 built
    // the alternative the round-3 verdict asked to have measured -- conv1's last workgroup of an utterance computes the gate in its
    // tail, an agent-scope release + ticket per workgroup -- and dropped it: at batch 256 every workgroup's release made the step
    // 24.0 instead of 5.8 ms; selected for small batches only, the dormant tail still cost the statistics kernels their occupancy
    // (layer 3: 161 -> 242 registers + 348 B of scratch, 6.68 ms per step); and at batch 1 the single-workgroup tail was slower than the
    // launch it replaced (0.93 vs 0.75 ms per utterance).  DESIGN.md section 5.)
    SeArgs se;
    se.se_part = (const float*)ln.ws_se.p; se.col_part = (const float*)ln.ws_col.p; se.edge = (const float*)ln.ws_edge.p;
    se.tiles = cdiv(Hl[li], b.c1.g.th); se.wm = b.c1.g.wm; se.th = b.c1.g.th; se.w2t = b.w2t; se.w2t_bf16 = h->cfg.dtype == XT_BF16; se.scale2 = b.c2.scale; se.shift2 = b.c2.shift;
    se.fc1 = b.se_w1; se.fc2 = b.se_w2; se.gate = (float*)ln.ws_gate.p; se.lens = m.lens; se.halvings = li; se.wout = wout; se.C = b.C; se.B = B;
    if (!o1_ready) { ProfScope ps(h, b.c1.shape, st); SK_TRY(launch_conv(b.c1.shape, dt, a, st)); }
    a.sc_wpack = nullptr;
    const void* shortcut = first ? SC : X;
    if (first && !fuse_sc && !inplace_sc) {  // 1x1 conv (stride s) + bn on the block input
      a.wpack = b.sc.wpack; a.scale = b.sc.scale; a.shift = b.sc.shift; a.out = SC;
      a.se_part = nullptr; a.col_part = nullptr; a.edge = nullptr; a.relu = 0;
      { ProfScope ps(h, b.sc.shape, st); SK_TRY(launch_conv(b.sc.shape, dt, a, st)); }
    }
    // small grids (at most SMALL_GRID_MAX_B utterances; 1 is the reference driver's call shape, sidekit/bin/extract_xvectors.py:146): a forward is a
    // chain of dependent launches, each as long as ONE wave's work: conv2 of layers 3-4 runs in 3- / 2-row tiles (more, shorter workgroups).  Same bits.
    // (A/B builds: a SIDEKIT_AMD_SHAPE_MAP may remap conv2's shape to a geometry the T shapes' weight pack does not match -- no small grids then.)
    const bool small = dt == DT_BF16 && !SK_AB_GETENV("SIDEKIT_AMD_SHAPE_MAP") &&
                       (h->small_grid == 2 || (h->small_grid == 1 && B <= xt_handle::SMALL_GRID_MAX_B && (long)B * Hl[li] <= 4096));
    const int c2shape = !small ? b.c2.shape : (li == 2 ? (int)CONV_L3T : (li == 3 ? (int)CONV_L4T : b.c2.shape));
    // the gate inside conv2 (A/B forms, off by default: see xt_handle::gate_prologue).  1 / 2: layers 1-2, conv2's fifth wave computes it beside the k-loop
    // (small grids / always); 3 / 4: every layer, every workgroup computes it before its k-loop (small grids / always)
    const bool small_b = B <= xt_handle::GATE_AB_MAX_B && (long)B * Hl[li] <= 4096;
    const int gate_pro = (h->gate_prologue == 2 || (h->gate_prologue == 1 && small_b)) ? (b.C <= 64 ? 2 : 0)
                         : ((h->gate_prologue == 4 || (h->gate_prologue == 3 && small_b)) ? 1 : 0);
    if (!gate_pro) {
      ProfScope ps(h, XT_PROF_SE_RES, st);
      SK_TRY(launch_se_pre(se, st));
    }
#ifdef SK_AB
    const bool pair = use_pair && li == 0 && bi + 1 < h->blocks.size() && h->blocks[bi + 1].li == li && (!first || inplace_sc);
    if (pair) {
      const Block& nb = h->blocks[bi + 1];
      ConvPairArgs pa;
      memset(&pa, 0, sizeof(pa));
      pa.C = b.C; pa.W = wout;
      pa.in = O1; pa.w2pack = b.c2.wpack; pa.scale2 = b.c2.scale; pa.shift2 = b.c2.shift; pa.gate = (const float*)ln.ws_gate.p;
      if (first) { pa.sc_in = X; pa.sc_wpack = b.sc_wfold; pa.sc_shift = b.sc.shift; } else { pa.shortcut = X; }
      pa.y_out = O2;
      pa.w1pack = nb.c1.wpack; pa.scale1 = nb.c1.scale; pa.shift1 = nb.c1.shift; pa.o_out = SC;
      pa.se_part = (float*)ln.ws_se.p; pa.col_part = (float*)ln.ws_col.p; pa.edge = (float*)ln.ws_edge.p;
      pa.zeros = h->d_zeros; pa.lens = m.lens; pa.B = B; pa.H = Hl[li]; pa.persist_cap = ln.persist_cap;
      { ProfScope ps(h, b.c2.shape, st); SK_TRY(launch_conv_pair(pa, st)); }
      // Y_k (O2) is the next block's input, O1_{k+1} (SC) its conv1 output; the two buffers just read are free
      void *old_x = X, *old_o1 = O1;
      X = O2; O1 = SC; O2 = old_x; SC = old_o1;
      o1_ready = true;
      prev_li = li;
      continue;
    }
#endif
    (void)use_pair;
    o1_ready = false;
    // conv2 + bn2, * gate, + shortcut, relu -> O2 (the block output)
    a.in = O1; a.wpack = b.c2.wpack; a.scale = b.c2.scale; a.shift = b.c2.shift; a.out = O2;
    a.se_part = nullptr; a.col_part = nullptr; a.edge = nullptr; a.gate = (const float*)ln.ws_gate.p; a.shortcut = shortcut;
    a.halvings_in = li; a.Hin = Hl[li]; a.Hout = Hl[li]; a.relu = 0;
    if (gate_pro) { a.gate_pro = gate_pro; a.se = se; }     // every workgroup of conv2 derives its utterance's gate itself; ws_gate is not read
    if (inplace_sc) {
      a.shortcut = nullptr; a.sc_in = X; a.sc_hin = Hl[lin];
      a.sc_wpack = b.sc_wfold; a.sc_scale = b.sc.scale; a.sc_shift = b.sc.shift;
    }
    { ProfScope ps(h, b.c2.shape, st); SK_TRY(launch_conv((gate_pro == 2 && li == 0 && dt == DT_BF16) ? (int)CONV_L1G : c2shape, dt, a, st)); }
    std::swap(X, O2);
    const bool last_of_layer = (bi + 1 == h->blocks.size()) || (h->blocks[bi + 1].li != li);
    if (last_of_layer) {
      const std::string nm = "layer" + std::to_string(li + 1);
      SK_TRY(tap(h, nm.c_str(), X, (size_t)B * Hl[li] * wout * b.C * EB, st));
    }
    prev_li = li;
  }
  (void)prev_li;
  // ---- attentive statistics pooling (pooling.py:151-171), rows = (b, t'), columns d' = f*256 + c
  const int H4 = Hl[3], D = 2560, R = B * H4;
  const int xbf = dt == DT_BF16;
  RowSpan rs{nullptr, H4, m.lens, 3, 0};
  ProfScope ps_pool(h, XT_PROF_POOL_TAIL, st);
  SK_TRY(launch_mean_std(X, xbf, D, D, rs, (float*)ln.ws_ctx.p, B, st));
  GemmArgs c = gemm_args();  // context term of attention.0: W1[:, 2560:] . [mean | std] + bias, once per utterance
  c.A = ln.ws_ctx.p; c.lda = 2 * D; c.a_rows = B; c.W = h->att_w1c; c.ldw = 2 * D; c.C = (float*)ln.ws_rb.p; c.ldc = 128;
  c.M = B; c.N = 128; c.K = 2 * D; c.bias = h->att_b1; c.splitk_ws = (float*)ln.ws_splitk.p;
  SK_TRY(launch_gemm(c, st));
  GemmArgs g1 = gemm_args();  // attention.0 on x + ReLU + BatchNorm1d + tanh
  g1.A = X; g1.a_bf16 = xbf; g1.lda = D; g1.a_rows = R; g1.W = h->att_w1x; g1.ldw = D; g1.C = (float*)ln.ws_h.p; g1.ldc = 128;
  g1.M = R; g1.N = 128; g1.K = D; g1.rowbias = (const float*)ln.ws_rb.p; g1.rows_per_group = H4;
  g1.act = ACT_RELU_BN_TANH; g1.scale = h->att_bn_scale; g1.shift = h->att_bn_shift; g1.W_bf16 = h->att_w1x_bf16;
  if (xbf && (size_t)8 * 512 * 128 * 4 <= ln.ws_splitk.bytes) g1.splitk_ws = (float*)ln.ws_splitk.p;   // bf16 path, at most 512 rows (a few utterances): K in eight slices side by side (gemm.hip)
  SK_TRY(launch_gemm(g1, st));
  if (xbf && !SK_AB_GETENV("SIDEKIT_AMD_ATT_SEPARATE")) {   // bf16 path: attention.4 + softmax + statistics fused, e never leaves the accumulators
    SK_TRY(launch_att_fused(X, (const float*)ln.ws_h.p, h->att_w2_bf16, h->att_b2, D, D, rs, (float*)ln.ws_pooled.p, B, st));
  } else {
  GemmArgs g2 = gemm_args();  // attention.4
  g2.A = ln.ws_h.p; g2.lda = 128; g2.a_rows = R; g2.W = h->att_w2; g2.ldw = 128; g2.C = (float*)ln.ws_e.p; g2.ldc = D;
  g2.M = R; g2.N = D; g2.K = 128; g2.bias = h->att_b2; g2.W_bf16 = h->att_w2_bf16;
  SK_TRY(launch_gemm(g2, st));
  SK_TRY(launch_att_stats(X, xbf, (const float*)ln.ws_e.p, D, D, rs, (float*)ln.ws_pooled.p, B, st));
  }
  SK_TRY(tap(h, "pooled", ln.ws_pooled.p, (size_t)B * 2 * D * 4, st));
  GemmArgs e = gemm_args();  // lin_be + bn_be (xvector.py:578-581)
  e.A = ln.ws_pooled.p; e.lda = 2 * D; e.a_rows = B; e.W = h->emb_w; e.ldw = 2 * D; e.C = (float*)ln.ws_pre.p;
  e.ldc = h->cfg.emb_dim; e.M = B; e.N = h->cfg.emb_dim; e.K = 2 * D; e.scale = h->emb_scale; e.shift = h->emb_shift;
  if (h->cfg.emb_dim <= 256) e.splitk_ws = (float*)ln.ws_splitk.p;
  int l2_done = 0;
  e.l2_out = d_emb; e.l2_done = &l2_done;     // the aam head always normalises (xvector.py:903)
  SK_TRY(launch_gemm(e, st));
  return tail(h, ln, B, d_emb, d_logits, st, l2_done != 0);
}

// TDNN from CMVN'ed MFCC rows [R][80]
static int tdnn_from_rows(xt_handle* h, Lane& ln, const float* rows, const BatchMeta& m, float* d_emb, float* d_logits, hipStream_t st) {
  const int R = m.R;
  const float* in = rows;
  int lda = 80;
  float* bufs[2] = {(float*)ln.ws_act[0].p, (float*)ln.ws_act[1].p};
  for (int i = 0; i < 5; ++i) {
    const auto& L = h->tdnn[i];
    float* out = bufs[i & 1];
    SK_CHECK((size_t)R * L.cout * 4 <= ln.ws_act[i & 1].bytes, SK_EWORKSPACE, "TDNN activation workspace too small (xt_reserve)");
    GemmArgs g = gemm_args();
    g.A = in; g.lda = lda; g.a_rows = R; g.kc = L.k > 1 ? L.cin : 0; g.dil = L.dil;
    g.W = L.w; g.ldw = (long)L.cin * L.k; g.C = out; g.ldc = L.cout; g.M = R; g.N = L.cout; g.K = L.cin * L.k;
    g.bias = L.bias; g.act = ACT_LRELU02; g.scale = L.scale; g.shift = L.shift;  // conv -> LeakyReLU(0.2) -> BatchNorm1d
    { ProfScope ps(h, XT_PROF_TDNN, st); SK_TRY(launch_gemm(g, st)); }
    const std::string nm = "conv" + std::to_string(i + 1);
    SK_TRY(tap(h, nm.c_str(), out, (size_t)R * L.cout * 4, st));
    in = out; lda = L.cout;
  }
  ProfScope ps_pool(h, XT_PROF_POOL_TAIL, st);
  RowSpan rs{m.d_offsets, 0, m.lens, 0, 14};  // context_size()-1 = 4 + 4 + 6 frames consumed by the valid convs
  SK_TRY(launch_mean_std(in, 0, 1536, 1536, rs, (float*)ln.ws_pooled.p, m.B, st));
  SK_TRY(tap(h, "pooled", ln.ws_pooled.p, (size_t)m.B * 3072 * 4, st));
  GemmArgs e = gemm_args();  // linear6 (xvector.py:489-491)
  e.A = ln.ws_pooled.p; e.lda = 3072; e.a_rows = m.B; e.W = h->emb_w; e.ldw = 3072; e.C = (float*)ln.ws_pre.p; e.ldc = h->cfg.emb_dim;
  e.M = m.B; e.N = h->cfg.emb_dim; e.K = 3072; e.bias = h->emb_bias;
  if (h->cfg.emb_dim <= 256) e.splitk_ws = (float*)ln.ws_splitk.p;
  int l2_done = 0;
  if (h->norm_embedding || h->cfg.loss != XT_LOSS_CCE) { e.l2_out = d_emb; e.l2_done = &l2_done; }
  SK_TRY(launch_gemm(e, st));
  return tail(h, ln, m.B, d_emb, h->cfg.loss == XT_LOSS_AAM ? d_logits : nullptr, st, l2_done != 0);
}

static int make_meta(xt_handle* h, Lane& ln, const int32_t* h_counts, int B, int64_t L_or_T, bool counts_are_samples, BatchMeta& m,
                     hipStream_t st) {
  const FrontCfg& f = h->fc;
  m.B = B;
  std::vector<int> frames(B), nsamp(B);
  bool uniform = true;
  int tmax = 0;
  for (int b = 0; b < B; ++b) {
    const int64_t c = h_counts ? h_counts[b] : L_or_T;
    SK_CHECK(c > 0 && c <= L_or_T, SK_EARG, "utterance %d: length %lld outside (0, %lld]", b, (long long)c, (long long)L_or_T);
    if (counts_are_samples) {
      // torch.stft(center=True, pad_mode='reflect') needs n_fft/2 < L (RuntimeError in the reference otherwise)
      SK_CHECK(c > f.n_fft / 2, SK_EARG, "utterance %d: %lld samples, reflect padding needs more than %d", b, (long long)c, f.n_fft / 2);
      nsamp[b] = (int)c;
      frames[b] = 1 + (int)(c / f.hop);
    } else {
      frames[b] = (int)c;
    }
    if (frames[b] != frames[0] || (counts_are_samples && nsamp[b] != nsamp[0])) uniform = false;
    tmax = frames[b] > tmax ? frames[b] : tmax;
  }
  m.T = tmax;
  const bool ragged = h->cfg.arch == XT_ARCH_TDNN;
  if (ragged)
    for (int b = 0; b < B; ++b) SK_CHECK(frames[b] >= 15, SK_EARG, "utterance %d: %d frames < TDNN context of 15", b, frames[b]);
  size_t off = 0;
  if (!(uniform && !ragged)) SK_TRY(ring_begin(ln));
  if (uniform && !ragged) {
    m.lens = Lens{nullptr, frames[0]};
    m.nsamples_uniform = counts_are_samples ? nsamp[0] : 0;
  } else {
    const int* p;
    SK_TRY(push_ints(ln, frames, off, &p, st)); off += B;
    m.lens = Lens{p, 0};
    if (counts_are_samples) { SK_TRY(push_ints(ln, nsamp, off, &m.d_nsamples, st)); off += B; }
  }
  if (ragged) {
    std::vector<int> offs(B), rb, rt;
    int R = 0;
    for (int b = 0; b < B; ++b) { offs[b] = R; R += frames[b]; }
    rb.resize(R); rt.resize(R);
    for (int b = 0; b < B; ++b)
      for (int t = 0; t < frames[b]; ++t) { rb[offs[b] + t] = b; rt[offs[b] + t] = t; }
    m.R = R;
    SK_TRY(push_ints(ln, offs, off, &m.d_offsets, st)); off += B;
    SK_TRY(push_ints(ln, rb, off, &m.d_row_b, st)); off += R;
    SK_TRY(push_ints(ln, rt, off, &m.d_row_t, st)); off += R;
  }
  if (!(uniform && !ragged)) SK_TRY(ring_end(ln, st));
  return SK_OK;
}

}  // namespace sk

// ================================================================================================
extern "C" {

const char* xt_last_error(void) { return sk::last_error(); }

int xt_create(const xt_config* cfg, xt_handle** out) {
  SK_CHECK(cfg && out, SK_EARG, "xt_create: null argument");
  SK_CHECK(cfg->arch == XT_ARCH_HALFRESNET34 || cfg->arch == XT_ARCH_TDNN, SK_EARG, "xt_create: unknown arch %d", cfg->arch);
  SK_CHECK(cfg->dtype == XT_F32 || cfg->dtype == XT_BF16, SK_EARG, "xt_create: dtype must be XT_F32 or XT_BF16");
  SK_CHECK(cfg->arch == XT_ARCH_HALFRESNET34 || cfg->dtype == XT_F32, SK_EARG, "xt_create: the TDNN runs in fp32 only");
  SK_CHECK(cfg->loss == XT_LOSS_AAM || (cfg->loss == XT_LOSS_CCE && cfg->arch == XT_ARCH_TDNN), SK_EARG, "xt_create: unsupported loss for this arch");
  SK_CHECK(cfg->n_spk > 0 && cfg->emb_dim > 0 && cfg->emb_dim % 4 == 0, SK_EARG, "xt_create: bad n_spk / emb_dim");
  xt_handle* h = new xt_handle();
  h->cfg = *cfg;
  h->fc = cfg->arch == XT_ARCH_HALFRESNET34 ? MELSPEC : MFCCCFG;
  int rc = hipGetDevice(&h->device);
  if (rc != hipSuccess) { sk::set_error("hipGetDevice failed: %s", hipGetErrorString((hipError_t)rc)); delete h; return SK_EHIP; }
  for (auto& kv : expected_keys(*cfg)) {
    h->keys.push_back(kv.first);
    HostTensor t; t.shape = kv.second;
    h->tensors[kv.first] = t;
  }
  *out = h;
  return SK_OK;
}

int xt_destroy(xt_handle* h) {
  if (!h) return SK_OK;
  for (void* p : h->dev_allocs) (void)hipFree(p);
  for (Lane& ln : h->lane) {
    DevBuf* bufs[] = {&ln.ws_S, &ln.ws_feat, &ln.ws_act[0], &ln.ws_act[1], &ln.ws_act[2], &ln.ws_act[3], &ln.ws_se, &ln.ws_col, &ln.ws_edge, &ln.ws_splitk, &ln.ws_gate,
                      &ln.ws_ctx, &ln.ws_rb, &ln.ws_h, &ln.ws_e, &ln.ws_pooled, &ln.ws_pre, &ln.ws_int};
    for (DevBuf* b : bufs) b->release();
    for (int i = 0; i < Lane::RING; ++i)
      if (ln.ring_host[i]) { (void)hipHostFree(ln.ring_host[i]); (void)hipEventDestroy(ln.ring_ev[i]); }
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
    if (ln.fork) (void)hipEventDestroy(ln.fork);
    if (ln.join) (void)hipEventDestroy(ln.join);
  }
  if (h->tail) (void)hipEventDestroy(h->tail);
  for (auto& kv : h->taps) kv.second.buf.release();
  for (auto& r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto e : h->prof_pool) (void)hipEventDestroy(e);
  delete h;
  return SK_OK;
}

int xt_num_keys(xt_handle* h) { return h ? (int)h->keys.size() : 0; }
const char* xt_key_name(xt_handle* h, int32_t i) { return (h && i >= 0 && i < (int)h->keys.size()) ? h->keys[i].c_str() : nullptr; }

int xt_set_tensor(xt_handle* h, const char* key, const void* h_data, const int64_t* shape, int32_t ndim, int32_t dtype) {
  SK_CHECK(h && key && (h_data || ndim == 0), SK_EARG, "xt_set_tensor: null argument");
  SK_CHECK(!h->finalized, SK_ESTATE, "xt_set_tensor after xt_finalize");
  auto it = h->tensors.find(key);
  SK_CHECK(it != h->tensors.end(), SK_ESHAPE, "Unexpected key(s) in state_dict: \"%s\"", key);
  HostTensor& t = it->second;
  bool same = (size_t)ndim == t.shape.size();
  for (int i = 0; same && i < ndim; ++i) same = shape[i] == t.shape[i];
  if (!same) {
    std::string got = "[", want = "[";
    for (int i = 0; i < ndim; ++i) got += std::to_string(shape[i]) + (i + 1 < ndim ? ", " : "");
    for (size_t i = 0; i < t.shape.size(); ++i) want += std::to_string(t.shape[i]) + (i + 1 < t.shape.size() ? ", " : "");
    sk::set_error("size mismatch for %s: copying a param with shape %s], the shape in current model is %s]", key, got.c_str(), want.c_str());
    return SK_ESHAPE;
  }
  const size_t n = t.numel();
  t.data.resize(n);
  if (dtype == XT_F32) memcpy(t.data.data(), h_data, n * 4);
  else if (dtype == XT_I64) for (size_t i = 0; i < n; ++i) t.data[i] = (float)((const int64_t*)h_data)[i];
  else if (dtype == XT_F64) for (size_t i = 0; i < n; ++i) t.data[i] = (float)((const double*)h_data)[i];
  else { sk::set_error("xt_set_tensor: unsupported dtype %d for %s", dtype, key); return SK_EARG; }
  t.set = true;
  return SK_OK;
}

int xt_finalize(xt_handle* h) {
  SK_CHECK(h, SK_EARG, "xt_finalize: null handle");
  SK_CHECK(!h->finalized, SK_ESTATE, "xt_finalize called twice");
  for (auto& k : h->keys) SK_CHECK(h->tensors[k].set, SK_ESHAPE, "Missing key(s) in state_dict: \"%s\"", k.c_str());
  SK_HIP(hipSetDevice(h->device));
  SK_TRY(build_frontend(h));
  if (h->cfg.arch == XT_ARCH_HALFRESNET34) SK_TRY(finalize_half(h)); else SK_TRY(finalize_tdnn(h));
  h->finalized = true;
  return SK_OK;
}

static int reserve_lane(xt_handle* h, Lane& ln, int32_t max_batch, int64_t max_samples) {
  const FrontCfg& f = h->fc;
  const size_t B = (size_t)max_batch;
  const size_t T = 1 + (size_t)(max_samples / f.hop);
  const size_t nbp = (size_t)((f.n_fft / 2 + 1 + 3) / 4 * 4);
  const size_t R = B * T;
  SK_TRY(ln.ws_S.ensure(R * 2 * nbp * 4));
  SK_TRY(ln.ws_feat.ensure(R * f.n_out * 4));
  const size_t int_bytes = (4 * B + 2 * R + 16) * 4;
  SK_TRY(ln.ws_int.ensure(int_bytes));
  if (int_bytes > ln.ring_bytes) {
    for (int i = 0; i < Lane::RING; ++i) {
      if (ln.ring_host[i]) { if (ln.ring_used[i]) SK_HIP(hipEventSynchronize(ln.ring_ev[i])); SK_HIP(hipHostFree(ln.ring_host[i])); }
      else SK_HIP(hipEventCreateWithFlags(&ln.ring_ev[i], hipEventDisableTiming));
      SK_HIP(hipHostMalloc((void**)&ln.ring_host[i], int_bytes, hipHostMallocDefault));
      ln.ring_used[i] = false;
    }
    ln.ring_bytes = int_bytes;
  }
  const size_t E = (size_t)h->cfg.emb_dim;
  SK_TRY(ln.ws_pre.ensure(B * E * 4));
  {
    const size_t skinny = (size_t)32 * (B < 512 ? B : 512) * 256 * 4, att0 = (size_t)8 * 512 * 128 * 4;   // [slices][rows <= 512][N]: the embedding / context GEMMs; attention.0 below 512 rows
    SK_TRY(ln.ws_splitk.ensure(skinny > att0 ? skinny : att0));
  }  // split-K partials of the skinny GEMMs (N <= 256)
  if (h->cfg.arch == XT_ARCH_HALFRESNET34) {
    const size_t EB = h->cfg.dtype == XT_BF16 ? 2 : 4;
    for (int i = 0; i < 4; ++i) SK_TRY(ln.ws_act[i].ensure(R * 80 * 32 * EB));
    // SE statistics of the statistics-form convolutions: [B][row tiles][wave rows][C] totals and [B][row tiles][2][C] column
    // sums, sized from the largest first convolution of a block (short utterances: one 2-KB tile of layer 4 per utterance
    // outgrows a per-frame estimate)
    size_t se_b = 0, col_b = 0;
    for (const Block& b : h->blocks) {
      const size_t tiles = (size_t)cdiv(halve((int)T, b.li), b.c1.g.th);
      const size_t s1 = B * tiles * b.c1.g.wm * b.C * 4, s2 = B * tiles * 2 * b.C * 4;
      se_b = s1 > se_b ? s1 : se_b;
      col_b = s2 > col_b ? s2 : col_b;
    }
    SK_TRY(ln.ws_se.ensure(se_b));
    SK_TRY(ln.ws_col.ensure(col_b));
    SK_TRY(ln.ws_edge.ensure(B * 6 * 256 * 4));
    SK_TRY(ln.ws_gate.ensure(B * 256 * 4));
    const size_t H4 = (size_t)halve((int)T, 3);
    SK_TRY(ln.ws_ctx.ensure(B * 5120 * 4));
    SK_TRY(ln.ws_rb.ensure(B * 128 * 4));
    SK_TRY(ln.ws_h.ensure(B * H4 * 128 * 4));
    SK_TRY(ln.ws_e.ensure(B * H4 * 2560 * 4));
    SK_TRY(ln.ws_pooled.ensure(B * 5120 * 4));
  } else {
    SK_TRY(ln.ws_act[0].ensure(R * 1536 * 4));
    SK_TRY(ln.ws_act[1].ensure(R * 512 * 4));
    SK_TRY(ln.ws_act[3].ensure(R * f.n_mels * 4));
    SK_TRY(ln.ws_pooled.ensure(B * 3072 * 4));
  }
  return SK_OK;
}

// lanes 1 .. n-1 take the later parts of a split batch (part k of n': rows [k B / n', (k + 1) B / n')).  A batch SMALLER than the
// reserved one is split into FEWER, larger parts (lanes = 4, reserve 256: B = 255 gives three parts of 85), so lane k is sized for the
// largest part it can ever be handed by a batch this shape covers: it runs only when n' >= k + 1, i.e. at most ceil(max_batch / (k + 1))
// utterances (sizing it for ceil(max_batch / n) let the STFT kernel write past ws_feat on a final partial batch).
static int reserve_side_lanes(xt_handle* h, int32_t max_batch, int64_t max_samples) {
  if (h->cfg.arch != XT_ARCH_HALFRESNET34) return SK_OK;
  const int n = lane_parts(h->lanes, max_batch, xt_handle::LANE_MIN);
  for (int k = 0; k < n && n > 1; ++k) {
    Lane& lk = h->lane[k];
    if (!lk.stream) {
      // Every part of a split batch runs on a stream the handle owns (part 0 as well: the caller's stream only forks and joins), all
      // created at ONE priority other than the caller's.  Why: the runtime multiplexes the streams of one priority onto at most
      // GPU_MAX_HW_QUEUES (4) hardware queues, least-referenced first, and two streams on one queue run one after the other.  In a
      // plain process the second stream created gets its own queue; in a process that has initialised RCCL (torch's pool of 32
      // normal-priority streams exists) a normal-priority lane stream landed on the caller's queue and the two-lane forward was
      // SLOWER than the serial one (round 4, one rank under torch.distributed.run: 6.41-6.60 vs 5.93-6.08 ms,
      // scripts/rccl_step_probe.py).  Queues are pooled per priority, so lanes of another priority get queues of their own; and
      // they must all have the SAME priority: one high-priority lane beside the caller's normal stream ran ahead of it instead of
      // beside it and the overlap was gone (5.92 vs 5.73 ms).  Product: low; A/B builds: SIDEKIT_AMD_LANE_PRIORITY = low | high | normal.
      static const char* pe = SK_AB_GETENV("SIDEKIT_AMD_LANE_PRIORITY");
      int least = 0, greatest = 0;
      SK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
      const int prio = (pe && !strcmp(pe, "normal")) ? 0 : ((pe && !strcmp(pe, "high")) ? greatest : least);
      SK_HIP(hipStreamCreateWithPriority(&lk.stream, hipStreamNonBlocking, prio));
      SK_HIP(hipEventCreateWithFlags(&lk.fork, hipEventDisableTiming));
      SK_HIP(hipEventCreateWithFlags(&lk.join, hipEventDisableTiming));
    }
    if (k == 0) continue;   // lane 0's workspace is the handle's full-size one
    const int part = (max_batch + k) / (k + 1);
    SK_TRY(reserve_lane(h, lk, part, max_samples));
    if (!lk.covers(part, max_samples)) lk.reserved.push_back({part, max_samples});
  }
  return SK_OK;
}

static int reserve_impl(xt_handle* h, int32_t max_batch, int64_t max_samples) {
  SK_CHECK(h && max_batch > 0 && max_samples > 0, SK_EARG, "xt_reserve: bad arguments");
  SK_HIP(hipSetDevice(h->device));
  SK_TRY(reserve_lane(h, h->lane[0], max_batch, max_samples));
  SK_TRY(reserve_side_lanes(h, max_batch, max_samples));
  bool covered = false;
  for (auto& r : h->reserved) covered = covered || (r.first >= max_batch && r.second >= max_samples);
  if (!covered) h->reserved.push_back({max_batch, max_samples});
  return SK_OK;
}

int xt_reserve(xt_handle* h, int32_t max_batch, int64_t max_samples) {
  SK_ENTER(h);
  return reserve_impl(h, max_batch, max_samples);
}

static int check_run(xt_handle* h, int B, int64_t L_samples) {
  SK_CHECK(h, SK_EARG, "null handle");
  SK_CHECK(h->finalized, SK_ESTATE, "forward before xt_finalize (load_state_dict)");
  SK_CHECK(B > 0, SK_EARG, "empty batch");
  SK_CHECK(!h->reserved.empty(), SK_ESTATE, "forward before xt_reserve");
  bool covered = false;
  for (auto& r : h->reserved) covered = covered || (r.first >= B && r.second >= L_samples);
  SK_CHECK(covered, SK_EWORKSPACE, "batch of %d x %lld samples exceeds every reserved workspace shape: call xt_reserve(%d, %lld)", B,
           (long long)L_samples, B, (long long)L_samples);
  SK_HIP(hipSetDevice(h->device));
  return SK_OK;
}

// One lane's forward in two halves: front-end (wav -> CMVN'ed features in the lane's workspace) and trunk + pooling + tail.
static int lane_frontend(xt_handle* h, Lane& ln, const void* d_wav, int pcm16, int64_t wav_ld, const int32_t* h_nsamples, int32_t B, int64_t L,
                         BatchMeta& m, hipStream_t st) {
  SK_TRY(make_meta(h, ln, h_nsamples, B, L, true, m, st));
  float* feat = (float*)ln.ws_feat.p;
  const int M = m.R ? m.R : m.B * m.T;
  SK_CHECK((size_t)M * h->fc.n_out * 4 <= ln.ws_feat.bytes, SK_EWORKSPACE, "feature workspace too small for %d x %d frames (xt_reserve)", m.B, m.T);
  SK_TRY(frontend_rows(h, ln, d_wav, pcm16, wav_ld, m, feat, st));
  return tap(h, "feats", feat, (size_t)M * h->fc.n_out * 4, st);
}

static int lane_trunk(xt_handle* h, Lane& ln, const BatchMeta& m, float* d_emb, float* d_logits, hipStream_t st) {
  float* feat = (float*)ln.ws_feat.p;
  if (h->cfg.arch == XT_ARCH_HALFRESNET34) return half_from_feats(h, ln, feat, (long)m.T * 80, 1, 80, m, d_emb, d_logits, st);
  return tdnn_from_rows(h, ln, feat, m, d_emb, d_logits, st);
}

static int forward_wav(xt_handle* h, const void* d_wav, int pcm16, int64_t wav_ld, const int32_t* h_nsamples, int32_t B, int64_t L,
                       float* d_emb, float* d_logits, void* stream) {
  SK_ENTER(h);
  SK_TRY(check_run(h, B, L));
  SK_CHECK(d_wav && d_emb && wav_ld >= L, SK_EARG, "xt_forward: bad buffers");
  hipStream_t st = (hipStream_t)stream;
  SK_TRY(enter_stream(h, st));
  StreamScope scope_(h, st);
  Lane& l0 = h->lane[0];
  int n = (h->debug || h->cfg.arch != XT_ARCH_HALFRESNET34) ? 1 : lane_parts(h->lanes, B, xt_handle::LANE_MIN);
  // fewer parts until every side lane exists (a handle switched after its last reserve) and its workspace covers its part: nothing
  // is enqueued before every part is known to fit
  auto parts_fit = [&](int np) {
    if (!h->lane[0].stream) return false;
    for (int k = 1; k < np; ++k) {
      const int r0 = (int)((long)k * B / np), r1 = (int)((long)(k + 1) * B / np);
      if (!h->lane[k].stream || !h->lane[k].covers(r1 - r0, L)) return false;
    }
    return true;
  };
  while (n > 1 && !parts_fit(n)) --n;
  BatchMeta m0;
  for (int k = 0; k < xt_handle::MAX_LANES; ++k) h->lane[k].persist_cap = 0;
  if (n == 1) {
    // an unsplit forward runs on the CALLER's stream in lane 0's workspace: a pipelined batch (xt_forward_begin, slot 0) may still be running there on
    // lane 0's own stream -- order behind it (found by scripts/soak_pipelined.py: a small plain forward between two submits raced with slot 0)
    if (l0.stream) SK_HIP(hipStreamWaitEvent(st, l0.join, 0));
    SK_TRY(lane_frontend(h, l0, d_wav, pcm16, wav_ld, h_nsamples, B, L, m0, st));
    return lane_trunk(h, l0, m0, d_emb, d_logits, st);
  }
  // n lanes: part k = rows [k B / n, (k + 1) B / n), every part on a stream the handle owns (reserve_side_lanes says why part 0 too).
  // A lane starts behind everything queued on the caller's stream so far (its input may still be in flight) and the caller's stream
  // continues only once every part is done.  Round 3 found the first version of this giving a few wrong spectrum bins per batch in
  // the second lane: the STFT kernel's SLP-formed packed-f32 instructions (v_pk_add_f32 / v_pk_mul_f32 with op_sel / neg modifiers)
  // misbehave on MI355X beside another stream's dense bf16 MFMAs.  The library is built without them now (csrc/Makefile; DESIGN 6;
  // tests/test_isa_guard.py keeps them out) and tests/test_gpu_fullsize.py repeats the split forward against the serial one.
  const size_t eb = pcm16 ? 2 : 4;
  SK_HIP(hipEventRecord(h->lane[0].fork, st));
  for (int k = 0; k < n; ++k) SK_HIP(hipStreamWaitEvent(h->lane[k].stream, h->lane[0].fork, 0));
  // An error in one part must not leave the others running unjoined: the caller's stream would no longer order behind the side
  // streams (which keep writing d_emb / d_logits) and the next call would reuse their workspaces.  So every lane that was forked
  // is joined whatever happened, and the first error is what the call returns.
  int rc = SK_OK;
  char first_err[sizeof(g_err)] = "";
  for (int k = 0; k < n && rc == SK_OK; ++k) {
    Lane& lk = h->lane[k];
    const int r0 = (int)((long)k * B / n), r1 = (int)((long)(k + 1) * B / n);
    hipStream_t sk_ = lk.stream;
    BatchMeta mk;
    rc = lane_frontend(h, lk, (const unsigned char*)d_wav + (size_t)r0 * wav_ld * eb, pcm16, wav_ld, h_nsamples ? h_nsamples + r0 : nullptr, r1 - r0, L, mk, sk_);
    if (rc == SK_OK) rc = lane_trunk(h, lk, mk, d_emb + (size_t)r0 * h->cfg.emb_dim, d_logits ? d_logits + (size_t)r0 * h->cfg.n_spk : nullptr, sk_);
    if (rc != SK_OK) snprintf(first_err, sizeof(first_err), "%s", g_err);
  }
  for (int k = 0; k < n; ++k) {
    const bool joined = hipEventRecord(h->lane[k].join, h->lane[k].stream) == hipSuccess && hipStreamWaitEvent(st, h->lane[k].join, 0) == hipSuccess;
    if (!joined) {   // last resort: drain the side stream on the host
      (void)hipStreamSynchronize(h->lane[k].stream);
      if (rc == SK_OK) { rc = SK_EHIP; snprintf(first_err, sizeof(first_err), "xt_forward: joining lane %d failed", k); }
    }
  }
  if (rc != SK_OK) set_error("%s", first_err);
  return rc;
}

int xt_forward(xt_handle* h, const float* d_wav, int64_t wav_ld, const int32_t* h_nsamples, int32_t B, int64_t L, float* d_emb,
               float* d_logits, void* stream) {
  return forward_wav(h, d_wav, 0, wav_ld, h_nsamples, B, L, d_emb, d_logits, stream);
}

int xt_forward_pcm16(xt_handle* h, const int16_t* d_pcm, int64_t pcm_ld, const int32_t* h_nsamples, int32_t B, int64_t L, float* d_emb,
                     float* d_logits, void* stream) {
  return forward_wav(h, d_pcm, 1, pcm_ld, h_nsamples, B, L, d_emb, d_logits, stream);
}

// ---- pipelined forwards (round 4) -----------------------------------------------------------------------------------------------
// Two WHOLE batches in flight on two streams the handle owns beat two half batches side by side (5.67 vs 5.87 ms per batch of 256 in one bench.py run, profiles/r05_bench_line.json; first measured by
// scripts/alt_streams.py): consecutive forwards run half a step apart, so one batch's HBM-bound layer 1 overlaps the other's
// MFMA-bound layers 3-4 -- and nothing joins at the end of a call.  A slot is a full-size workspace + a stream; xt_forward_begin queues
// the whole (serial) forward of a batch on its slot's stream behind everything queued on the caller's stream so far and returns;
// xt_forward_end makes a stream wait for that forward.  A caller keeps `slots` batches in flight: begin(k), end(k - slots + 1), ...
static int reserve_slot(xt_handle* h, int slot, int32_t max_batch, int64_t max_samples) {
  Lane& lk = h->lane[slot];
  if (!lk.stream) {
    static const char* pe = SK_AB_GETENV("SIDEKIT_AMD_LANE_PRIORITY");
    int least = 0, greatest = 0;
    SK_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int prio = (pe && !strcmp(pe, "normal")) ? 0 : ((pe && !strcmp(pe, "high")) ? greatest : least);
    SK_HIP(hipStreamCreateWithPriority(&lk.stream, hipStreamNonBlocking, prio));
    SK_HIP(hipEventCreateWithFlags(&lk.fork, hipEventDisableTiming));
    SK_HIP(hipEventCreateWithFlags(&lk.join, hipEventDisableTiming));
  }
  SK_TRY(reserve_lane(h, lk, max_batch, max_samples));
  if (slot > 0 && !lk.covers(max_batch, max_samples)) lk.reserved.push_back({max_batch, max_samples});
  return SK_OK;
}

int xt_reserve_slots(xt_handle* h, int32_t slots, int32_t max_batch, int64_t max_samples) {
  SK_CHECK(h && slots >= 1 && slots <= xt_handle::MAX_LANES && max_batch > 0 && max_samples > 0, SK_EARG, "xt_reserve_slots: 1 .. %d slots", xt_handle::MAX_LANES);
  SK_ENTER(h);
  SK_HIP(hipSetDevice(h->device));
  SK_TRY(reserve_impl(h, max_batch, max_samples));          // slot 0 = the handle's full-size workspace; records the shape
  for (int k = 0; k < slots; ++k) SK_TRY(reserve_slot(h, k, max_batch, max_samples));
  return SK_OK;
}

int xt_forward_begin(xt_handle* h, int32_t slot, const void* d_wav, int32_t in_dtype, int64_t wav_ld, const int32_t* h_nsamples, int32_t B, int64_t L,
                     float* d_emb, float* d_logits, void* stream) {
  SK_ENTER(h);
  SK_TRY(check_run(h, B, L));
  SK_CHECK(slot >= 0 && slot < xt_handle::MAX_LANES && h->lane[slot].stream && (slot == 0 || h->lane[slot].covers(B, L)), SK_EWORKSPACE,
           "xt_forward_begin: slot %d has no workspace for %d x %lld samples (xt_reserve_slots)", slot, B, (long long)L);
  SK_CHECK(d_wav && d_emb && wav_ld >= L && (in_dtype == XT_F32 || in_dtype == XT_I16), SK_EARG, "xt_forward_begin: bad buffers");
  SK_CHECK(!h->debug, SK_ESTATE, "xt_forward_begin: debug taps belong to the plain forward");
  Lane& lk = h->lane[slot];
  SK_TRY(enter_stream(h, (hipStream_t)stream));
  StreamScope scope_(h, (hipStream_t)stream);
  SK_HIP(hipEventRecord(lk.fork, (hipStream_t)stream));
  SK_HIP(hipStreamWaitEvent(lk.stream, lk.fork, 0));
  static const int cap = SK_AB_ENV_INT("SIDEKIT_AMD_SLOT_PERSIST_CAP", 1);
  lk.persist_cap = cap;
  BatchMeta m;
  int rc = lane_frontend(h, lk, d_wav, in_dtype == XT_I16 ? 1 : 0, wav_ld, h_nsamples, B, L, m, lk.stream);
  if (rc == SK_OK) rc = lane_trunk(h, lk, m, d_emb, d_logits, lk.stream);
  char err[sizeof(g_err)];
  snprintf(err, sizeof(err), "%s", g_err);
  const bool recorded = hipEventRecord(lk.join, lk.stream) == hipSuccess;   // also after an error: xt_forward_end then orders behind whatever was queued
  if (!recorded) (void)hipStreamSynchronize(lk.stream);
  if (rc != SK_OK) set_error("%s", err);
  return rc;
}

int xt_forward_end(xt_handle* h, int32_t slot, void* stream) {
  SK_ENTER(h);
  SK_CHECK(h && slot >= 0 && slot < xt_handle::MAX_LANES && h->lane[slot].stream, SK_EARG, "xt_forward_end: slot %d was never reserved", slot);
  SK_HIP(hipStreamWaitEvent((hipStream_t)stream, h->lane[slot].join, 0));
  return SK_OK;
}

int xt_forward_features(xt_handle* h, const float* d_feats, const int32_t* h_frames, int32_t B, int32_t T, float* d_emb,
                        float* d_logits, void* stream) {
  SK_ENTER(h);
  SK_TRY(check_run(h, B, (int64_t)(T > 0 ? T - 1 : 0) * (h ? h->fc.hop : 1)));
  SK_CHECK(d_feats && d_emb && T > 0, SK_EARG, "xt_forward_features: bad buffers");
  hipStream_t st = (hipStream_t)stream;
  SK_TRY(enter_stream(h, st));
  StreamScope scope_(h, st);
  BatchMeta m;
  Lane& ln = h->lane[0];
  ln.persist_cap = 0;
  if (ln.stream) SK_HIP(hipStreamWaitEvent(st, ln.join, 0));   // lane 0's workspace: behind whatever its own stream still runs (forward_wav)
  SK_TRY(make_meta(h, ln, h_frames, B, T, false, m, st));
  if (h->cfg.arch == XT_ARCH_HALFRESNET34) {
    m.T = T;  // rows are addressed through the caller's (B, 80, T) strides
    return half_from_feats(h, ln, d_feats, (long)80 * T, T, 1, m, d_emb, d_logits, st);
  }
  float* rows = (float*)ln.ws_feat.p;
  RowSpan rs{m.d_offsets, 0, m.lens, 0, 0};
  hipLaunchKernelGGL(bft_to_rows_kernel, dim3(B), dim3(256), 0, st, d_feats, rows, 80, T, rs);
  SK_HIP(hipGetLastError());
  return tdnn_from_rows(h, ln, rows, m, d_emb, d_logits, st);
}

int xt_features(xt_handle* h, const float* d_wav, int64_t wav_ld, const int32_t* h_nsamples, int32_t B, int64_t L,
                float* d_feats_out, void* stream) {
  SK_ENTER(h);
  SK_TRY(check_run(h, B, L));
  SK_CHECK(d_wav && d_feats_out && wav_ld >= L, SK_EARG, "xt_features: bad buffers");
  hipStream_t st = (hipStream_t)stream;
  SK_TRY(enter_stream(h, st));
  StreamScope scope_(h, st);
  BatchMeta m;
  Lane& ln = h->lane[0];
  if (ln.stream) SK_HIP(hipStreamWaitEvent(st, ln.join, 0));   // lane 0's workspace: behind whatever its own stream still runs (forward_wav)
  SK_TRY(make_meta(h, ln, h_nsamples, B, L, true, m, st));
  float* feat = (float*)ln.ws_feat.p;
  SK_TRY(frontend_rows(h, ln, d_wav, 0, wav_ld, m, feat, st));
  const int T = 1 + (int)(L / h->fc.hop);
  RowSpan rs{m.d_offsets, m.T, m.lens, 0, 0};
  hipLaunchKernelGGL(rows_to_bft_kernel, dim3(B), dim3(256), 0, st, feat, d_feats_out, h->fc.n_out, T, rs);
  SK_HIP(hipGetLastError());
  return SK_OK;
}

int xt_set_norm_embedding(xt_handle* h, int32_t on) {
  SK_CHECK(h, SK_EARG, "null handle");
  h->norm_embedding = on != 0;
  return SK_OK;
}

int xt_set_lanes(xt_handle* h, int32_t lanes) {
  SK_CHECK(h && lanes >= 1 && lanes <= xt_handle::MAX_LANES, SK_EARG, "xt_set_lanes: 1 (serial) .. %d", xt_handle::MAX_LANES);
  SK_ENTER(h);
  h->lanes = lanes;
  if (lanes > 1) {   // size the second lane for every shape reserved while the handle was serial
    SK_HIP(hipSetDevice(h->device));
    for (auto& r : h->reserved) SK_TRY(reserve_side_lanes(h, r.first, r.second));
  }
  return SK_OK;
}

int xt_get_lanes(xt_handle* h) { return h ? h->lanes : 0; }

int xt_set_profile(xt_handle* h, int32_t on) {
  SK_CHECK(h, SK_EARG, "null handle");
  h->profile = (uint32_t)on;
  return SK_OK;
}

int xt_get_profile(xt_handle* h, double* ms, int64_t* launches, int32_t reset) {
  SK_CHECK(h && ms && launches, SK_EARG, "xt_get_profile: null argument");
  SK_HIP(hipSetDevice(h->device));
  for (auto& r : h->prof_recs) {
    SK_HIP(hipEventSynchronize(r.b));
    float t = 0.f;
    SK_HIP(hipEventElapsedTime(&t, r.a, r.b));
    h->prof_ms[r.slot] += t;
    h->prof_n[r.slot] += 1;
    h->prof_pool.push_back(r.a); h->prof_pool.push_back(r.b);
  }
  h->prof_recs.clear();
  for (int i = 0; i < XT_PROF_SLOTS; ++i) { ms[i] = h->prof_ms[i]; launches[i] = h->prof_n[i]; }
  if (reset) for (int i = 0; i < XT_PROF_SLOTS; ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
  return SK_OK;
}

// Kernel-level timing harness for tuning (diagnostic; not used by the product path): runs one trunk
// convolution shape `iters` times on zero-initialised buffers and returns the mean device time.
#ifdef SK_AB
// shape 48 (A/B builds): the layer-1 pair kernel (conv_pair.hip) on random operands; variant bit 0: the first block's in-place shortcut form
static int bench_conv_pair(int32_t B, int32_t T, int32_t iters, int32_t variant, float* ms_out, double* phase_cycles) {
  const size_t act = (size_t)B * T * 80 * 32 * 2, wbytes = 32 * 32 * 9 * 2;
  void *bufs[4] = {nullptr, nullptr, nullptr, nullptr}, *w[3] = {nullptr, nullptr, nullptr}, *zeros = nullptr;
  float *cst = nullptr, *gate = nullptr, *se = nullptr, *colp = nullptr, *edge = nullptr;
  const int tiles = cdiv(T, 8);
  for (auto& p : bufs) SK_HIP(hipMalloc(&p, act));
  for (auto& p : w) SK_HIP(hipMalloc(&p, wbytes));
  SK_HIP(hipMalloc(&zeros, 256)); SK_HIP(hipMemset(zeros, 0, 256));
  SK_HIP(hipMalloc((void**)&cst, 5 * 32 * 4)); SK_HIP(hipMalloc((void**)&gate, (size_t)B * 32 * 4));
  SK_HIP(hipMalloc((void**)&se, (size_t)B * tiles * 4 * 32 * 4)); SK_HIP(hipMalloc((void**)&colp, (size_t)B * tiles * 2 * 32 * 4)); SK_HIP(hipMalloc((void**)&edge, (size_t)B * 6 * 32 * 4));
  {
    uint32_t x = 0x9E3779B9u;
    auto next = [&]() { x = x * 1664525u + 1013904223u; return (float)((x >> 8) & 0xffff) / 32768.f - 1.f; };
    std::vector<uint16_t> hb(act / 2);
    for (int k = 0; k < 2; ++k) {   // O1 (post-ReLU: non-negative) and the block input
      for (auto& v : hb) { const float f = next(); v = f32_to_bf16(f < 0 ? 0.f : f); }
      SK_HIP(hipMemcpy(bufs[k], hb.data(), act, hipMemcpyHostToDevice));
    }
    std::vector<uint16_t> hw(wbytes / 2);
    for (auto& p : w) { for (auto& v : hw) v = f32_to_bf16(0.05f * next()); SK_HIP(hipMemcpy(p, hw.data(), wbytes, hipMemcpyHostToDevice)); }
    std::vector<float> c(5 * 32);
    for (int i = 0; i < 32; ++i) { c[i] = 1.f; c[32 + i] = 0.f; c[64 + i] = 1.f; c[96 + i] = 0.f; c[128 + i] = 0.f; }
    SK_HIP(hipMemcpy(cst, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> gv((size_t)B * 32, 0.5f);
    SK_HIP(hipMemcpy(gate, gv.data(), gv.size() * 4, hipMemcpyHostToDevice));
  }
  ConvPairArgs pa;
  memset(&pa, 0, sizeof(pa));
  pa.C = 32; pa.W = 80; pa.in = bufs[0]; pa.w2pack = w[0]; pa.scale2 = cst; pa.shift2 = cst + 32; pa.gate = gate;
  if (variant & 1) { pa.sc_in = bufs[1]; pa.sc_wpack = w[2]; pa.sc_shift = cst + 128; } else { pa.shortcut = bufs[1]; }
  pa.y_out = bufs[2]; pa.w1pack = w[1]; pa.scale1 = cst + 64; pa.shift1 = cst + 96; pa.o_out = bufs[3];
  pa.se_part = se; pa.col_part = colp; pa.edge = edge; pa.zeros = zeros; pa.lens = Lens{nullptr, T}; pa.B = B; pa.H = T;
  unsigned long long* stamps = nullptr;
  const int nblk = 512;   // at most two persistent workgroups per CU
  if (phase_cycles) { SK_HIP(hipMalloc((void**)&stamps, (size_t)nblk * 128)); SK_HIP(hipMemset(stamps, 0, (size_t)nblk * 128)); pa.stamps = stamps; }
  hipEvent_t e0, e1;
  SK_HIP(hipEventCreate(&e0)); SK_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) SK_TRY(launch_conv_pair(pa, nullptr));
  SK_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) SK_TRY(launch_conv_pair(pa, nullptr));
  SK_HIP(hipEventRecord(e1, nullptr));
  SK_HIP(hipEventSynchronize(e1));
  SK_HIP(hipEventElapsedTime(ms_out, e0, e1));
  *ms_out /= iters;
  if (phase_cycles) {   // mean cycles between consecutive stamps (nine phases) of each workgroup's last item; [9] = the item, [10] = its 100-MHz ticks
    std::vector<unsigned long long> hs((size_t)nblk * 16);
    SK_HIP(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < 11; ++k) phase_cycles[k] = 0;
    int n = 0;
    for (int i = 0; i < nblk; ++i) {
      if (!hs[(size_t)i * 16 + 9]) continue;
      for (int k = 0; k < 9; ++k) phase_cycles[k] += (double)(hs[(size_t)i * 16 + k + 1] - hs[(size_t)i * 16 + k]);
      phase_cycles[9] += (double)(hs[(size_t)i * 16 + 9] - hs[(size_t)i * 16]);
      phase_cycles[10] += (double)hs[(size_t)i * 16 + 15];
      ++n;
    }
    for (int k = 0; k < 11; ++k) phase_cycles[k] /= (n ? n : 1);
    (void)hipFree(stamps);
  }
  for (auto p : bufs) (void)hipFree(p);
  for (auto p : w) (void)hipFree(p);
  (void)hipFree(zeros); (void)hipFree(cst); (void)hipFree(gate); (void)hipFree(se); (void)hipFree(colp); (void)hipFree(edge);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return SK_OK;
}

#endif

int sk_bench_conv(int32_t shape, int32_t dtype, int32_t B, int32_t T, int32_t iters, int32_t variant, float* ms_out, double* phase_cycles) {
  SK_CHECK(ms_out && B > 0 && T > 0 && iters > 0, SK_EARG, "sk_bench_conv: bad arguments");
#ifdef SK_AB
  if (shape == 48) {
    SK_CHECK(dtype == XT_BF16, SK_EARG, "sk_bench_conv: the pair kernel is bf16 only");
    return bench_conv_pair(B, T, iters, variant, ms_out, phase_cycles);   // phase_cycles: 11 doubles here
  }
#endif
  ConvGeom g;
  const int dt = dtype == XT_BF16 ? DT_BF16 : DT_F32;
  SK_TRY(conv_geom(shape, dt, &g));
  const int hin = T, hout = g.stride == 2 ? (T + 1) / 2 : T;
  const size_t in_b = (size_t)B * hin * g.win * g.cin * g.eb, out_b = (size_t)B * hout * (g.win / g.stride) * g.cout * g.eb;
  void *in = nullptr, *out = nullptr, *w = nullptr, *zeros = nullptr; float *sc = nullptr, *sh = nullptr, *se = nullptr;
  SK_HIP(hipMalloc(&in, in_b)); SK_HIP(hipMalloc(&out, out_b)); SK_HIP(hipMalloc(&w, conv_pack_bytes(g) + 4096));
  SK_HIP(hipMalloc(&zeros, 256)); SK_HIP(hipMalloc((void**)&sc, g.cout * 4)); SK_HIP(hipMalloc((void**)&sh, g.cout * 4));
  SK_HIP(hipMalloc((void**)&se, (size_t)B * (hout / g.th + 2) * (g.wm > 4 ? g.wm : 4) * g.cout * 4));
  SK_HIP(hipMemset(zeros, 0, 256));
  {  // random operands (uniform in [-1, 1)): constant fills toggle no bits and let the chip hold a clock real data never sees
    auto fill = [&](void* dst, size_t bytes, float amp) -> int {
      std::vector<uint32_t> hbuf(bytes / 4 + 1);
      uint32_t x = 0x9E3779B9u;
      for (auto& v : hbuf) {
        auto next = [&]() { x = x * 1664525u + 1013904223u; return (float)((x >> 8) & 0xffff) / 32768.f - 1.f; };
        if (g.eb == 2) v = (uint32_t)f32_to_bf16(amp * next()) | ((uint32_t)f32_to_bf16(amp * next()) << 16);
        else v = __builtin_bit_cast(uint32_t, amp * next());
      }
      SK_HIP(hipMemcpy(dst, hbuf.data(), bytes, hipMemcpyHostToDevice));
      return SK_OK;
    };
    SK_TRY(fill(in, in_b, 1.f));
    SK_TRY(fill(w, conv_pack_bytes(g), 0.05f));
    std::vector<float> ones(g.cout, 1.f), zs(g.cout, 0.f);
    SK_HIP(hipMemcpy(sc, ones.data(), g.cout * 4, hipMemcpyHostToDevice));
    SK_HIP(hipMemcpy(sh, zs.data(), g.cout * 4, hipMemcpyHostToDevice));
  }
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.in = in; a.wpack = w; a.scale = sc; a.shift = sh; a.out = out; a.se_part = nullptr; a.zeros = zeros;
  a.lens = Lens{nullptr, T}; a.halvings_in = 0; a.B = B; a.Hin = hin; a.Hout = hout; a.relu = 1; a.dbg = (variant & 7) | ((variant & 32) ? 8 : 0) | ((variant & 64) ? 16 : 0);
  const int nblk = B * cdiv(hout, g.th);
  unsigned long long* stamps = nullptr;
  float *gate = nullptr; void* scut = nullptr; float *colp = nullptr, *edge = nullptr;
  if (variant & 8) {   // statistics-mode epilogue
    SK_HIP(hipMalloc((void**)&colp, (size_t)nblk * 2 * g.cout * 4)); SK_HIP(hipMalloc((void**)&edge, (size_t)B * 6 * g.cout * 4));
    a.se_part = se; a.col_part = colp; a.edge = edge;
  }
  if (variant & 16) {  // residual-mode epilogue
    SK_HIP(hipMalloc((void**)&gate, (size_t)B * g.cout * 4)); SK_HIP(hipMalloc(&scut, out_b));
    {
      std::vector<float> gv((size_t)B * g.cout, 0.5f);
      SK_HIP(hipMemcpy(gate, gv.data(), gv.size() * 4, hipMemcpyHostToDevice));
      SK_HIP(hipMemcpy(scut, in, out_b < in_b ? out_b : in_b, hipMemcpyDeviceToDevice));   // random shortcut rows
    }
    a.gate = gate; a.shortcut = scut;
  }
  if (phase_cycles) { SK_HIP(hipMalloc((void**)&stamps, (size_t)nblk * 64)); SK_HIP(hipMemset(stamps, 0, (size_t)nblk * 64)); a.stamps = stamps; }
  hipEvent_t e0, e1;
  SK_HIP(hipEventCreate(&e0)); SK_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) SK_TRY(launch_conv(shape, dt, a, nullptr));
  SK_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters; ++i) SK_TRY(launch_conv(shape, dt, a, nullptr));
  SK_HIP(hipEventRecord(e1, nullptr));
  SK_HIP(hipEventSynchronize(e1));
  SK_HIP(hipEventElapsedTime(ms_out, e0, e1));
  *ms_out /= iters;
  if (phase_cycles) {  // mean cycles between consecutive stamps over all workgroups of the last launch; [7] = whole workgroup
    std::vector<unsigned long long> hs((size_t)nblk * 8);
    SK_HIP(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; ++k) phase_cycles[k] = 0;
    int n = 0;
    for (int i = 0; i < nblk; ++i) {
      if (!hs[(size_t)i * 8 + 6]) continue;
      for (int k = 0; k < 6; ++k) phase_cycles[k] += (double)(hs[(size_t)i * 8 + k + 1] - hs[(size_t)i * 8 + k]);
      phase_cycles[7] += (double)(hs[(size_t)i * 8 + 6] - hs[(size_t)i * 8]);
      phase_cycles[6] += (double)hs[(size_t)i * 8 + 7];   // 100 MHz ticks of the same span
      ++n;
    }
    for (int k = 0; k < 8; ++k) phase_cycles[k] /= (n ? n : 1);
    (void)hipFree(stamps);
  }
  if (colp) (void)hipFree(colp); if (edge) (void)hipFree(edge); if (gate) (void)hipFree(gate); if (scut) (void)hipFree(scut);
  (void)hipFree(in); (void)hipFree(out); (void)hipFree(w); (void)hipFree(zeros); (void)hipFree(sc); (void)hipFree(sh); (void)hipFree(se);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return SK_OK;
}

int xt_set_debug(xt_handle* h, int32_t on) {
  SK_CHECK(h, SK_EARG, "null handle");
  h->debug = on != 0;
  return SK_OK;
}

int xt_debug_tap(xt_handle* h, const char* name, void* h_dst, size_t capacity, size_t* bytes) {
  SK_CHECK(h && name && bytes, SK_EARG, "xt_debug_tap: null argument");
  auto it = h->taps.find(name);
  SK_CHECK(it != h->taps.end(), SK_EARG, "xt_debug_tap: no tap named %s (xt_set_debug before forward?)", name);
  *bytes = it->second.bytes;
  if (!h_dst) return SK_OK;
  SK_CHECK(capacity >= it->second.bytes, SK_EARG, "xt_debug_tap: buffer too small (%zu < %zu)", capacity, it->second.bytes);
  SK_HIP(hipDeviceSynchronize());
  SK_HIP(hipMemcpy(h_dst, it->second.buf.p, it->second.bytes, hipMemcpyDeviceToHost));
  return SK_OK;
}

}  // extern "C"
