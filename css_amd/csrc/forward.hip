// Whole-model forward: Conformer.forward (css/models/conformer.py:83-104) from the feature
// tensor to the ReLU masks as ONE host call.  It only sequences the kernels of this library
// on the caller's stream (no allocation, no synchronisation), so the Python plugin pays one
// ctypes call per forward instead of ~180, and the launch is hipGraph-capturable.
#include "common.hpp"

#include <stdlib.h>
#include <vector>

namespace {

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Workspace {
  float *mean, *rinv, *x0, *x1;
  unsigned short *ao_bf, *qkv_bf;
  unsigned short *a_bf, *h_bf;  // generic (d_model != 256) path only
  void* conv_ws;
  size_t total;
};

Workspace carve(const csship_conformer_weights* w, int B, int T, void* base) {
  const size_t M = (size_t)B * T, D = w->d_model;
  unsigned char* p = reinterpret_cast<unsigned char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    void* r = p ? p + off : nullptr;
    off += align256(bytes);
    return r;
  };
  Workspace ws;
  ws.mean = (float*)take((size_t)B * w->idim * 4);
  ws.rinv = (float*)take((size_t)B * w->idim * 4);
  ws.x0 = (float*)take(M * D * 4);
  ws.x1 = (float*)take(M * D * 4);
  ws.ao_bf = (unsigned short*)take(M * D * 2);
  ws.qkv_bf = (unsigned short*)take(M * 3 * D * 2);
  ws.a_bf = w->generic ? (unsigned short*)take(M * D * 2) : nullptr;
  ws.h_bf = w->generic ? (unsigned short*)take(M * (size_t)w->d_ffn * 2) : nullptr;
  ws.conv_ws = w->bn_mode ? take(csship_convmod_workspace(B, T, (int)D)) : nullptr;
  ws.total = off;
  return ws;
}

#define TRY(expr)              \
  do {                         \
    const int rc_ = (expr);    \
    if (rc_ != CSSHIP_OK) return rc_; \
  } while (0)

// C = epi(A @ W^T + bias), row-major bf16 weights with K padded to 64: the batched-GEMM kernel where its loader applies,
// else the classic tiled GEMM
int tiled_linear(const void* A, int a_mode, const uint16_t* W, const float* bias, const float* res, void* C, int M, int N,
                 int K, int lda, int epi, float alpha, csship_stream_t s) {
  const int ldw = (K + 63) / 64 * 64;
  if (a_mode != CSSHIP_A_F32_MVN && (K % 8) == 0 && (lda % 8) == 0 && !(epi & CSSHIP_EPI_LAYERNORM)) {
    // the batched-GEMM kernel (descriptor loads issued back to back, XCD-aware tile order) is 1.3-1.9x faster than the
    // classic tiled kernel on these shapes (12800 x 512 x 2048: 57 vs 105 us)
    csship_bgemm_params q{};
    q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldw; q.ldc = N; q.ldaux = N;
    q.a_f32 = a_mode == CSSHIP_A_F32; q.batch1 = q.batch2 = 1; q.splits = 1;
    q.alpha = (epi & CSSHIP_EPI_RESIDUAL) ? alpha : 1.f;  // the tiled kernel applies alpha to the residual form only
    q.epilogue = ((epi & CSSHIP_EPI_OUT_BF16) ? CSSHIP_BEPI_OUT_BF16 : 0) | ((epi & CSSHIP_EPI_RELU) ? CSSHIP_BEPI_RELU : 0) |
                 ((epi & CSSHIP_EPI_RESIDUAL) ? (CSSHIP_BEPI_ADD | CSSHIP_BEPI_SCALE_BIAS) : 0);
    return csship_bgemm(A, W, C, bias, (epi & CSSHIP_EPI_RESIDUAL) ? res : nullptr, &q, nullptr, 0, s);
  }
  csship_gemm_params p{};
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = N; p.ldr = N;
  p.a_mode = a_mode; p.epilogue = epi; p.rows_per_batch = 1; p.alpha = alpha;
  return csship_gemm(A, W, bias, res, C, nullptr, nullptr, nullptr, nullptr, &p, s);
}

// Any-width encoder (e.g. Conformer-large: d_model 512, 8 heads, ffn 2048): the same operators as the tuned
// d_model = 256 path, unfused, built from the size-generic kernels.  Weights are ROW-MAJOR bf16 here.
int forward_generic(const csship_conformer_weights* w, const float* f, int64_t ldf, int B, int T, float* masks,
                    int64_t ld_masks, float* enc_out, const Workspace& ws, csship_stream_t stream) {
  const int M = B * T, D = w->d_model, FF = w->d_ffn;
  const int BF = CSSHIP_EPI_OUT_BF16;
  TRY(csship_mvn_stats(f, B, T, w->idim, ldf, -INFINITY, 1.1920928955078125e-07f, ws.mean, ws.rinv, stream));
  {
    csship_gemm_params p{};
    p.M = M; p.N = D; p.K = w->idim; p.lda = (int)ldf; p.ldw = w->idim_pad; p.ldc = D; p.ldr = D;
    p.a_mode = CSSHIP_A_F32_MVN; p.epilogue = 0; p.rows_per_batch = T;
    TRY(csship_gemm(f, w->embed_w, w->embed_b, nullptr, ws.x1, ws.mean, ws.rinv, nullptr, nullptr, &p, stream));
    TRY(csship_layernorm_relu(ws.x1, M, D, w->embed_ln_g, w->embed_ln_b, 1, nullptr, ws.x0, stream));
  }
  float *x = ws.x0, *y = ws.x1;
  // one FeedForward: x <- x + 0.5 FFN(LN(x)) in place, or (fin) out <- LayerNorm_fin(x + 0.5 FFN(LN(x)))
  auto feed_forward = [&](const csship_ffn_weights& f, float* xin, float* out, const float* fin_g, const float* fin_b) -> int {
    if (w->ffn_mode == 2)
      return csship_ffn_mx(xin, out, f.ln_g, f.ln_b, reinterpret_cast<const uint8_t*>(f.w1), f.w1_scale, f.b1,
                           reinterpret_cast<const uint8_t*>(f.w2), f.w2_scale, f.b2, fin_g, fin_b, M, D, FF, 0.5f, stream);
    if (w->ffn_mode == 1)
      return csship_ffn_wide(xin, out, f.ln_g, f.ln_b, f.w1, f.b1, f.w2, f.b2, fin_g, fin_b, M, D, FF, 0.5f, stream);
    TRY(csship_layernorm(xin, M, D, f.ln_g, f.ln_b, ws.a_bf, nullptr, stream));
    TRY(tiled_linear(ws.a_bf, CSSHIP_A_BF16, f.w1, f.b1, nullptr, ws.h_bf, M, FF, D, D, CSSHIP_EPI_RELU | BF, 0.f, stream));
    TRY(tiled_linear(ws.h_bf, CSSHIP_A_BF16, f.w2, f.b2, xin, xin, M, D, FF, FF, CSSHIP_EPI_RESIDUAL, 0.5f, stream));
    if (fin_g) return csship_layernorm(xin, M, D, fin_g, fin_b, nullptr, out, stream);
    return CSSHIP_OK;
  };
  for (int l = 0; l < w->n_layer; ++l) {
    const csship_layer_weights& L = w->layers[l];
    TRY(feed_forward(L.ffn_in, x, x, nullptr, nullptr));
    if (w->lin_mode == 1) {
      TRY(csship_ln_linear_wide(x, L.mha.ln_g, L.mha.ln_b, L.mha.wqkv, L.mha.bqkv, ws.qkv_bf, M, D, 3 * D, stream));
    } else {
      TRY(csship_layernorm(x, M, D, L.mha.ln_g, L.mha.ln_b, ws.a_bf, nullptr, stream));
      TRY(tiled_linear(ws.a_bf, CSSHIP_A_BF16, L.mha.wqkv, L.mha.bqkv, nullptr, ws.qkv_bf, M, 3 * D, D, D, BF, 0.f, stream));
    }
    TRY(csship_mha_relpos_ex(ws.qkv_bf, w->pe_k, w->pe_q_min, w->pe_q_max, ws.ao_bf, B, T, w->n_head,
                                   (w->chain_flags & CSSHIP_FWD_Q_LOG2E) ? CSSHIP_MHA_Q_LOG2E : 0, stream));
    if (w->lin_mode == 1) {
      TRY(csship_linear_res_wide(ws.ao_bf, L.mha.wo, L.mha.bo, x, x, M, D, stream));
    } else {
      TRY(tiled_linear(ws.ao_bf, CSSHIP_A_BF16, L.mha.wo, L.mha.bo, x, x, M, D, D, D, CSSHIP_EPI_RESIDUAL, 1.f, stream));
    }
    TRY(csship_convmod(x, y, &L.conv, B, T, D, w->kernel_size, w->bn_mode, ws.conv_ws, stream));
    { float* t = x; x = y; y = t; }
    TRY(feed_forward(L.ffn_out, x, y, L.ln_g, L.ln_b));   // FFN_out + the block-final LayerNorm (conformer.py:358-360)
    { float* t = x; x = y; y = t; }
  }
  if (enc_out && hipMemcpyAsync(enc_out, x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess)
    return CSSHIP_ELAUNCH;
  if (!masks) return CSSHIP_OK;
  csship_gemm_params p{};
  p.M = M; p.N = w->n_out; p.K = D; p.lda = D; p.ldw = (D + 63) / 64 * 64; p.ldc = (int)ld_masks; p.ldr = (int)ld_masks;
  p.a_mode = CSSHIP_A_F32; p.epilogue = CSSHIP_EPI_RELU; p.rows_per_batch = 1;
  return csship_gemm(x, w->head_w, w->head_b, nullptr, masks, nullptr, nullptr, nullptr, nullptr, &p, stream);
}

}  // namespace

unsigned long long* g_csship_stamps = nullptr;

// ---- per-launch timeline (see common.hpp)
bool g_csship_timeline_on = false;
int g_csship_timeline_depth = 0;
namespace {
struct TlRec { int tag; hipEvent_t e0, e1; bool closed; };
std::vector<TlRec> g_tl;
}  // namespace
void csship_timeline_push(int tag, hipStream_t s, bool stop) {
  if (!stop) {
    TlRec r{tag, nullptr, nullptr, false};
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, s);
    g_tl.push_back(r);
  } else {
    for (size_t i = g_tl.size(); i-- > 0;)
      if (g_tl[i].tag == tag && !g_tl[i].closed) {
        (void)hipEventRecord(g_tl[i].e1, s);
        g_tl[i].closed = true;
        break;
      }
  }
}
// Diagnostic only (not part of include/csship.h's product surface): enable = 1 starts a fresh timeline, 0 stops it.
extern "C" void csship_debug_timeline(int enable) {
  for (auto& r : g_tl) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
  g_tl.clear();
  g_csship_timeline_depth = 0;
  g_csship_timeline_on = enable != 0;
}
// `pairs` empty brackets on `stream` (tag CSSHIP_TL_EMPTY): what a bracket itself adds to the launch it times.
extern "C" void csship_debug_timeline_calibrate(csship_stream_t stream_, int pairs) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!g_csship_timeline_on) return;
  for (int i = 0; i < pairs; ++i) {
    csship_timeline_push(CSSHIP_TL_EMPTY, stream, false);
    csship_timeline_push(CSSHIP_TL_EMPTY, stream, true);
  }
}
// After the caller has synchronised the stream: total milliseconds and number of launches recorded under `tag`.
extern "C" int csship_debug_timeline_read(int tag, double* total_ms, int* count) {
  double t = 0.0; int n = 0;
  for (auto& r : g_tl)
    if (r.tag == tag && r.closed) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { t += ms; ++n; }
    }
  if (total_ms) *total_ms = t;
  if (count) *count = n;
  return CSSHIP_OK;
}
// Diagnostic only (not part of include/csship.h): device buffer of [n_workgroups][8] u64 that instrumented
// kernels fill with s_memrealtime phase stamps; pass NULL to switch stamping off.
extern "C" void csship_debug_stamps(void* device_buffer) {
  g_csship_stamps = reinterpret_cast<unsigned long long*>(device_buffer);
}

extern "C" int csship_version(void) { return 100; }

extern "C" const char* csship_strerror(int rc) {
  switch (rc) {
    case CSSHIP_OK: return "ok";
    case CSSHIP_EBADSHAPE: return "csship: unsupported size or stride";
    case CSSHIP_EALIGN: return "csship: pointer or stride not aligned as required";
    case CSSHIP_EUNSUPPORTED: return "csship: dimension/feature combination not built";
    case CSSHIP_ELAUNCH: return "csship: kernel launch failed (hipGetLastError)";
    case CSSHIP_ENULL: return "csship: required pointer is NULL";
    case CSSHIP_EWORKSPACE: return "csship: workspace missing or too small";
    default: return "csship: unknown error";
  }
}

extern "C" size_t csship_conformer_workspace(const csship_conformer_weights* w, int32_t B, int32_t T) {
  if (!w || B <= 0 || T <= 0) return 0;
  return carve(w, B, T, nullptr).total;
}

// Diagnostic only (CSSHIP_DEBUG_NODEP=N, eager launches, results are WRONG): the encoder's launches are dealt
// round-robin to N internal streams that only fork from / join the caller's stream, i.e. the dependencies between
// consecutive launches are cut.  What the step then takes is the ceiling of any scheme that overlaps one launch's ramp /
// tail with its neighbours (persistent kernels, cross-launch prefetch): DESIGN.md, "launch boundaries".
namespace {
struct NoDep {
  int n = 0, k = 0;
  hipStream_t s[8];
  hipEvent_t fork, join[8];
  hipStream_t base = nullptr;
  bool init() {
    static const int want = [] { const char* e = getenv("CSSHIP_DEBUG_NODEP"); return e ? atoi(e) : 0; }();
    if (want <= 0) return false;
    if (n == 0) {
      n = want > 8 ? 8 : want;
      for (int i = 0; i < n; ++i) { (void)hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); (void)hipEventCreateWithFlags(&join[i], hipEventDisableTiming); }
      (void)hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    }
    return true;
  }
  void begin(hipStream_t b) {
    base = b; k = 0;
    (void)hipEventRecord(fork, b);
    for (int i = 0; i < n; ++i) (void)hipStreamWaitEvent(s[i], fork, 0);
  }
  csship_stream_t next() { return reinterpret_cast<csship_stream_t>(s[k++ % n]); }
  void end() {
    for (int i = 0; i < n; ++i) { (void)hipEventRecord(join[i], s[i]); (void)hipStreamWaitEvent(base, join[i], 0); }
  }
};
NoDep g_nodep;
}  // namespace

extern "C" int csship_conformer_forward(const csship_conformer_weights* w, const float* f, int64_t ldf, int32_t B,
                                        int32_t T, float* masks, int64_t ld_masks, float* enc_out, void* workspace,
                                        size_t workspace_bytes, csship_stream_t stream) {
  if (!w || !f || (!masks && !enc_out) || !workspace || !w->layers) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || ldf < w->idim || (masks && ld_masks < w->n_out)) return CSSHIP_EBADSHAPE;
  if (w->d_model / w->n_head != 64 || w->kernel_size != 33 || w->idim_pad < w->idim) return CSSHIP_EUNSUPPORTED;
  if (w->generic) {
    if ((w->d_model != 256 && w->d_model != 512) || (w->d_ffn % 64) != 0 || (w->idim_pad % 64) != 0) return CSSHIP_EUNSUPPORTED;
    if (w->ffn_mode < 0 || w->ffn_mode > 2 || (w->ffn_mode != 0 && (w->d_ffn % 256) != 0)) return CSSHIP_EUNSUPPORTED;
    if (w->lin_mode < 0 || w->lin_mode > 1) return CSSHIP_EUNSUPPORTED;
  } else if (w->d_model != 256 || (w->d_ffn % 256) != 0 || (w->idim_pad % 256) != 0) {
    return CSSHIP_EUNSUPPORTED;
  }
  if (!aligned16(workspace)) return CSSHIP_EALIGN;
  Workspace ws = carve(w, B, T, workspace);
  if (ws.total > workspace_bytes) return CSSHIP_EWORKSPACE;
  if (w->generic) return forward_generic(w, f, ldf, B, T, masks, ld_masks, enc_out, ws, stream);
  const int M = B * T, D = w->d_model, FF = w->d_ffn;

  // input MVN over time (conformer.py:95) folded into the embed GEMM's A load
  {
    TimelineScope tl(CSSHIP_TL_MVN, as_stream(stream));
    TRY(csship_mvn_stats(f, B, T, w->idim, ldf, -INFINITY, 1.1920928955078125e-07f, ws.mean, ws.rinv, stream));
  }
  {
    TimelineScope tl(CSSHIP_TL_EMBED, as_stream(stream));
    TRY(csship_embed(f, ldf, ws.mean, ws.rinv, w->embed_w, w->embed_b, w->embed_ln_g, w->embed_ln_b, ws.x0, B, T, w->idim,
                     D, stream));
  }
  float *x = ws.x0, *y = ws.x1;
  // fuse_conv 0 = default: out-proj + conv module as one kernel in eval mode, separate kernels when BatchNorm needs
  // batch statistics (a grid-wide reduction sits between the convolution and the ReLU then)
  bool causal = false;   // (conformer.py:265-273: every block or none) the causal module runs through csship_convmod
  for (int l = 0; l < w->n_layer; ++l) causal = causal || w->layers[l].conv.causal != 0;
  const int conv_mode = causal ? 4 : (w->fuse_conv != 0 ? w->fuse_conv : (w->bn_mode == 0 ? 3 : 4));
  if (w->bn_mode == 0 && conv_mode == 1) {
    // optional (measured slower at B=32 x T=400: 83 vs 75 us per block): 3 launches per block.  Row-local operators are chained per 64-row panel (csship_encoder_chain):
    //   [conv back half(l-1) -> FFN_out(l-1) + block LN] -> FFN_in(l) -> pre-LN + q/k/v projection(l)
    float* gbuf = y;  // glu(LN_conv(x)) rows handed from the output projection to the next chain
    TRY(csship_encoder_chain(x, x, nullptr, nullptr, T, nullptr, nullptr, nullptr, &w->layers[0].ffn_in,
                             w->layers[0].mha.ln_g, w->layers[0].mha.ln_b, w->layers[0].mha.wqkv,
                             w->layers[0].mha.bqkv, ws.qkv_bf, M, D, FF, 3 * D, w->chain_flags & ~CSSHIP_FWD_Q_LOG2E, stream));
    for (int l = 0; l < w->n_layer; ++l) {
      const csship_layer_weights& L = w->layers[l];
      TRY(csship_mha_relpos_ex(ws.qkv_bf, w->pe_k, w->pe_q_min, w->pe_q_max, ws.ao_bf, B, T, w->n_head,
                                   (w->chain_flags & CSSHIP_FWD_Q_LOG2E) ? CSSHIP_MHA_Q_LOG2E : 0, stream));
      TRY(csship_outproj_glu(ws.ao_bf, L.mha.wo, L.mha.bo, x, x, &L.conv, gbuf, M, D, stream));
      if (l + 1 < w->n_layer) {
        const csship_layer_weights& N = w->layers[l + 1];
        TRY(csship_encoder_chain(x, x, gbuf, &L.conv, T, &L.ffn_out, L.ln_g, L.ln_b, &N.ffn_in, N.mha.ln_g, N.mha.ln_b,
                                 N.mha.wqkv, N.mha.bqkv, ws.qkv_bf, M, D, FF, 3 * D, w->chain_flags & ~CSSHIP_FWD_Q_LOG2E, stream));
      } else {
        TRY(csship_encoder_chain(x, x, gbuf, &L.conv, T, &L.ffn_out, L.ln_g, L.ln_b, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, M, D, FF, 0, 0, stream));
      }
    }
  } else {
    // default: 3 launches per block: chain (FFN_out(l-1) + LN -> FFN_in(l) -> q/k/v projection(l), per 64-row
    // panel), attention, out-proj + conv module
    const bool nodep = g_nodep.init();
    const csship_stream_t stream0 = stream;
    if (nodep) g_nodep.begin(as_stream(stream0));
    for (int l = 0; l < w->n_layer; ++l) {
      const csship_layer_weights& L = w->layers[l];
      const csship_layer_weights* P = l > 0 ? &w->layers[l - 1] : nullptr;
      {
        if (nodep) stream = g_nodep.next();
        TimelineScope tl(P ? CSSHIP_TL_CHAIN : CSSHIP_TL_CHAIN_EDGE, as_stream(stream));
        TRY(csship_encoder_chain(x, x, nullptr, nullptr, T, P ? &P->ffn_out : nullptr, P ? P->ln_g : nullptr,
                                 P ? P->ln_b : nullptr, &L.ffn_in, L.mha.ln_g, L.mha.ln_b, L.mha.wqkv, L.mha.bqkv,
                                 ws.qkv_bf, M, D, FF, 3 * D, w->chain_flags & ~CSSHIP_FWD_Q_LOG2E, stream));
      }
      {
        if (nodep) stream = g_nodep.next();
        TimelineScope tl(CSSHIP_TL_ATTN, as_stream(stream));
        TRY(csship_mha_relpos_ex(ws.qkv_bf, w->pe_k, w->pe_q_min, w->pe_q_max, ws.ao_bf, B, T, w->n_head,
                                   (w->chain_flags & CSSHIP_FWD_Q_LOG2E) ? CSSHIP_MHA_Q_LOG2E : 0, stream));
      }
      if (w->bn_mode == 0 && conv_mode == 2) {
        // optional (28.4 vs 27.2 us): x += out-proj(attn); g = glu(LN_conv(x)) in the same epilogue; conv back half from g
        TRY(csship_outproj_glu(ws.ao_bf, L.mha.wo, L.mha.bo, x, x, &L.conv, y, M, D, stream));
        TRY(csship_conv_from_glu(x, y, x, &L.conv, B, T, D, w->kernel_size, stream));
      } else if (w->bn_mode == 0 && conv_mode == 3) {
        // one kernel: x2 = x + out-proj(attn) and the conv module on it (256 workgroups at T = 400)
        if (nodep) stream = g_nodep.next();
        TimelineScope tl(CSSHIP_TL_OUTPROJ_CONV, as_stream(stream));
        TRY(csship_outproj_conv(ws.ao_bf, L.mha.wo, L.mha.bo, x, y, &L.conv, B, T, D, w->kernel_size, stream));
        { float* t = x; x = y; y = t; }
      } else {
        TRY(csship_panel_linear(ws.ao_bf, CSSHIP_PRO_BF16, nullptr, nullptr, L.mha.wo, L.mha.bo, x, D, 1.f, x, D, M, D,
                                D, CSSHIP_EPI_RESIDUAL, stream));
        TRY(csship_convmod(x, y, &L.conv, B, T, D, w->kernel_size, w->bn_mode, ws.conv_ws, stream));
        { float* t = x; x = y; y = t; }
      }
    }
    if (nodep) { g_nodep.end(); stream = stream0; }
    const csship_layer_weights& L = w->layers[w->n_layer - 1];
    TimelineScope tl(CSSHIP_TL_CHAIN_EDGE, as_stream(stream));
    // the last FeedForward + block LayerNorm writes the encoder output where the caller wants it (no copy afterwards)
    float* const xo = enc_out ? enc_out : x;
    TRY(csship_encoder_chain(x, xo, nullptr, nullptr, T, &L.ffn_out, L.ln_g, L.ln_b, nullptr, nullptr, nullptr, nullptr,
                             nullptr, nullptr, M, D, FF, 0, 0, stream));
    x = xo;
  }
  if (enc_out && x != enc_out) {
    if (hipMemcpyAsync(enc_out, x, (size_t)M * D * sizeof(float), hipMemcpyDeviceToDevice, as_stream(stream)) !=
        hipSuccess)
      return CSSHIP_ELAUNCH;
  }
  if (!masks) return CSSHIP_OK;  // the caller continues from enc_out (csship_head_pit: head + mask application + PIT-MSE)
  // mask head: ReLU(Linear(d_model -> num_bins*(K+1)))         (conformer.py:98-100)
  TimelineScope tl(CSSHIP_TL_HEAD, as_stream(stream));
  TRY(csship_panel_linear(x, CSSHIP_PRO_F32, nullptr, nullptr, w->head_w, w->head_b, nullptr, 0, 0.f, masks,
                          ld_masks, M, w->n_out, D, CSSHIP_EPI_RELU, stream));
  return CSSHIP_OK;
}
