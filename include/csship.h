/* csship.h -- C ABI of libcsship.so, the MI355X (gfx950) HIP library behind css_amd.
 *
 * The reference (desh2608/css) has no FFI: its hot path sits behind Python plugin
 * registries (css/models/__init__.py:19-27, css/objectives/__init__.py:19-23) and
 * executor classes.  This header is the build-defined boundary SURVEY.md 8(b)
 * prescribes: one extern "C" entry per kernel family, plain pointers and sizes, no
 * torch types.  The Python plugin (css_amd/) binds it with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name says host; the caller owns
 *     all memory including the workspace; the library never allocates, frees or
 *     retains device memory and never synchronises the host;
 *   - kernels are enqueued on `stream` (a hipStream_t; NULL = default stream);
 *   - return value 0 on success, a negative CSSHIP_E* code otherwise (never throws);
 *   - "bf16" buffers are uint16_t holding the upper 16 bits of an IEEE float
 *     (round-to-nearest-even);
 *   - STFT geometry is fixed at frame_len 512 / hop 256 (conf/config_*.yaml:9-10).
 */
#ifndef CSSHIP_H
#define CSSHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* csship_stream_t; /* hipStream_t */

enum {
  CSSHIP_OK = 0,
  CSSHIP_EBADSHAPE = -1,    /* a size/stride the kernels do not support */
  CSSHIP_EALIGN = -2,       /* pointer or stride not aligned as required */
  CSSHIP_EUNSUPPORTED = -3, /* feature/dimension combination not built */
  CSSHIP_ELAUNCH = -4,      /* hipGetLastError() reported a launch failure */
  CSSHIP_ENULL = -5,        /* required pointer is NULL */
  CSSHIP_EWORKSPACE = -6    /* workspace too small */
};

int csship_version(void);
const char* csship_strerror(int rc);
/* Limits of the tuned kernel forms, so that callers gate on the kernels' own constants instead of copies of them:
 * the widest FeedForward the static row-panel forms take (csship_encoder_chain's static form, csship_ffn_bwd_ln), and
 * the longest chunk csship_attn_pos_bwd(_ds) takes (longer chunks: CSSHIP_EUNSUPPORTED; the tiled products serve them). */
int csship_ffn_static_max_ffn(void);
int csship_attn_pos_bwd_max_t(void);

/* ------------------------------------------------------------------ dropout (training mode)
 * The reference applies nn.Dropout(p) at: the embedding (css/models/conformer.py:127, css/models/blstm.py:147), the
 * attention probabilities and the attention output (conformer.py:249,254), the end of the ConvModule (:302), both
 * FeedForward linears (:317,319) and every BLSTM layer (blstm.py:158).  Here a mask is a pure function of
 * (seed, stream, element index): elements 4q .. 4q+3 (row-major indices in the dropped tensor) share the words
 * w0 = hash32(hash32(seed ^ stream*0x9E3779B9) + q) and w1 = hash_derive(w0); element 4q + f survives iff the 16-bit
 * field f of (w0 low, w0 high, w1 low, w1 high) is >= round(p*65536); survivors are scaled by
 * 65536 / (65536 - round(p*65536)).  hash32 is the "lowbias32" mixer, hash_derive one 24-bit multiply + shift-add +
 * xorshift (css_amd/csrc/common.hpp; oracle/dropout.py is their numpy twin).  Entry points that take a cfg index whole
 * groups of four (their column counts are multiples of 4).  `seed` points to ONE uint32 in DEVICE memory (read by the kernels, so a captured step can be
 * replayed with new masks); a NULL cfg pointer, a NULL seed or p <= 0 mean "no dropout".  Forward and backward
 * kernels take the same cfg and regenerate the same mask. */
typedef struct csship_dropout_cfg_s {
  const uint32_t* seed; /* device */
  uint32_t stream;      /* dropout site id (the host gives every site of every layer its own) */
  float p;
} csship_dropout_cfg;
/* y[r, c] = x[r, c] * keep(r*cols + c) / (1 - p), f32 [rows, cols] with row strides ldx / ldy (y may alias x). */
int csship_dropout_apply(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int32_t cols,
                         const csship_dropout_cfg* cfg, csship_stream_t stream);
/* out = x + alpha * dropout(y): the residual add behind a dropped branch (f32 [rows, cols], dense). */
int csship_dropout_add(const float* x, const float* y, float* out, float alpha, int64_t rows, int32_t cols,
                       const csship_dropout_cfg* cfg, csship_stream_t stream);
/* keep[e] = 1 / 0 (uint8 [n]): the mask itself, for tests and for callers that want to inspect it. */
int csship_dropout_mask(uint8_t* keep, int64_t n, const csship_dropout_cfg* cfg, csship_stream_t stream);

/* ------------------------------------------------------------------ features
 * Replaces STFT.forward (css/executor/feature.py:45-82): 512-point Hann-windowed
 * real DFT of every frame, hop 256, no centre padding, T = (n_samples-512)/256+1.
 * Signals are addressed as audio[s0*stride0 + s1*stride1 + i], s = s0*n1 + s1, so
 * both a dense [B,D,n] batch and the overlapping unfold() windows of
 * css/executor/separator.py:67 are expressible without a copy.
 * Outputs (each may be NULL) are [S, T, 257] float32, frame-major ("TF" layout). */
int csship_stft(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1,
                int64_t n_samples, float* mag, float* re, float* im, float* pha,
                csship_stream_t stream);
/* The same with ld_out >= 257 floats between the frames of every output ([S, T, ld_out] storage, 257 valid columns):
 * ld_out = 260 gives 16-byte aligned rows to the consumers that read 16-byte pieces of a row (csship_head_pit,
 * csship_mvn_stats, csship_embed); the pad columns are not written. */
int csship_stft_ld(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1,
                   int64_t n_samples, float* mag, float* re, float* im, float* pha, int64_t ld_out,
                   csship_stream_t stream);

/* Log-power variant of the same transform: logpow[s, t, k] = log(max(|X[s, t, k]|^2, eps)) (natural log), the
 * build-defined stand-in for the reference's training-time features (Lhotse Spectrogram: third-party, unpinned;
 * SURVEY.md 8c).  Same addressing and output layout as csship_stft; eps > 0. */
int csship_stft_logpower(const float* audio, int64_t n0, int64_t stride0, int64_t n1, int64_t stride1,
                         int64_t n_samples, float eps, float* logpow, csship_stream_t stream);

/* ------------------------------------------------------------------ on-device training mixtures
 * One call builds a batch of B windows of n samples the way css/datasets/separation.py:163-267 builds them (two
 * utterances, the second delayed and both zero padded to the mixture length, summed, cut into windows; augmentations on
 * the mixture only: a sparse 3-tap impulse response and additive noise at a drawn SNR - stand-ins for Lhotse's
 * ReverbWithImpulseResponse / CutMix, which are third-party and unpinned).  pool: f32 [n_utt, pool_len] device-resident
 * utterances; noise_pool: f32 [n_noise, noise_len] or NULL.  params: DEVICE array of B records of 16 x 4 bytes
 *   int32 u1, u2, w0, delay, len1, len2, rir_on, d1, d2, noise_on, nu, n0;  float h0, h1, h2, snr_db
 * (window start w0 on the mixture's time line; source k valid for 0 <= t < len_k of its own time line; impulse
 * response y[t] = h0 m[t] + h1 m[t - d1] + h2 m[t - d2]; noise segment noise_pool[nu][n0 .. n0 + n), gain such that
 * 10 log10(E_mix / E_noise) = snr_db over the window).  Outputs: f32 [B, ld] rows (ld >= n, ld % 4 == 0).
 * The host draws the parameters; every sample is touched on the device only. */
size_t csship_mix_workspace(int32_t B);
int csship_mix_batch(const float* pool, int64_t pool_len, int32_t n_utt, const float* noise_pool, int64_t noise_len,
                     int32_t n_noise, const void* params, int32_t B, int32_t n, float* mix, float* source1,
                     float* source2, int64_t ld, void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* Per-(batch, column) mean and 1/(unbiased std + eps) over the time axis of x[B,T,F]
 * (row stride ldx >= F).  Replaces the statistics of css/models/conformer.py:95,
 * css/models/blstm.py:80 and css/executor/feature.py:169-171 (clamp_min applies
 * torch.clamp(x, min=clamp_min) first, as the feature extractor does; pass -INFINITY
 * for no clamp). */
int csship_mvn_stats(const float* x, int64_t B, int64_t T, int64_t F, int64_t ldx,
                     float clamp_min, float eps, float* mean, float* rinv,
                     csship_stream_t stream);

/* out[b, f, t] = (max(x[b, t, f], clamp_min) - mean[b, f]) * rinv[b, f]   (mean/rinv NULL:
 * plain transpose).  Converts TF layout to the reference's [B, F, T] layout
 * (css/executor/feature.py:156-175 returns mag/f/re/im as B x F x T).  out row b has
 * ld_out_b floats between batches and the f rows are written at row offset f_off. */
int csship_transpose_norm(const float* x, int64_t B, int64_t T, int64_t F, int64_t ldx,
                          float clamp_min, const float* mean, const float* rinv,
                          float* out, int64_t out_batch_stride, int64_t f_off,
                          csship_stream_t stream);

/* Inter-channel phase difference with circular-mean removal, IPDFeature.forward
 * (css/executor/feature.py:105-134).  re/im: [B, C, T, 257] (TF layout from csship_stft
 * with S = B*C).  pairs_l/pairs_r: HOST arrays of n_pairs channel indices.  Writes
 * out[b, f_off + p*257 + f, t] in the reference's [B, *, T] layout.
 * workspace: 2*B*n_pairs*257 floats. */
int csship_ipd(const float* re, const float* im, int64_t B, int64_t C, int64_t T,
               const int32_t* pairs_l, const int32_t* pairs_r, int32_t n_pairs,
               float* out, int64_t out_batch_stride, int64_t f_off, float* workspace,
               csship_stream_t stream);
/* The same features frame-major, the layout the mask estimators consume: out[b, t, f_off + p*257 + f] with ld_out
 * floats between frames (>= f_off + n_pairs*257) - no transposed write, 16-byte row loads.
 * workspace: csship_ipd_tf_workspace(B, n_pairs) bytes. */
size_t csship_ipd_tf_workspace(int64_t B, int32_t n_pairs);
int csship_ipd_tf(const float* re, const float* im, int64_t B, int64_t C, int64_t T,
                  const int32_t* pairs_l, const int32_t* pairs_r, int32_t n_pairs,
                  float* out, int64_t ld_out, int64_t f_off, float* workspace, size_t workspace_bytes,
                  csship_stream_t stream);

/* Inverse of the centre-padded Hann STFT: torch.istft(center=True) as used at
 * css/executor/beamformer.py:83-102.  re/im: [S, 257, T] (the reference's layout);
 * wav: [S, length], length <= 256*(T-1). */
int csship_istft(const float* re, const float* im, int64_t S, int64_t T, int64_t length,
                 float* wav, csship_stream_t stream);

/* ------------------------------------------------------------------ encoder pieces */

/* LayerNorm over the last dim (D == 256 or 512) of x[M, D] float32 -> bf16 and/or f32. */
int csship_layernorm(const float* x, int64_t M, int32_t D, const float* gamma, const float* beta,
                     uint16_t* out_bf16, float* out_f32, csship_stream_t stream);

/* LayerNorm for any D, optionally followed by ReLU (BLSTM embed: Linear + LayerNorm + ReLU,
 * css/models/blstm.py:105-110; BLSTMLayer LayerNorm, blstm.py:146). */
int csship_layernorm_relu(const float* x, int64_t M, int32_t D, const float* gamma, const float* beta,
                          int32_t relu, uint16_t* out_bf16, float* out_f32, csship_stream_t stream);

/* Bidirectional LSTM recurrence (torch.nn.LSTM(bidirectional=True, batch_first=True), zero initial state,
 * gate order i,f,g,o; css/models/blstm.py:139-145).  gx: f32 [B, T, 2, 4H] = x W_ih^T + b_ih + b_hh for
 * (forward, reverse); whh_packed: both directions' W_hh [4H, H] in MFMA fragment order (csship_ffn_fused
 * layout, K zero padded to a multiple of 32), direction-major; out: f32 [B, T, 2H] (forward | reverse).
 * H % 16 == 0, and H % 128 == 0 when H > 128. */
int csship_lstm_bidir(const float* gx, const uint16_t* whh_packed, float* out, int32_t B, int32_t T,
                      int32_t H, csship_stream_t stream);

/* The same recurrence spread over H/32 CUs per (16 batch rows, direction): every wave keeps the W_hh rows of its 16
 * hidden units in registers for the whole sequence and the workgroups exchange h_t once per step through a
 * global panel of epoch-tagged values (bounded spins).  A group whose workgroups all run on one XCD (checked per
 * launch from the hardware's XCC_ID) hands over through that XCD's L2, any other placement through memory with
 * write-through stores; the environment variable CSSHIP_LSTM_L2=0 forces the latter.  All workgroups of a group must
 * be resident together: 2 * ceil(B/16) * H/32 workgroups of 128 threads here, 2 * ceil(B/16) * H/16 workgroups of 64
 * threads (64 KB of LDS at H = 512) in csship_lstm_bidir_bwd_dist.  H = 256 or 512; identical results.
 * gates_out / c_out: both NULL, or both given (training: activated gates f32 [B, T, 2, 4H], cell states f32
 * [B, T, 2, H]).  workspace: csship_lstm_dist_workspace(B, H) bytes (0 = unsupported shape), 16-byte aligned; its
 * head is zeroed by the call.  After the stream has drained, the 32-bit word at workspace +
 * csship_lstm_dist_workspace_error_offset(B, H) is 0, or 1 if a wait timed out (the workgroups of a group were not
 * resident together): the outputs are then invalid. */
size_t csship_lstm_dist_workspace(int32_t B, int32_t H);
size_t csship_lstm_dist_workspace_error_offset(int32_t B, int32_t H);
int csship_lstm_bidir_dist(const float* gx, const uint16_t* whh_packed, float* out, float* gates_out, float* c_out,
                           int32_t B, int32_t T, int32_t H, void* workspace, size_t workspace_bytes,
                           csship_stream_t stream);

/* C[M,N] = epilogue( A[M,K] @ W[N,K]^T + bias ).  bf16 MFMA, fp32 accumulate.
 * Replaces every torch.nn.Linear on the path (css/models/conformer.py:98,124,205-209,
 * 315-318).  W is bf16 [N, ldw] with ldw a multiple of 64 and >= K, zero padded. */
enum { CSSHIP_A_BF16 = 0, CSSHIP_A_F32 = 1, CSSHIP_A_F32_MVN = 2 };
enum {
  CSSHIP_EPI_RELU = 1,     /* max(0, .) applied last */
  CSSHIP_EPI_RESIDUAL = 2, /* v = residual + alpha * v */
  CSSHIP_EPI_LAYERNORM = 4,/* LayerNorm over the N == 256 output row (after residual) */
  CSSHIP_EPI_OUT_BF16 = 8  /* C is bf16 (else float32) */
};
typedef struct {
  int32_t M, N, K;
  int32_t lda, ldw, ldc, ldr; /* row strides in elements */
  int32_t a_mode;             /* CSSHIP_A_* */
  int32_t epilogue;           /* OR of CSSHIP_EPI_* */
  int32_t rows_per_batch;     /* CSSHIP_A_F32_MVN: rows sharing one mean/rinv row */
  float alpha;                /* residual scale */
} csship_gemm_params;
int csship_gemm(const void* A, const uint16_t* W, const float* bias, const float* residual,
                void* C, const float* mvn_mean, const float* mvn_rinv, const float* ln_gamma,
                const float* ln_beta, const csship_gemm_params* p, csship_stream_t stream);

/* ---- row-panel kernels (css_amd/csrc/panel.hip), d_model == 256.
 * Weights for these entry points are bf16 in MFMA FRAGMENT ORDER ("packed"): a matrix
 * W[R, K] (R output features, K % 32 == 0) is stored as blocks [ceil(R/16)][K/32], each
 * block 64 lanes x 8 bf16 = 1 KB with
 *     block(rt, ks)[lane = 16*g + m][j] = W[16*rt + m][32*ks + 8*g + j]   (zero beyond R),
 * which is exactly one wave's A-operand fragment of v_mfma_f32_16x16x32_bf16
 * (css_amd.ops.pack_frag builds it).
 *
 * csship_ffn_fused: out = x + alpha * FeedForward(x), optionally followed by a LayerNorm
 * (css/models/conformer.py:322-326 with the residuals of :355/:358 and the block-final
 * LayerNorm of :360) as ONE kernel; x/out f32 [M, 256] (out may alias x);
 * w1_packed from [d_ffn, 256], w2_packed from [256, d_ffn]; d_ffn % 256 == 0. */
int csship_ffn_fused(const float* x, float* out, const float* ln_g, const float* ln_b,
                     const uint16_t* w1_packed, const float* b1, const uint16_t* w2_packed,
                     const float* b2, const float* final_ln_g, const float* final_ln_b,
                     int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                     csship_stream_t stream);

/* csship_ffn_wide: the same fused FeedForward for d_model 256 OR 512 (css_amd/csrc/ffnw.hip; Conformer-large of
 * BASELINE.json configs[4]: d_model 512, d_ffn 2048): w1_packed from [d_ffn, d_model], w2_packed from [d_model, d_ffn],
 * both in the fragment order above; x / out f32 [M, d_model] (out may alias x); d_ffn % 256 == 0. */
int csship_ffn_wide(const float* x, float* out, const float* ln_g, const float* ln_b, const uint16_t* w1_packed,
                    const float* b1, const uint16_t* w2_packed, const float* b2, const float* final_ln_g,
                    const float* final_ln_b, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                    csship_stream_t stream);

/* The two Linears of the attention stage at d_model 256 OR 512 as row-panel kernels (css_amd/csrc/ffnw.hip):
 * csship_ln_linear_wide: out_bf16[M, N] = bf16(LayerNorm(x) W^T + b)  (conformer.py:222-225; N % 256 == 0; w / bias rows
 *   in the 8-feature interleave of css_amd.ops.pack_frag_il8 / il8_row_order - out itself is in natural feature order);
 * csship_linear_res_wide: out[M, d_model] = res + a_bf16 W^T + b  (conformer.py:254, 356; w from css_amd.ops.pack_frag;
 *   out may alias res). */
int csship_ln_linear_wide(const float* x, const float* ln_g, const float* ln_b, const uint16_t* w_il8_packed,
                          const float* bias_il8, uint16_t* out_bf16, int32_t M, int32_t d_model, int32_t N,
                          csship_stream_t stream);
int csship_linear_res_wide(const uint16_t* a_bf16, const uint16_t* w_packed, const float* bias, const float* res,
                           float* out, int32_t M, int32_t d_model, csship_stream_t stream);

/* csship_ffn_mx: the fused FeedForward with BLOCK-SCALED FP8 operands (OCP MXFP8: e4m3 elements, one e8m0 power-of-two
 * scale per 32 consecutive elements of the reduction dimension; v_mfma_scale_f32_16x16x128_f8f6f4) in both products:
 * LayerNorm output and ReLU'd hidden activation are quantised in the kernel (scale 2^(floor(log2(max|block|)) - 8),
 * round to nearest even, saturating at +-448), weights arrive quantised and packed (css_amd.ops.pack_mx):
 *   w_mx     e4m3 bytes [R/16 tiles][K/128 k-steps][2 halves][64 lanes][16]: lane 16*g + m holds
 *            W[16*tile + m][128*ks + 64*half + 16*g + (0..15)]   (the operand layout of the instruction, see ffnw.hip)
 *   w_scales uint32 [R/16][K/256][64 lanes]: byte q (q = 0, 1) of lane 16*b + m = e8m0 scale of the 32-block
 *            4*(2*kp + q) + b of row 16*tile + m; bytes 2, 3 = 127.
 * Accumulation, biases, residual and LayerNorms are f32.  d_model 256 or 512, d_ffn % 256 == 0.  OPT-IN numerics: not
 * within the 1e-4 mask budget on arbitrary weights; parity is stated against oracle/mxfp8.py. */
int csship_ffn_mx(const float* x, float* out, const float* ln_g, const float* ln_b, const uint8_t* w1_mx,
                  const uint32_t* w1_scales, const float* b1, const uint8_t* w2_mx, const uint32_t* w2_scales,
                  const float* b2, const float* final_ln_g, const float* final_ln_b, int32_t M, int32_t d_model,
                  int32_t d_ffn, float alpha, csship_stream_t stream);

/* Training form of csship_ffn_fused with the reference's two dropouts (conformer.py:317,319):
 *   out = x + alpha * D_out( D_hid( relu(LN(x) W1^T + b1) ) W2^T + b2 )  [-> LayerNorm];
 * D_hid indexes the hidden activation as [M, d_ffn] row-major, D_out the output as [M, 256].  NULL = no dropout. */
int csship_ffn_fused_drop(const float* x, float* out, const float* ln_g, const float* ln_b,
                          const uint16_t* w1_packed, const float* b1, const uint16_t* w2_packed,
                          const float* b2, const float* final_ln_g, const float* final_ln_b,
                          int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                          const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out,
                          csship_stream_t stream);
/* The same; with the final LayerNorm, pre_ln_out (optional, f32 [M, 256], != out) also receives the rows in front of it:
 * the block's last FeedForward and the block's LayerNorm (conformer.py:360-361) as one launch in the training forward,
 * whose backward needs both. */
int csship_ffn_fused_train(const float* x, float* out, float* pre_ln_out, const float* ln_g, const float* ln_b,
                           const uint16_t* w1_packed, const float* b1, const uint16_t* w2_packed,
                           const float* b2, const float* final_ln_g, const float* final_ln_b,
                           int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                           const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out,
                           csship_stream_t stream);

/* csship_panel_linear: out[M, N] = epi( pro(a)[M, 256] @ W[N, 256]^T + bias ).
 * prologue: LayerNorm of f32 rows (pre-LN of MultiHeadedAttention, conformer.py:222),
 * bf16 rows as they are, or f32 rows rounded to bf16.  epilogue: OR of CSSHIP_EPI_RELU,
 * CSSHIP_EPI_RESIDUAL (f32 residual, N % 4 == 0; out may alias it), CSSHIP_EPI_OUT_BF16. */
enum { CSSHIP_PRO_LN = 0, CSSHIP_PRO_BF16 = 1, CSSHIP_PRO_F32 = 2 };
int csship_panel_linear(const void* a, int32_t prologue, const float* ln_g, const float* ln_b,
                        const uint16_t* w_packed, const float* bias, const float* residual,
                        int64_t ldr, float alpha, void* out, int64_t ldo, int32_t M, int32_t N,
                        int32_t d_model, int32_t epilogue, csship_stream_t stream);
/* The same with dropout of the projection's output in front of the residual add (training: x + D(a W^T + b),
 * conformer.py:254): out = residual + alpha * mask / (1 - p) o (a W^T + b), element index row * N + column as
 * csship_dropout_add.  With a dropout cfg: epilogue == CSSHIP_EPI_RESIDUAL, N % 4 == 0, ldo % 4 == 0.  NULL = none. */
int csship_panel_linear_drop(const void* a, int32_t prologue, const float* ln_g, const float* ln_b,
                             const uint16_t* w_packed, const float* bias, const float* residual,
                             int64_t ldr, float alpha, void* out, int64_t ldo, int32_t M, int32_t N,
                             int32_t d_model, int32_t epilogue, const csship_dropout_cfg* drop, csship_stream_t stream);

/* csship_outproj_glu: attention output projection + residual, and the front half of the conv module on the
 * fresh row:  x_out = x + attn_out @ Wo^T + bo  (conformer.py:254,356);  g_out = glu(LayerNorm_conv(x_out))
 * (conformer.py:291-293).  attn_out bf16 [M,256]; x, x_out, g_out f32 [M,256] (x_out may alias x). */
struct csship_conv_weights_s;
int csship_outproj_glu(const uint16_t* attn_out, const uint16_t* wo_packed, const float* bo, const float* x,
                       float* x_out, const struct csship_conv_weights_s* conv_w, float* g_out, int32_t M,
                       int32_t d_model, csship_stream_t stream);

/* csship_encoder_chain: the row-local operators between two attention stages as ONE kernel, per 64-row panel:
 *   [conv_g != NULL] x = x + pw2(relu(BN_eval(dwconv_33(conv_g))))   back half of the conv module of block l-1
 *                    (conformer.py:294-301,357; conv_g from csship_outproj_glu; T = frames per utterance,
 *                    taps never cross an utterance.  Only conv_g is read with halo rows, x is row-local:
 *                    x_out may alias x but not conv_g)
 *   [ffn_a != NULL]  x = LayerNorm_final( x + 0.5 FFN_a(x) )      FFN_out of block l-1 (conformer.py:358-360)
 *   [ffn_b != NULL]  x = x + 0.5 FFN_b(x)                         FFN_in of block l   (conformer.py:355)
 *   x_out <- x
 *   [wqkv != NULL]   qkv_out = bf16( LayerNorm_qkv(x) @ Wqkv^T + bqkv )   (conformer.py:222-225)
 * Supported combinations: (a,b,qkv), (b,qkv), (a), (conv,a,b,qkv), (conv,a).
 * A LayerNorm whose gamma AND beta pointers are NULL (ffn->ln_g / ln_b, qkv_ln_g / qkv_ln_b; never the block-final
 * one) is applied WITHOUT its affine part: the caller folded it into the Linear behind it, W' = W diag(gamma),
 * b' = b + W beta (css_amd.ops.fold_layernorm) - the kernel then reads no LayerNorm parameters in its prologues.
 * flags: CSSHIP_CHAIN_QKV_IL8 = wqkv_packed / bqkv rows are in the 8-feature interleaved order of
 * css_amd.ops.pack_frag_il8 (packed row 16*tile + m holds feature 32*(tile/2) + 8*(m/4) + 4*(tile%2) + m%4), with which
 * a lane's results are 8 consecutive features of a token row and qkv_out is written with 16-byte stores
 * (n_qkv % 32 == 0); qkv_out itself is in natural feature order either way. */
#define CSSHIP_CHAIN_QKV_IL8 1
/* (csship_conformer_weights.chain_flags only, ignored by csship_encoder_chain) the q rows of every layer's wqkv / bqkv
 * carry log2(e) / sqrt(d_k): csship_conformer_forward runs the attention with CSSHIP_MHA_Q_LOG2E */
#define CSSHIP_FWD_Q_LOG2E 4
struct csship_ffn_weights_s;
int csship_encoder_chain(const float* x, float* x_out, const float* conv_g,
                         const struct csship_conv_weights_s* conv_w, int32_t T,
                         const struct csship_ffn_weights_s* ffn_a, const float* final_ln_g,
                         const float* final_ln_b, const struct csship_ffn_weights_s* ffn_b,
                         const float* qkv_ln_g, const float* qkv_ln_b, const uint16_t* wqkv_packed,
                         const float* bqkv, uint16_t* qkv_out, int32_t M, int32_t d_model, int32_t d_ffn,
                         int32_t n_qkv, int32_t flags, csship_stream_t stream);

/* The training forward through the same kernel: stage A = the previous block's last FeedForward + residual + that block's
 * final LayerNorm, stage B = this block's first FeedForward + residual, stage C = LayerNorm + q | k | v, with the
 * reference's dropouts (element indices as csship_ffn_fused_drop) and the rows the backward pass keeps.  ffn_a may be
 * NULL (first block: stages B + C only); ffn_b and wqkv_packed are required; the LayerNorm parameters must be given
 * (no folded weights), wqkv_packed / bqkv in plain row order.  x_out = the rows behind stage B. */
typedef struct {
  const csship_dropout_cfg *drop_a_hidden, *drop_a_out, *drop_b_hidden, *drop_b_out;  /* NULL = no dropout */
  float* pre_ln_out;  /* optional, with ffn_a: rows in front of the final LayerNorm (f32 [M, 256]) */
  float* ln_out;      /* optional, with ffn_a: rows behind it = stage B's input */
  /* Optional (bn_z != NULL; needs ffn_a and whole 64-row panels): the kernel forms its own input rows - the tail of the conv
   * module behind csship_outproj_conv_train(..., y = NULL): x3 = bn_x2 + dropout(pw2_w * relu(bn_z * scale + shift) + pw2_b)
   * (conformer.py:298-303; bn_z = the workspace's pre-BatchNorm activations, bn_scale_shift = its scale | shift block,
   * f32 [2 * 256]; drop_bn = the module's final dropout, element index row * 256 + channel) - and writes them to bn_out
   * (f32 [M, 256]).  `x` is ignored then.  Replaces the BatchNorm-apply launch. */
  const float *bn_z, *bn_x2, *bn_scale_shift, *bn_pw2_w, *bn_pw2_b;
  const csship_dropout_cfg* drop_bn;
  float* bn_out;
} csship_chain_train;
int csship_encoder_chain_train(const float* x, float* x_out, const struct csship_ffn_weights_s* ffn_a,
                               const float* final_ln_g, const float* final_ln_b, const struct csship_ffn_weights_s* ffn_b,
                               const float* qkv_ln_g, const float* qkv_ln_b, const uint16_t* wqkv_packed,
                               const float* bqkv, uint16_t* qkv_out, int32_t M, int32_t d_model, int32_t d_ffn,
                               int32_t n_qkv, const csship_chain_train* train, csship_stream_t stream);

/* csship_embed: x0 = ReLU(LayerNorm(Linear(MVN_time(f))))  (css/models/conformer.py:95, 124-129) as one
 * row-panel kernel; f f32 [B, T, idim] (row stride ldf, any idim); mvn_mean / mvn_rinv from
 * csship_mvn_stats; w_packed = fragment-packed [256, Kp] with Kp = idim rounded up to 256 (zero padded);
 * out f32 [B*T, 256]. */
int csship_embed(const float* f, int64_t ldf, const float* mvn_mean, const float* mvn_rinv,
                 const uint16_t* w_packed, const float* bias, const float* ln_g, const float* ln_b,
                 float* out, int32_t B, int32_t T, int32_t idim, int32_t d_model, csship_stream_t stream);
/* Training form: the same kernel also leaves what the backward keeps - fn_bf16 = MVN_time(f) as bf16 [B*T, ld_fn] (ld_fn a
 * multiple of 8 with idim <= ld_fn <= Kp, zero beyond idim: the operand of the embedding's weight gradient) and e_out = the
 * rows in front of the LayerNorm (f32 [B*T, 256]) - and applies the embedding's dropout (conformer.py:127, element index
 * row * 256 + column; NULL = none).  Replaces csship_mvn_apply + csship_bgemm + csship_layernorm_relu + csship_dropout_apply. */
int csship_embed_train(const float* f, int64_t ldf, const float* mvn_mean, const float* mvn_rinv,
                       const uint16_t* w_packed, const float* bias, const float* ln_g, const float* ln_b, float* out,
                       uint16_t* fn_bf16, int32_t ld_fn, float* e_out, int32_t B, int32_t T, int32_t idim, int32_t d_model,
                       const csship_dropout_cfg* drop, csship_stream_t stream);

/* Relative-position multi-head attention core, MultiHeadedAttention.forward
 * (css/models/conformer.py:229-253) with RelativePositionalEncoding
 * (conformer.py:179-185) folded in:
 *   score[i,j] = q_i . k_j + q_i . pe_k[clamp(i-j, -maxlen, maxlen-1) + maxlen]
 * qkv: bf16 [B, T, 3*H*64] (q | k | v, q ALREADY scaled by 1/sqrt(d_k)); out: bf16
 * [B, T, H*64]; d_k must be 64.  The T x T x d_k table of conformer.py:159 is never
 * materialised.  pe_packed holds pe_k as reversed, aligned 16-row blocks in MFMA fragment
 * order with the clamp baked in: for q in [q_min, q_max], ks in {0,1}
 *   block(q, ks)[lane = 16*g + m][j] = pe_k[clamp(16*q + 15 - m, -maxlen, maxlen-1) + maxlen][32*ks + 8*g + j]
 * at pe_packed + ((q - q_min)*2 + ks) * 512 bf16.  Block q_min must be entirely below the
 * clamp and block q_max entirely above it (css_amd.ops.pack_relpos builds it). */
int csship_mha_relpos(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                      uint16_t* out, int32_t B, int32_t T, int32_t H, csship_stream_t stream);
/* The same with flags.  CSSHIP_MHA_Q_LOG2E: q carries log2(e) / sqrt(d_k) instead of 1 / sqrt(d_k) (the caller folded
 * the factor into the q rows of the projection, as css_amd.models.Conformer does for the inference forward): scores come
 * out of the matrix products in log2 units and relative to the running maximum, the exponentials take them as they are
 * (same softmax; the inference form only - no dropout, no log-sum-exp output). */
#define CSSHIP_MHA_Q_LOG2E 1
int csship_mha_relpos_ex(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                         uint16_t* out, int32_t B, int32_t T, int32_t H, int32_t flags, csship_stream_t stream);

/* ConvModule.forward + residual (css/models/conformer.py:289-303, 357):
 *   y = x + pw2( relu( BN( dwconv_k( glu( LN(x) ) ) ) ) ),  x,y: f32 [B, T, 256].
 * pw1_w/pw1_b: 2 floats each (Conv2d(1,2,1)), pw2_w/pw2_b: 1 float each (device).
 * bn_mode 0: running statistics (eval).  bn_mode 1: batch statistics (train forward); bn_mode 2: running statistics
 * but through the same three passes as 1, keeping the pre-BatchNorm activations for csship_convmod_bwd (gradients with
 * a frozen BatchNorm).  Modes 1 and 2 need a workspace of csship_convmod_workspace() bytes; nothing is updated. */
typedef struct csship_conv_weights_s {
  const float *ln_g, *ln_b;
  const float *pw1_w, *pw1_b;
  const float *dw_w, *dw_b; /* depthwise taps TRANSPOSED to [k, D] (coalesced per-channel reads), bias [D] */
  const float *bn_g, *bn_b, *bn_mean, *bn_var;
  const float *pw2_w, *pw2_b;
  int32_t causal; /* 0: taps centred on the frame (padding (k-1)/2, conformer.py:274-282); 1: the causal module of
                   * conformer.py:265-273, 297-298 (tap j of frame t reads frame t + j - (k-1)).  Honoured by csship_convmod*
                   * and their backward twins; the fused forms (csship_outproj_conv, csship_outproj_glu + csship_conv_from_glu,
                   * csship_encoder_chain with conv_g) return CSSHIP_EUNSUPPORTED for it and csship_conformer_forward
                   * takes the separate output projection + csship_convmod instead */
} csship_conv_weights;
size_t csship_convmod_workspace(int32_t B, int32_t T, int32_t D);
int csship_convmod(const float* x, float* y, const csship_conv_weights* w, int32_t B, int32_t T,
                   int32_t D, int32_t kernel_size, int32_t bn_mode, void* workspace,
                   csship_stream_t stream);

/* Training form with the module's final dropout (conformer.py:302): y = x + D(pw2(relu(BN(...)))), bn_mode 1 or 2;
 * D indexes the branch as [B*T, D] row-major. */
int csship_convmod_drop(const float* x, float* y, const csship_conv_weights* w, int32_t B, int32_t T,
                        int32_t D, int32_t kernel_size, int32_t bn_mode, void* workspace,
                        const csship_dropout_cfg* drop, csship_stream_t stream);

/* SyncBatchNorm (train.py:70 convert_sync_batchnorm) as two calls with the caller's all-reduce in between:
 *   csship_convmod_stats  : LayerNorm .. depthwise conv, pre-BN activations kept; this rank's per-channel sums land as
 *                           f64 [2D] = (sum z | sum z^2) at byte csship_convmod_sums_offset() of the workspace;
 *   (caller: all-reduce SUM of those 2D doubles over the ranks; total_rows = sum of B*T over the ranks)
 *   csship_convmod_finish : statistics from the reduced sums, then BN + ReLU + pw2 + dropout + residual as above.
 * After finish the workspace's last 4*D floats are [scale | shift | mean | biased var] of the GLOBAL batch. */
size_t csship_convmod_sums_offset(int32_t B, int32_t T, int32_t D);
int csship_convmod_stats(const float* x, const csship_conv_weights* w, int32_t B, int32_t T, int32_t D,
                         int32_t kernel_size, void* workspace, csship_stream_t stream);
int csship_convmod_finish(const float* x, float* y, const csship_conv_weights* w, int32_t B, int32_t T, int32_t D,
                          double total_rows, void* workspace, const csship_dropout_cfg* drop, csship_stream_t stream);

/* Back half of the conv module from a precomputed g = glu(LayerNorm_conv(x)) (csship_outproj_glu):
 *   y = x + pw2( relu( BN_eval( dwconv_k(g) ) ) ).  x, g, y: f32 [B, T, D]; y may alias x (x is read
 * row-locally) but not g (neighbouring tiles read halo rows of g). */
int csship_conv_from_glu(const float* x, const float* g, float* y, const csship_conv_weights* w, int32_t B,
                         int32_t T, int32_t D, int32_t kernel_size, csship_stream_t stream);

/* ------------------------------------------------------------------ PIT-MSE
 * Fused mask application + permutation-invariant MSE: pit_loss (css/objectives/pit.py:5-62)
 * looped over the batch as MeanSquaredError.forward does (css/objectives/mse.py:31-37),
 * with y_pred = mask_k * f_orig (css/models/conformer.py:103) formed on the fly.
 *   masks: f32 [B, T, ld_mask] with mask k at columns [k*F, (k+1)*F)
 *   f_orig: f32 [B, T, ld_f] (first F columns are the magnitudes)
 *   targets: HOST array of K device pointers, each f32 [B, T, F]
 *   y_pred (optional): f32 [B, K, T, F]
 *   loss: 1 float (mean over batch of the per-sample minimum); perm: int32 [B, K] with
 *   perm[b][i] = index of the estimate matched to target i (pit.py:57 convention);
 *   workspace: csship_pit_workspace(B, T, F, K) bytes.  K <= 4. */
size_t csship_pit_workspace(int32_t B, int32_t T, int32_t F, int32_t K);
int csship_pit_mse(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f,
                   const float* const* targets_host, int32_t B, int32_t T, int32_t F, int32_t K,
                   float* y_pred, float* loss, int32_t* perm, float* pair_sse, void* workspace,
                   csship_stream_t stream);

/* y_pred[b,k,t,f] = masks[b,t,k*F+f] * f_orig[b,t,f], k < K  (css/models/conformer.py:103,
 * css/models/blstm.py:87): the stacked masked spectra the model's forward returns. */
int csship_apply_masks(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f,
                       int32_t B, int32_t T, int32_t F, int32_t K, float* y_pred,
                       csship_stream_t stream);

/* d loss / d masks for the chosen permutation (hand-written backward of the op above):
 * grad_masks[b,t,k*F+f] = 2*(m_k f - tgt_{inv(k)}) * f / (B*K*T*F) * grad_loss; columns of
 * masks beyond K*F (noise mask) get 0. */
int csship_pit_mse_bwd(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f,
                       const float* const* targets_host, const int32_t* perm, int32_t B,
                       int32_t T, int32_t F, int32_t K, int32_t n_masks, float grad_loss,
                       float* grad_masks, csship_stream_t stream);

/* The same with the upstream gradient of the loss ALSO read from device memory (grad_loss_dev: one f32, may be NULL):
 * grad_masks = (the above with grad_loss) * *grad_loss_dev, the two factors applied in that order.  An autograd
 * backward receives d(objective)/d(loss) as a device scalar; multiplying by it afterwards was a second pass over the
 * B*T*n_masks*F gradient. */
int csship_pit_mse_bwd_dev(const float* masks, int64_t ld_mask, const float* f_orig, int64_t ld_f,
                           const float* const* targets_host, const int32_t* perm, int32_t B,
                           int32_t T, int32_t F, int32_t K, int32_t n_masks, float grad_loss,
                           const float* grad_loss_dev, float* grad_masks, csship_stream_t stream);

/* ------------------------------------------------------------------ stitching
 * Stitcher.get_stitch / get_connect (css/executor/stitcher.py:14-110).
 *   mags: f32 [W, F, T]; masks: f32 [W, F, T, 3] (spk1, spk2, noise);
 *   sim:  f32 [W-1, 4] = S[i][j] at index 2*i + j (keep if S00 + S11 >= S01 + S10);
 *   state: int32 [W] running swap parity per window; out: f32 [3, F, hop*(W-1) + T]. */
int csship_stitch_similarity(const float* mags, const float* masks, int32_t W, int32_t F, int32_t T,
                             int32_t margin, float* sim, csship_stream_t stream);
int csship_stitch_connect(const float* masks, const int32_t* state, int32_t W, int32_t F, int32_t T,
                          int32_t hop, float wta_floor, float* out, csship_stream_t stream);

/* ------------------------------------------------------------------ whole forward
 * Conformer.forward (css/models/conformer.py:83-104) from the feature tensor to the
 * ReLU masks, as ONE host call that enqueues every kernel (no per-op Python overhead).
 * All weight pointers are device pointers; the structs themselves live in host memory. */
typedef struct csship_ffn_weights_s {
  const float *ln_g, *ln_b;
  const uint16_t* w1; const float* b1; /* PACKED bf16 from [d_ffn, d_model]; f32 [d_ffn] */
  const uint16_t* w2; const float* b2; /* PACKED bf16 from [d_model, d_ffn]; f32 [d_model] */
  const uint32_t *w1_scale, *w2_scale; /* csship_conformer_weights.ffn_mode == 2 only: w1 / w2 are the e4m3 bytes of
                                        * css_amd.ops.pack_mx and these its e8m0 scale dwords (csship_ffn_mx); else NULL */
} csship_ffn_weights;
typedef struct {
  const float *ln_g, *ln_b;
  const uint16_t* wqkv; const float* bqkv; /* PACKED bf16 from [3*d_model, d_model], q rows pre-scaled by 1/sqrt(d_k) */
  const uint16_t* wo; const float* bo;     /* PACKED bf16 from [d_model, d_model] */
} csship_mha_weights;
typedef struct {
  csship_ffn_weights ffn_in;
  csship_mha_weights mha;
  csship_conv_weights conv;
  csship_ffn_weights ffn_out;
  const float *ln_g, *ln_b;
} csship_layer_weights;
typedef struct {
  int32_t idim, idim_pad;  /* input feature dim and its zero-padded multiple of 64 */
  int32_t d_model, n_head, d_ffn, n_layer, kernel_size;
  int32_t n_out;           /* num_bins * (num_spk + num_noise) */
  int32_t pe_q_min, pe_q_max; /* block range of pe_k (csship_mha_relpos) */
  int32_t bn_mode;         /* 0 eval, 1 batch statistics */
  int32_t generic;         /* 0: tuned d_model = 256 path, PACKED weights as documented per field; 1: any-width
                            * unfused path (d_model 256 or 512), ALL Linear weights ROW-MAJOR bf16 [N, K padded to 64] */
  int32_t fuse_conv;       /* 0 = default (3 in eval mode, 4 with batch statistics); eval only: 1 = conv fused into
                            * csship_outproj_glu + csship_encoder_chain; 2 = csship_outproj_glu + csship_conv_from_glu;
                            * 3 = csship_outproj_conv (output projection + conv module in one kernel);
                            * 4 = separate csship_panel_linear + csship_convmod */
  const uint16_t* embed_w; const float* embed_b; /* PACKED bf16 from [d_model, idim_pad], idim_pad = idim rounded up to 256 */
  const float *embed_ln_g, *embed_ln_b;
  const uint16_t* pe_k;    /* PACKED relative-position blocks (csship_mha_relpos) */
  const csship_layer_weights* layers; /* HOST array [n_layer] */
  const uint16_t* head_w; const float* head_b; /* PACKED bf16 from [n_out, d_model] */
  int32_t ffn_mode;        /* generic == 1 only: 0 = FeedForward unfused (LayerNorm + two tiled Linears, row-major weights);
                            * 1 = csship_ffn_wide (FeedForward weights in FRAGMENT order, d_model 256 / 512, d_ffn % 256 == 0);
                            * 2 = csship_ffn_mx (opt-in block-scaled fp8: w1 / w2 / w*_scale from css_amd.ops.pack_mx) */
  int32_t lin_mode;        /* generic == 1 only: 0 = q/k/v and output projections as LayerNorm + tiled Linear (row-major
                            * weights); 1 = csship_ln_linear_wide / csship_linear_res_wide (mha.wqkv / bqkv in the IL8
                            * fragment order, mha.wo in fragment order; d_model 256 / 512) */
  int32_t chain_flags;     /* generic == 0: flags handed to csship_encoder_chain (CSSHIP_CHAIN_QKV_IL8: every layer's
                            * mha.wqkv / bqkv in the interleaved row order).  FeedForward / attention LayerNorms with NULL
                            * gamma and beta are folded into w1 / wqkv (see csship_encoder_chain) */
} csship_conformer_weights;
size_t csship_conformer_workspace(const csship_conformer_weights* w, int32_t B, int32_t T);
/* f: f32 [B, T, idim] (row stride ldf); masks: f32 [B, T, n_out] with row stride ld_masks >= n_out
 * (a multiple of 4 enables vector stores); enc_out (optional): f32 [B, T, d_model] encoder output
 * before the head.  masks == NULL (with enc_out): the mask head is skipped - csship_head_pit continues from enc_out. */
int csship_conformer_forward(const csship_conformer_weights* w, const float* f, int64_t ldf,
                             int32_t B, int32_t T, float* masks, int64_t ld_masks, float* enc_out,
                             void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* csship_head_pit (css_amd/csrc/headpit.hip): mask head + mask application + PIT-MSE as one kernel for the loss-only
 * path - ReLU(Linear(x_enc)) (conformer.py:98-101), y_k = m_k * f_orig (:103), the K x K pairwise squared errors and the
 * minimum over permutations / batch mean (pit.py:53-57, mse.py:31-37), without ever writing the mask tensors (the
 * noise mask, which does not enter the loss, is not computed).  x_enc: f32 [B*T, d_model] (csship_conformer_forward
 * with masks == NULL writes it to enc_out); w_perm_packed / bias_perm: the first K*F head rows in the order of
 * css_amd.ops.pack_head_pit (bins 0..255 of speaker 0, of speaker 1, then one tile with bin 256 of every speaker),
 * fragment-packed; f_orig f32 [B, T, >= F] (row stride ld_f); targets: K x f32 [B, T, >= F] (row stride ld_t: 257 dense,
 * or 260 - csship_stft_ld's 16-byte aligned rows, with which a lane's 16-byte pieces never straddle a line).  loss: 0-dim f32; perm
 * int32 [B, K] (optional); pair_sse f32 [B, K, K] (optional).  Built for K == 2, F == 257, d_model == 256, T >= 64:
 * anything else returns CSSHIP_EUNSUPPORTED (callers then run csship_conformer_forward + csship_pit_mse).
 * Bit-reproducible (fixed-order partial sums, no atomics). */
size_t csship_head_pit_workspace(int32_t B, int32_t T);
int csship_head_pit(const float* x_enc, const uint16_t* w_perm_packed, const float* bias_perm, const float* f_orig,
                    int64_t ld_f, const float* const* targets_host, int64_t ld_t, int32_t B, int32_t T, int32_t F,
                    int32_t K, int32_t d_model, float* loss, int32_t* perm, float* pair_sse, void* workspace,
                    size_t workspace_bytes, csship_stream_t stream);

/* ======================================================================================
 * Backward pass ("_bwd twins").  The reference gets these from torch.autograd over the ops of
 * css/models/conformer.py; here every gradient is an explicit kernel.
 * ====================================================================================== */

/* ---- csship_bgemm (css_amd/csrc/bgemm.hip): batched C[z] = epi(alpha * opA(A[z]) . opB(B[z])), bf16 MFMA,
 * f32 accumulate.  C is [M, N]; the reduction has length K.  An operand is either reduction-contiguous
 * (A: [M, K] rows, B: [N, K] rows) or reduction-strided / "kmajor" (A: [K, M], B: [K, N]) - so dX = dY.W uses the
 * forward's [N, K] weight as a kmajor B, and dW = dY^T.X takes both activations kmajor; nothing is transposed in
 * memory.  f32 operands are rounded to bf16 on load.  Row strides: multiples of 8 (bf16) / 4 (f32) elements.
 * Batch index z = z1 * batch2 + z2 with element strides s?1 / s?2 (sX* for aux).  splits: 0 = choose
 * (long reductions with few output tiles are split into f32 slabs in the workspace and summed in a fixed order),
 * splits > 1 needs batch1 * batch2 == 1 and an f32 output without CSSHIP_BEPI_MASK_POS. */
enum {
  CSSHIP_BEPI_OUT_BF16 = 1, /* C is bf16 (else f32) */
  CSSHIP_BEPI_MASK_POS = 2, /* v = aux_bf16[row, col] > 0 ? v : 0   (ReLU backward) */
  CSSHIP_BEPI_ADD = 4,      /* v += aux_f32[row, col]               (gradient accumulation / residual; aux may alias C) */
  CSSHIP_BEPI_SCALE_BIAS = 8, /* v = alpha * (acc + bias) instead of alpha * acc + bias (forward Linear with a scaled residual branch) */
  CSSHIP_BEPI_RELU = 16     /* v = max(v, 0), applied last */
};
typedef struct {
  int32_t M, N, K;
  int64_t lda, ldb, ldc, ldaux;
  int32_t a_kmajor, b_kmajor, a_f32, b_f32;
  int32_t epilogue;         /* OR of CSSHIP_BEPI_* */
  int32_t batch1, batch2, splits;
  int64_t sA1, sA2, sB1, sB2, sC1, sC2, sX1, sX2; /* sX*: aux */
  float alpha;
  int32_t band;             /* > 0: A is banded and output tiles skip the 64-deep reduction tiles outside their rows'
                             * band - same results (the skipped products are exact zeros).  Row-major A: A[m, k] = 0
                             * unless m <= k <= m + band (the skewed dS of the relative-position attention as the left
                             * operand: band = T - 1 of W = 2T - 1 columns, a third less to read).  kmajor A with
                             * band_stride > 0: the reduction index is k = z * period + t * band_stride + h with
                             * period = K / splits (splits given, K % (splits * 64) == 0), and A[k, m] = 0 unless
                             * t <= m <= t + band (the same matrix as the left operand of d(pe_window) = dS_skew^T q,
                             * rows (b, t, h): band_stride = H, splits = B).  0 = dense. */
  int32_t band_stride;
} csship_bgemm_params;
size_t csship_bgemm_workspace(const csship_bgemm_params* p);
int csship_bgemm(const void* A, const void* B, void* C, const float* bias, const void* aux,
                 const csship_bgemm_params* p, void* workspace, size_t workspace_bytes, csship_stream_t stream);
/* Weight AND bias gradient of a Linear y = x W^T + b in one pass over dY (torch.autograd's two products in the
 * reference): dW[M, N] = alpha * dY^T . X (+ aux with CSSHIP_BEPI_ADD) and db[M] = alpha * column sums of dY, where
 * dY is [K, M] and X is [K, N] (both kmajor: p->a_kmajor = p->b_kmajor = 1, no batch, f32 output, M and N multiples of
 * 8; anything else: CSSHIP_EUNSUPPORTED).  The column sums are one more MFMA per dY fragment against ones in the
 * workgroups of the first tile column; with a split reduction they are reduced by the same launch as the slabs
 * (fixed order).  db sums the bf16-ROUNDED dY the product uses.  head_rows > 0: rows [0, head_rows) of dW and db take
 * alpha_head instead of p->alpha (the fused q | k | v projection: q rows carry 1/sqrt(d_k)).
 * Workspace: csship_bgemm_wgrad_workspace(). */
size_t csship_bgemm_wgrad_workspace(const csship_bgemm_params* p);
int csship_bgemm_wgrad(const void* dY, const void* X, float* dW, float* db, const void* aux,
                       const csship_bgemm_params* p, float alpha_head, int32_t head_rows, void* workspace,
                       size_t workspace_bytes, csship_stream_t stream);

/* Deferred reductions of parameter gradients.  The backward kernels leave per-workgroup partial sums ("slabs") in their
 * workspaces and sum them in a second, tiny launch at the ~5 us floor of a launch.  Between csship_reduce_defer(1) and
 * csship_reduce_flush() the calling host thread's SMALL reductions whose outputs are parameter gradients (the dgamma /
 * dbeta of csship_layernorm_bwd and csship_ffn_bwd_fused; at most 4 MB of slabs) are queued instead and run as ONE
 * launch by the flush, with the summation order their own launches would have used (identical results).  Contract while
 * deferring: every such call gets its OWN workspace, left untouched until the flush, and nobody reads those gradients
 * before it.  The large reductions (weight-gradient products) always run at once: right behind their producer the slabs
 * are still in the last-level cache.  csship_reduce_defer returns the previous mode; switching the mode off drops
 * anything still queued (error paths).  csship_reduce_pending: queued jobs. */
int csship_reduce_defer(int32_t on);
/* csship_reduce_hold(1) .. csship_reduce_hold(0), csship_reduce_flush(): EVERY slab reduction of the calling host thread in
 * between is queued, whatever its size - for a PAIR of weight-gradient products (a FeedForward's dW1 and dW2: 2 x 16 MB
 * of slabs, still in the last-level cache when the second product is done) summed by one launch instead of two.  Every
 * product in between needs a workspace of its own, untouched until the flush.  Returns the previous mode. */
int csship_reduce_hold(int32_t on);
int csship_reduce_pending(void);
int csship_reduce_flush(csship_stream_t stream);
/* Grouped weight gradients.  Between csship_wgrad_hold(1) and csship_wgrad_flush() the calling host thread's
 * csship_bgemm_wgrad calls that fit the LDS-DMA loop (bf16 operands, M and N multiples of 128, any K >= 1024 - the rows
 * of a last 64-deep tile beyond K lie outside the buffer descriptor and read as zeros - the same K for all of them, at
 * most 96 tiles, no bias) are queued and run by the flush as ONE launch over all their
 * tiles, the reduction cut only as often as fills the chip once (an encoder block's six products: 80 tiles x 3 pieces
 * instead of 16-64 pieces each), followed by one reduction launch (csship_reduce_flush, which also runs whatever else
 * is queued).  Contract: operands and outputs of a queued call stay valid and unread until the flush; its per-call
 * workspace is not used.  Other calls run at once.  workspace: csship_wgrad_group_workspace() bytes (for what is
 * queued at that moment).  csship_wgrad_hold returns the previous mode; switching off drops what is still queued. */
int csship_wgrad_hold(int32_t on);
int csship_wgrad_pending(void);
size_t csship_wgrad_group_workspace(void);
int csship_wgrad_flush(void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* out[c] = alpha * sum_r x[r, c] for x f32 or bf16 [M, N] (row stride ld): the bias gradient of a Linear. */
size_t csship_colsum_workspace(int64_t M, int32_t N);
int csship_colsum(const void* x, int32_t x_is_f32, int64_t M, int32_t N, int64_t ld, float alpha, float* out,
                  void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* ---- csship_outproj_conv (css_amd/csrc/panel.hip): attention output projection + residual (conformer.py:254,356)
 * AND the convolution module + residual (conformer.py:289-303,357, BatchNorm with running statistics) as one kernel:
 * out = x2 + ConvModule(x2), x2 = x + attn_out Wo^T + bo.  attn_out bf16 [B*T, 256]; x, out f32 [B*T, 256], out != x;
 * conv as for csship_convmod (bn_mean / bn_var required). */
int csship_outproj_conv(const uint16_t* attn_out, const uint16_t* wo_packed, const float* bo, const float* x,
                        float* out, const csship_conv_weights* conv, int32_t B, int32_t T, int32_t d_model,
                        int32_t kernel_size, csship_stream_t stream);

/* Training form (batch-statistics BatchNorm, the reference's dropouts; replaces csship_panel_linear_drop followed by
 * csship_convmod_drop(bn_mode 1), conformer.py:254,356 + 289-303,357 under model.train()):
 *   x2_out = x + D_proj(attn_out Wo^T + bo),   y = x2_out + D_conv(pw2(relu(BN_batch(dwconv(glu(LN(x2_out)))))))
 * One kernel forms x2, LayerNorm, GLU and the depthwise convolution (pre-BatchNorm activations + per-workgroup partial
 * statistics), then the BatchNorm reduce and apply kernels of csship_convmod finish.  workspace: csship_convmod_workspace
 * bytes, left exactly as csship_convmod(bn_mode 1) leaves it (csship_convmod_bwd* take it).  d_model 256, kernel 33,
 * centred taps; drop_proj indexes [B*T, 256] row-major like csship_panel_linear_drop, drop_conv like csship_convmod_drop. */
/* (y may be NULL: the BatchNorm statistics are reduced into the workspace, the apply pass is left to the consumer -
 * csship_encoder_chain_train's bn_* fields) */
int csship_outproj_conv_train(const uint16_t* attn_out, const uint16_t* wo_packed, const float* bo, const float* x,
                              float* x2_out, float* y, const csship_conv_weights* w, int32_t B, int32_t T, int32_t d_model,
                              int32_t kernel_size, const csship_dropout_cfg* drop_proj, const csship_dropout_cfg* drop_conv,
                              void* workspace, csship_stream_t stream);

/* ---- csship_ffn_bwd (css_amd/csrc/panel.hip): the activation side of the FeedForward backward as ONE row-panel kernel.
 * For out = x + alpha * (relu(LN(x) W1^T + b1) W2^T + b2) and dy = d(out) it produces
 *   a_bf16 = LN(x) [M, 256], h_bf16 = relu(.) [M, d_ffn], dh_bf16 = alpha * (dy W2) o [h > 0] [M, d_ffn]
 *   (the operands of the weight-gradient GEMMs dW1 = dh^T a, db1 = colsum(dh), dW2 = alpha dy^T h, db2 = alpha colsum(dy)),
 *   dx = dy + LayerNorm-backward(dh W1) [M, 256] and dgamma_dbeta f32 [2, 256].
 * w1_packed as in csship_ffn_fused; w2t_packed / w1t_packed = fragment-packed W2^T [d_ffn, 256] / W1^T [256, d_ffn]. */
size_t csship_ffn_bwd_workspace(int32_t M);
int csship_ffn_bwd(const float* x, const float* dy, const float* ln_g, const float* ln_b,
                   const uint16_t* w1_packed, const float* b1, const uint16_t* w2t_packed,
                   const uint16_t* w1t_packed, uint16_t* a_bf16, uint16_t* h_bf16, uint16_t* dh_bf16,
                   float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                   void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* Backward of csship_ffn_fused_drop (same cfgs): h_bf16 is the DROPPED hidden activation, dh_bf16 carries both masks,
 * and dyd_bf16 (optional, bf16 [M, 256]) = alpha * dy o output mask / (1 - p): with dropout the weight-gradient GEMMs
 * are dW2 = dyd^T h, db2 = colsum(dyd). */
int csship_ffn_bwd_drop(const float* x, const float* dy, const float* ln_g, const float* ln_b,
                        const uint16_t* w1_packed, const float* b1, const uint16_t* w2t_packed,
                        const uint16_t* w1t_packed, uint16_t* a_bf16, uint16_t* h_bf16, uint16_t* dh_bf16,
                        float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                        const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out, uint16_t* dyd_bf16,
                        void* workspace, size_t workspace_bytes, csship_stream_t stream);
/* csship_ffn_bwd_ln: the same with the backward of a SECOND LayerNorm behind it, for x = LayerNorm2(ln2_x) (the input of
 * a block's first FeedForward is the previous block's final LayerNorm, conformer.py:360): dx receives
 * LayerNorm2-backward(dx) - the gradient wrt ln2_x - and dgamma_dbeta2 (f32 [2, 256]) that LayerNorm's parameter
 * gradients; replaces a csship_layernorm_bwd launch and its read of ln2_x and dx.  ln2_x = NULL: csship_ffn_bwd_drop.
 * Whole 64-row panels only (M % 64 == 0, d_ffn <= 2048; CSSHIP_EUNSUPPORTED otherwise).
 * Workspace: csship_ffn_bwd_ln_workspace(M). */
size_t csship_ffn_bwd_ln_workspace(int32_t M);
int csship_ffn_bwd_ln(const float* x, const float* dy, const float* ln_g, const float* ln_b,
                      const uint16_t* w1_packed, const float* b1, const uint16_t* w2t_packed,
                      const uint16_t* w1t_packed, uint16_t* a_bf16, uint16_t* h_bf16, uint16_t* dh_bf16,
                      float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                      const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out,
                      uint16_t* dyd_bf16, const float* ln2_x, const float* ln2_g, float* dgamma_dbeta2,
                      void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* ---- attention, training (css_amd/csrc/attn.hip).  csship_mha_relpos_train = csship_mha_relpos that also returns
 * lse[b, h, i] = log sum_j exp(score[i, j]) (f32 [B, H, T]).  csship_mha_relpos_probs recomputes the softmax for the
 * backward pass: p[b, h, i, j] = exp(score[i, j] - lse[b, h, i]) as bf16 [B, H, T, ld_p] (ld_p >= T, ld_p % 4 == 0;
 * columns T..ld_p-1 are written as exact zeros up to the last multiple of 4 below ld_p). */
int csship_mha_relpos_train(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                            uint16_t* out, float* lse, int32_t B, int32_t T, int32_t H, csship_stream_t stream);
int csship_mha_relpos_probs(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                            const float* lse, uint16_t* p_bf16, int64_t ld_p, int32_t B, int32_t T, int32_t H,
                            csship_stream_t stream);

/* Training forms with dropout of the attention probabilities (conformer.py:249): out = (mask / (1 - p) o softmax(.)) v.
 * Element index of P[b, h, i, j] for the mask: ((b*H + h)*T + i) * pad8(T) + j.  csship_mha_relpos_probs_drop also
 * writes the masked, rescaled probabilities (operand of dV); with dropout ld_p must be pad8(T). */
int csship_mha_relpos_train_drop(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                 uint16_t* out, float* lse, int32_t B, int32_t T, int32_t H,
                                 const csship_dropout_cfg* drop, csship_stream_t stream);
int csship_mha_relpos_probs_drop(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                                 const float* lse, uint16_t* p_bf16, uint16_t* p_dropped_bf16, int64_t ld_p,
                                 int32_t B, int32_t T, int32_t H, const csship_dropout_cfg* drop, csship_stream_t stream);
/* dS = P o (mask / (1 - p) o (dO v^T) - D) */
int csship_attn_softmax_bwd_drop(const uint16_t* qkv, const uint16_t* d_out, const uint16_t* p_bf16, int64_t ld_p,
                                 const float* d, uint16_t* ds_bf16, uint16_t* ds_skew_bf16, int32_t W, int32_t B,
                                 int32_t T, int32_t H, const csship_dropout_cfg* drop, csship_stream_t stream);

/* csship_mha_relpos_bwd_core: csship_mha_relpos_probs(_drop) + csship_attn_softmax_bwd(_drop) + the product dS k in ONE
 * pass over the scores (the probabilities never make a round trip through memory on their way to dS):
 *   p_v_bf16  [B,H,T,ld_p] = the operand of dV = p_v^T dO (the probabilities; with dropout mask / (1 - p) o P),
 *   ds_bf16   [B,H,T,ld_p] = dS = P o (mask / (1 - p) o (dO v^T) - d),  ds_skew_bf16 [B,T,H,W] its skewed copy
 *   (csship_relpos_softmax_bwd; zero outside the band: the caller zeroes the buffer once) - or NULL: no skewed copy
 *   is written (csship_attn_pos_bwd_ds skews the row-major dS itself; W is ignored),
 *   dq1       f32 [B,H,T,64] = dS k (the key part of the query gradient).
 * d_out bf16 [B*T, H*64]; d f32 [B,H,T] = rowsum(dO o O) (csship_attn_rowdot) - or NULL with out = the forward's
 * output bf16 [B*T, H*64], from which the pass computes it itself; q_rows (optional) bf16 [B*T*H, 64] receives the q
 * rows in ds_skew's row order (b, t, h), the second operand of d(pe_window) = ds_skew^T q; lse from
 * csship_mha_relpos_train(_drop); drop may be NULL; with dropout ld_p must be pad8(T) (the mask's element index, as in
 * the forward). */
int csship_mha_relpos_bwd_core(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max,
                               const float* lse, const uint16_t* d_out, const float* d, const uint16_t* out,
                               uint16_t* p_v_bf16, uint16_t* ds_bf16, uint16_t* ds_skew_bf16, float* dq1, uint16_t* q_rows,
                               int64_t ld_p, int32_t W, int32_t B, int32_t T, int32_t H, const csship_dropout_cfg* drop,
                               csship_stream_t stream);

/* csship_attn_softmax_bwd: dS[b,h,i,j] = P[i,j] * (dO_i . v_j - d[b,h,i]) computed attention-shaped (V tiles in LDS,
 * dO rows as MFMA operand), no [B,H,T,T] f32 intermediate.  qkv as in csship_mha_relpos; d_out bf16 [B*T, H*64];
 * p_bf16 / ds_bf16: [B,H,T,ld_p]; d: f32 [B,H,T] (csship_attn_rowdot); ds_skew_bf16 [B,T,H,W] as in
 * csship_relpos_softmax_bwd, but only the band entries are written (zero the buffer once). */
int csship_attn_softmax_bwd(const uint16_t* qkv, const uint16_t* d_out, const uint16_t* p_bf16, int64_t ld_p,
                            const float* d, uint16_t* ds_bf16, uint16_t* ds_skew_bf16, int32_t W, int32_t B,
                            int32_t T, int32_t H, csship_stream_t stream);

/* ---- row kernels of the backward pass (css_amd/csrc/bwd.hip)
 * csship_layernorm_bwd: y = LayerNorm(x) [then ReLU if relu != 0]; given dy returns dx (+ dx_add, may alias dx or dy)
 * and dgamma_dbeta = f32 [2, D] (dgamma row, dbeta row).  D <= 1024 (vector path for 256 / 512 / 1024); x, dy, dx
 * f32 [M, D] dense. */
size_t csship_layernorm_bwd_workspace(int64_t M, int32_t D);
int csship_layernorm_bwd(const float* x, const float* dy, const float* gamma, const float* beta,
                         const float* dx_add, int32_t relu, int64_t M, int32_t D, float* dx,
                         float* dgamma_dbeta, void* workspace, size_t workspace_bytes, csship_stream_t stream);
/* csship_layernorm_bwd_y: the same, and y_bf16 (optional, bf16 [M, D], D = 256 / 512 / 1024) receives LayerNorm(x) itself
 * (before the ReLU): the backward of "LayerNorm -> Linear" needs it as the second operand of the Linear's weight
 * gradient, and this pass has just recomputed the normalised rows. */
int csship_layernorm_bwd_y(const float* x, const float* dy, const float* gamma, const float* beta,
                           const float* dx_add, int32_t relu, int64_t M, int32_t D, float* dx,
                           float* dgamma_dbeta, uint16_t* y_bf16, void* workspace, size_t workspace_bytes,
                           csship_stream_t stream);
/* out = dy where y > 0, else 0 (mask head ReLU, conformer.py:99); 2-D with row strides; out may alias dy. */
int csship_relu_bwd(const float* dy, int64_t ld_dy, const float* y, int64_t ld_y, int64_t M, int32_t N,
                    float* out, int64_t ld_out, csship_stream_t stream);
/* The same into bf16 rows zero padded to ld_out (even, >= N) columns: with ld_out a multiple of 8 the two products of
 * the mask head's backward (dW = dz^T x over the B*T rows, dx = dz W) take whole 8-element groups although
 * n_out = 3 * 257 does not. */
int csship_relu_bwd_bf16(const float* dy, int64_t ld_dy, const float* y, int64_t ld_y, int64_t M, int32_t N,
                         uint16_t* out_bf16, int64_t ld_out, csship_stream_t stream);
/* out_bf16[b*T + t, :] = MVN(x) as csship_mvn_stats defines it, zero padded to ld_out columns: the activation operand
 * of the embed weight gradient. */
int csship_mvn_apply(const float* x, int64_t B, int64_t T, int64_t F, int64_t ldx, float clamp_min,
                     const float* mean, const float* rinv, uint16_t* out_bf16, int64_t ld_out,
                     csship_stream_t stream);
/* Attention backward brackets (conformer.py:229-253 under autograd).  With z = (b, h):
 * csship_relpos_softmax: P[z,i,j] = softmax_j(s_qk[z,i,j] + rel[z,i, i-j+T-1]) where rel = q . pe_window^T,
 *   pe_window[r] = pe_k[clamp(r-(T-1))] ; bf16 out, columns T..ld_p-1 zeroed.
 * csship_relpos_softmax_bwd: dS = P o (dP - rowsum(P o dP)) as bf16 [B,H,T,ld_p] and its skewed copy
 *   ds_skew[b, i, h, r] = dS[b, h, i, i+T-1-r] (bf16 [B,T,H,W], W >= 2T-1, zero outside), which turns the
 *   relative-position gradients (dq += dS_skew . pe_window, dpe_window = dS_skew^T . q) into plain GEMMs. */
int csship_relpos_softmax(const float* s_qk, int64_t ld_s, const float* rel, int64_t ld_rel, int32_t B,
                          int32_t H, int32_t T, uint16_t* p_bf16, int64_t ld_p, csship_stream_t stream);
/* D[b, h, i] = sum_d dO[b, i, h, d] * O[b, i, h, d]  (= rowsum(P o dP)); dO, O: bf16 [B*T, H*64]; out f32 [B, H, T]. */
int csship_attn_rowdot(const uint16_t* d_out, const uint16_t* out, int32_t B, int32_t T, int32_t H, float* d,
                       csship_stream_t stream);
int csship_relpos_softmax_bwd(const uint16_t* p_bf16, int64_t ld_p, const float* dp, int64_t ld_dp,
                              int32_t B, int32_t H, int32_t T, uint16_t* ds_bf16, uint16_t* ds_skew_bf16,
                              int32_t W, csship_stream_t stream);

/* ---- csship_convmod_bwd (css_amd/csrc/convmod.hip): backward of csship_convmod run with bn_mode = 1 (batch
 * statistics; bn_eval = 0) or bn_mode = 2 (running statistics, activations kept; bn_eval = 1).
 * fwd_workspace is that forward call's workspace, untouched since (it holds the pre-BatchNorm activations and the
 * batch statistics: after the call its last 4*D floats are [scale | shift | batch mean | biased batch variance]).
 * dx = dy + (gradient through the module).  The parameter gradients are written to ONE f32 buffer `grads` of
 * 38*D + 8 floats, laid out (offsets in floats):
 *   0: bn_b[D]   D: bn_g[D]   2D: pw2_w   2D+1: pw2_b   (2D+2, 2D+3 unused)
 *   2D+4: dw_w[k=33][D]   35D+4: dw_b[D]
 *   36D+4: ln_g[D]   37D+4: ln_b[D]   38D+4: pw1_w[2]   38D+6: pw1_b[2] */
size_t csship_convmod_bwd_workspace(int32_t B, int32_t T, int32_t D);
int csship_convmod_bwd(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                       int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx,
                       float* grads, int32_t bn_eval, void* workspace, size_t workspace_bytes,
                       csship_stream_t stream);
/* ... of csship_convmod_drop (same cfg): the branch gradient is dy o mask / (1 - p), the residual path dy. */
int csship_convmod_bwd_drop(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                            int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx,
                            float* grads, int32_t bn_eval, const csship_dropout_cfg* drop, void* workspace,
                            size_t workspace_bytes, csship_stream_t stream);

/* csship_convmod_bwd_drop that also writes dx_next_bf16 = bf16(mask_next / (1 - p_next) o dx) (bf16 [B*T, D]; next_drop
 * NULL: bf16(dx)): the operand the NEXT operator of the backward pass multiplies with - in the Conformer block the
 * attention output projection, whose weight and input gradients take the gradient through ITS output dropout
 * (conformer.py:254) as bf16 MFMA operands.  Saves that operator a dropout pass over dx and the fp32 -> bf16 rounding in
 * its two products (same rounding, same results). */
int csship_convmod_bwd_next(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                            int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx,
                            float* grads, int32_t bn_eval, const csship_dropout_cfg* drop,
                            const csship_dropout_cfg* next_drop, uint16_t* dx_next_bf16, void* workspace,
                            size_t workspace_bytes, csship_stream_t stream);

/* SyncBatchNorm backward, split like the forward: _stats leaves this rank's S1 = sum dy_bn and S2 = sum dy_bn * zhat
 * (f32 [2D]) at byte csship_convmod_bwd_sums_offset() of `workspace`; the caller all-reduces them (SUM); _finish uses
 * them with 1 / total_rows in the BatchNorm input gradient.  bn_g / bn_b gradients in `grads` are this rank's share
 * (the usual gradient all-reduce completes them).  Same workspace for both calls. */
size_t csship_convmod_bwd_sums_offset(int32_t B, int32_t T, int32_t D);
int csship_convmod_bwd_stats(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                             int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx, float* grads,
                             const csship_dropout_cfg* drop, void* workspace, size_t workspace_bytes,
                             csship_stream_t stream);
int csship_convmod_bwd_finish(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                              int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx, float* grads,
                              double total_rows, const csship_dropout_cfg* drop, void* workspace,
                              size_t workspace_bytes, csship_stream_t stream);

/* Weight shadows for one Linear in ONE launch: f32 w[R, K] (row stride ld_w), every element times `scale`, written
 * as bf16 row-major rows dst_row0.. of a [.., ld_row_major] matrix (optional; the kmajor operand of csship_bgemm)
 * and/or in MFMA fragment order (optional; rows dst_row0.. of the packed matrix, dst_row0 % 16 == 0, K rounded up to
 * 32, rows rounded up to 16 with zeros).  Fused q/k/v weights are three calls with dst_row0 = 0, D, 2D. */
int csship_pack_weight(const float* w, int32_t R, int32_t K, int64_t ld_w, float scale,
                       uint16_t* row_major_bf16, int64_t ld_row_major, uint16_t* frag_packed_bf16,
                       int32_t dst_row0, csship_stream_t stream);

/* The same for a table of weights in one launch.  jobs_device: n_jobs records IN DEVICE MEMORY (the table is built
 * once per model: parameter storage does not move between steps).  Per job: w f32, logical [R, K] with
 * W[r][k] = w[r * ld_w + k] (transposed = 0) or w[k * ld_w + r] (transposed = 1); Kp = K rounded up to 32 when frag
 * is given, else to 8; the other fields as in csship_pack_weight.  max_chunks = max over jobs of
 * roundup(R, 16) * Kp / 8. */
typedef struct csship_pack_job {
  const float* w;
  uint16_t* row_major;
  uint16_t* frag;
  int64_t ld_w, ld_row_major;
  int32_t R, K, dst_row0, Kp;
  float scale;
  int32_t transposed;
  int32_t k0, Kp_dst;  /* frag only: this job fills the k-slice [k0, k0 + Kp) of a table Kp_dst wide (k0 % 32 == 0);
                          Kp_dst = 0: the table is Kp wide (k0 must be 0) */
} csship_pack_job;
int csship_pack_weights_multi(const csship_pack_job* jobs_device, int32_t n_jobs, int64_t max_chunks,
                              csship_stream_t stream);

/* csship_ffn_bwd_conv (round 6): csship_ffn_bwd_drop for the FeedForward BEHIND a conv module (a block's feed_forward_out,
 * conformer.py:357-358) whose dx is that module's output gradient: the statistics pass of csship_convmod_bwd (per channel
 * S1 = sum dzh, S2 = sum dzh * zn, and d pw2_w, d pw2_b; torch.autograd through BatchNorm1d / ReLU / the scalar affine in the
 * reference, conformer.py:293-301) rides in the kernel's epilogue on the dx rows it has just formed, and its reduction
 * leaves them where the module's own pass leaves them: the first 2 D + 2 floats of `conv_grads` (CSSHIP_CONV_GRADS layout:
 * d bn_b | d bn_g | d pw2_w | d pw2_b as S1 | S2 | .. until csship_convmod_bwd_after_stats finishes them).
 * conv_fwd_workspace: the module's training workspace (csship_convmod bn_mode 1 / 2, csship_outproj_conv_train);
 * conv_drop: its output dropout site (NULL: none).  M = B * T, M % 64 == 0, d_model 256, d_ffn <= csship_ffn_static_max_ffn().
 * csship_convmod_bwd_after_stats: csship_convmod_bwd_next (dx_next_bf16 may be NULL) without its statistics pass, reading the
 * sums from `grads`. */
size_t csship_ffn_bwd_conv_workspace(int32_t M);
int csship_ffn_bwd_conv(const float* x, const float* dy, const float* ln_g, const float* ln_b, const uint16_t* w1_packed,
                        const float* b1, const uint16_t* w2t_packed, const uint16_t* w1t_packed, uint16_t* a_bf16,
                        uint16_t* h_bf16, uint16_t* dh_bf16, float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model,
                        int32_t d_ffn, float alpha, const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out,
                        uint16_t* dyd_bf16, const void* conv_fwd_workspace, int32_t B, int32_t T, const float* conv_pw2_w,
                        const csship_dropout_cfg* conv_drop, float* conv_grads, void* workspace, size_t workspace_bytes,
                        csship_stream_t stream);
int csship_convmod_bwd_after_stats(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                                   int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx, float* grads,
                                   int32_t bn_eval, const csship_dropout_cfg* drop, const csship_dropout_cfg* next_drop,
                                   uint16_t* dx_next_bf16, void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* ---- csship_qkv_in_bwd (css_amd/csrc/panel.hip): the input side of the q | k | v projection's backward in one kernel
 * per 64-row panel.  Forward (conformer.py:213-224,356): qkv = LayerNorm(x) Wqkv^T + b, out = x + attention(qkv) ...;
 * given dqkv (bf16 [M, 768]) and dy = d(out) (f32 [M, 256]):
 *   da = dqkv Wqkv (wqkvT_packed = MFMA fragments of Wqkv^T [256, 768], the q rows of Wqkv carrying the forward's
 *   1/sqrt(d_k)); dx = dy + LayerNorm-backward(da); y_bf16 = LayerNorm(x) (operand of dWqkv = dqkv^T y);
 *   dgamma_dbeta = f32 [2, 256].
 * d_model 256, M % 64 == 0 (CSSHIP_EUNSUPPORTED otherwise: csship_bgemm + csship_layernorm_bwd_y serve those).
 * dx may alias dy. */
size_t csship_qkv_in_bwd_workspace(int32_t M);
int csship_qkv_in_bwd(const uint16_t* dqkv, const uint16_t* wqkvT_packed, const float* x, const float* dy,
                      const float* ln_g, const float* ln_b, float* dx, uint16_t* y_bf16, float* dgamma_dbeta,
                      int32_t M, int32_t d_model, void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* csship_qkv_ffn_bwd (round 6): csship_qkv_in_bwd and the csship_ffn_bwd_ln BEHIND it in the backward pass - the block's
 * feed_forward_in, whose output x_attn is (conformer.py:355-356, torch.autograd through both in the reference) - as ONE
 * kernel per 64-row panel: the first stage's dx is the second's dy and stays in registers (no f32 [M, 256] round trip, one
 * launch less per block).  Arguments: csship_qkv_in_bwd's without dx (x_attn = its x), then csship_ffn_bwd_ln's without
 * dy.  dyd_bf16 is required (dy itself exists nowhere: dW2 = dyd^T h); ln2_x may be NULL (no second LayerNorm).  d_model 256, M % 64 == 0, d_ffn % 256 == 0, d_ffn <= csship_ffn_static_max_ffn(). */
size_t csship_qkv_ffn_bwd_workspace(int32_t M);
int csship_qkv_ffn_bwd(const uint16_t* dqkv, const uint16_t* wqkvT_packed, const float* x_attn, const float* dy,
                       const float* attn_ln_g, const float* attn_ln_b, uint16_t* y_bf16, float* attn_dgamma_dbeta,
                       const float* x, const float* ln_g, const float* ln_b, const uint16_t* w1_packed, const float* b1,
                       const uint16_t* w2t_packed, const uint16_t* w1t_packed, uint16_t* a_bf16, uint16_t* h_bf16,
                       uint16_t* dh_bf16, float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model, int32_t d_ffn,
                       float alpha, const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out,
                       uint16_t* dyd_bf16, const float* ln2_x, const float* ln2_g, float* dgamma_dbeta2, void* workspace,
                       size_t workspace_bytes, csship_stream_t stream);

/* csship_reduce_side_stream (round 6, experiment): a second stream of the caller's for launches that yield PARAMETER gradients
 * only - the flush of the queued reductions, reductions marked deferrable that are too large to queue, the relative-position
 * window reduction: they run on `side` behind everything the call's own stream has been given so far, beside what it is given
 * next.  NULL switches the mode off.  Per host thread.  The caller owes csship_reduce_side_join(main) before anyone reads
 * those gradients (and before a stream capture that saw the side stream ends), and keeps every buffer such a launch reads
 * alive until then (the library joins by itself before it overwrites its own slabs). */
int csship_reduce_side_stream(csship_stream_t side);
int csship_reduce_side_join(csship_stream_t main);

/* csship_qkv_ffn2_bwd (round 6): csship_qkv_ffn_bwd with the FeedForward backward of the BLOCK BELOW behind it - that block's
 * feed_forward_out (conformer.py:358-360): the gradient the second stage carries through the block's final LayerNorm (ln2) is
 * its dy - including the conv module's statistics pass of csship_ffn_bwd_conv: three stages of one kernel per 64-row panel, two
 * launches and two f32 [M, 256] round trips less per block than the single operators.  The three argument blocks are the
 * operands of csship_qkv_in_bwd (without dx), of csship_ffn_bwd_ln (without dy; f_in: ln2_* required, dx unused; f_out_below: dx
 * required - the conv module's backward reads it -, ln2_* NULL) and of csship_ffn_bwd_conv's conv part.  dyd_bf16 is required in
 * both FeedForward blocks.  Same shape limits as csship_qkv_ffn_bwd. */
typedef struct {
  const uint16_t* dqkv; const uint16_t* wqkvT_packed; const float* x; const float* dy; const float* ln_g; const float* ln_b;
  uint16_t* y_bf16; float* dgamma_dbeta;
} csship_qkv_bwd_args;
typedef struct {
  const float* x; const float* ln_g; const float* ln_b; const uint16_t* w1_packed; const float* b1; const uint16_t* w2t_packed;
  const uint16_t* w1t_packed; uint16_t* a_bf16; uint16_t* h_bf16; uint16_t* dh_bf16; float* dx; float* dgamma_dbeta;
  const csship_dropout_cfg* drop_hidden; const csship_dropout_cfg* drop_out; uint16_t* dyd_bf16;
  const float* ln2_x; const float* ln2_g; float* dgamma_dbeta2;
} csship_ffn_bwd_args;
typedef struct {
  const void* fwd_workspace; int32_t B, T; const float* pw2_w; const csship_dropout_cfg* drop; float* grads;
} csship_conv_stats_args;
size_t csship_qkv_ffn2_bwd_workspace(int32_t M);
int csship_qkv_ffn2_bwd(const csship_qkv_bwd_args* q, const csship_ffn_bwd_args* f_in, const csship_ffn_bwd_args* f_out_below,
                        const csship_conv_stats_args* conv_below, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                        void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* ---- BLSTM recurrence, training (css_amd/csrc/lstm.hip).  csship_lstm_bidir_train = csship_lstm_bidir that also keeps
 * the activated gates (i, f, g, o; f32 [B, T, 2, 4H]) and the cell states (f32 [B, T, 2, H]).
 * csship_lstm_bidir_bwd: back-propagation through time.  dy: f32 [B, T, 2H]; whhT_packed: both directions'
 * W_hh^T [H, 4H] in MFMA fragment order, direction-major; dgates: f32 [B, T, 2, 4H] = gradient wrt the
 * pre-activation gates, from which dW_ih, dW_hh, the bias gradients and dx follow as plain GEMMs / column sums
 * (csship_bgemm, csship_colsum).  H % 16 == 0, H % 128 == 0 above 128, H <= 512. */
int csship_lstm_bidir_train(const float* gx, const uint16_t* whh_packed, float* out, float* gates_out,
                            float* c_out, int32_t B, int32_t T, int32_t H, csship_stream_t stream);
int csship_lstm_bidir_bwd(const float* gates, const float* c, const float* dy, const uint16_t* whhT_packed,
                          float* dgates, int32_t B, int32_t T, int32_t H, csship_stream_t stream);
/* Back-propagation through time spread over H/16 one-wave workgroups per (16 batch rows, direction) like
 * csship_lstm_bidir_dist (W_hh^T rows in LDS, the previous step's gate gradients exchanged per step behind epoch
 * flags); H = 256 or 512, identical results, same workspace and placement contract. */
int csship_lstm_bidir_bwd_dist(const float* gates, const float* c, const float* dy, const uint16_t* whhT_packed,
                               float* dgates, int32_t B, int32_t T, int32_t H, void* workspace,
                               size_t workspace_bytes, csship_stream_t stream);

/* ---- csship_attn_dvdk (css_amd/csrc/attnbwd.hip): the value and key gradients of the attention in one streaming pass over
 * the two [B, H, T, T] operands the one-pass backward kernel leaves (torch.autograd's two batched matmuls behind
 * conformer.py:236-253 in the reference):
 *   dV[b, j, h, :] = sum_i p_v[b, h, i, j] d_out[b, i, h, :]     dK[b, j, h, :] = sum_i ds[b, h, i, j] q[b, i, h, :]
 * p_v / ds: bf16 [B, H, T, ld_p] (ld_p >= T, a multiple of 8, zeros in [T, ld_p): what csship_mha_relpos_bwd_core writes);
 * d_out: bf16 [B*T, H*64]; qkv: bf16 [B*T, 3*H*64] (q = the first third, as the forward used it); dqkv: bf16
 * [B*T, 3*H*64], of which the k and v thirds are written.  f32 accumulation over the queries in ascending order. */
int csship_attn_dvdk(const uint16_t* p_v, const uint16_t* ds, int64_t ld_p, const uint16_t* d_out, const uint16_t* qkv,
                     uint16_t* dqkv, int32_t B, int32_t T, int32_t H, csship_stream_t stream);
/* csship_attn_pos_bwd: the two products the relative-position term leaves behind the one-pass backward, in one pass
 * over the skewed dS (ds_skew: bf16 [B, T, H, ld_w], entry (b, t, h, w) = dS[b, h, t, j] at w = t - j + T - 1, zero
 * outside w in [t, t + T - 1]; conformer.py:236-241 under torch.autograd):
 *   dqkv's q third [b, t, h, :] = dq1[b, h, t, :] + sum_w ds_skew[b, t, h, w] pe_window[w, :]     (bf16)
 *   dwin_acc[w, :]             += sum_{b, t, h} ds_skew[b, t, h, w] q_rows[(b, t, h), :]            (f32 [W, 64], in place)
 * pe_window_t: bf16 [64, ld_pt] = the window transposed, zero beyond its rows (ld_pt >= 64 (ceil(T / 64) - 1) + 32
 * ceil((T + 63) / 32)); q_rows: bf16 [B*T*H, 64] in ds_skew's row order; dq1: f32 [B, H, T, 64].  d_k = 64,
 * T <= 449 (CSSHIP_EUNSUPPORTED beyond: csship_bgemm with band serves longer chunks), W >= 2 T - 1.  Summation order
 * fixed (deterministic). */
size_t csship_attn_pos_bwd_workspace(int32_t B, int32_t T);
/* csship_attn_pos_bwd_ds: the same from the ROW-MAJOR dS (bf16 [B, H, T, ld_p], the operand of csship_attn_dvdk) - the
 * kernel skews it while it stages its tiles, so csship_mha_relpos_bwd_core may be called with ds_skew_bf16 = NULL and no
 * skewed copy is ever written.  The tile loads are 16-byte pieces at 2-byte alignment that reach up to 14 bytes in front
 * of and behind a row: the caller keeps 16 readable bytes on both sides of the ds buffer. */
int csship_attn_pos_bwd_ds(const uint16_t* ds, int64_t ld_p, const uint16_t* pe_window_t, int64_t ld_pt,
                           const uint16_t* q_rows, const float* dq1, uint16_t* dqkv, float* dwin_acc, int32_t W, int32_t B,
                           int32_t T, int32_t H, void* workspace, size_t workspace_bytes, csship_stream_t stream);
int csship_attn_pos_bwd(const uint16_t* ds_skew, int64_t ld_w, const uint16_t* pe_window_t, int64_t ld_pt,
                        const uint16_t* q_rows, const float* dq1, uint16_t* dqkv, float* dwin_acc, int32_t W, int32_t B,
                        int32_t T, int32_t H, void* workspace, size_t workspace_bytes, csship_stream_t stream);
/* csship_attn_pos_bwd_ds behind csship_mha_relpos_bwd_fused: dq1 arrives as n_parts (1 .. 16) partials f32 [n_parts][B, H, T, 64]
 * (one per key group) that are added here, and the q rows are read in place from the projection (q: bf16, row (b, t) at
 * q + (b T + t) ld_q, head h at + 64 h; ld_q = 3 H 64 inside qkv) - no q_rows copy. */
int csship_attn_pos_bwd_ds_parts(const uint16_t* ds, int64_t ld_p, const uint16_t* pe_window_t, int64_t ld_pt,
                                 const uint16_t* q, int64_t ld_q, const float* dq1_parts, int32_t n_parts, uint16_t* dqkv,
                                 float* dwin_acc, int32_t W, int32_t B, int32_t T, int32_t H, void* workspace,
                                 size_t workspace_bytes, csship_stream_t stream);

/* ---- csship_mha_relpos_bwd_fused (css_amd/csrc/attnfb.hip): the attention backward up to the window axis in ONE
 * key-stationary pass (replaces csship_mha_relpos_bwd_core + csship_attn_dvdk; conformer.py:229-253 under autograd).
 * From q | k | v (bf16 [B*T, 3 H 64], q pre-scaled), the packed position blocks, the training forward's log-sum-exp
 * (f32 [B, H, T]) and output (bf16 [B*T, H 64]) and the output gradient d_out (bf16 [B*T, H 64]) it recomputes the
 * probabilities and writes: dS (bf16 [B, H, T, ld_p], ld_p a multiple of 8 >= T, columns T .. ld_p - 1 zero) - the one
 * [B, H, T, T] tensor left in memory, the operand of csship_attn_pos_bwd_ds_parts -, dq1 = dS k as one f32 [B, H, T, 64]
 * partial per key group (csship_mha_relpos_bwd_fused_parts(B, T, H) of them, contiguous), and dK, dV into the k and v
 * thirds of dqkv (bf16 [B*T, 3 H 64]).  The probabilities' dropout of the forward is regenerated from `drop` (then
 * ld_p == pad8(T) and B H T ld_p < 2^32).  d_k = 64.  Deterministic (fixed summation orders, no atomics). */
int csship_mha_relpos_bwd_fused_parts(int32_t B, int32_t T, int32_t H);
int csship_mha_relpos_bwd_fused(const uint16_t* qkv, const uint16_t* pe_packed, int32_t q_min, int32_t q_max, const float* lse,
                                const uint16_t* d_out, const uint16_t* out, uint16_t* ds_bf16, int64_t ld_p, float* dq1_parts,
                                uint16_t* dqkv, int32_t B, int32_t T, int32_t H, const csship_dropout_cfg* drop,
                                csship_stream_t stream);

/* ---- csship_p2p_reduce (css_amd/csrc/optim.hip): out[lo:hi] = scale * (src_0 + src_1 + ... + src_{world-1})[lo:hi],
 * summed in that order (bit-reproducible), for the direct (one-shot) all-reduce of css_amd.parallel.P2PAllReduce
 * (replaces scripts/python/combine_models.py:55-82): srcs_host is a HOST array of world <= 8 device
 * pointers - this rank's flat gradient buffer and its peers' buffers as mapped into this process with hipIpc - each
 * 16-byte aligned; lo a multiple of 4; out may be one of the sources. */
int csship_p2p_reduce(const float* const* srcs_host, int32_t world, int64_t lo, int64_t hi, float scale, float* out,
                      csship_stream_t stream);

/* ---- csship_grad_accumulate (css_amd/csrc/optim.hip): dst += src for a table of small f32 tensors in ONE launch (the
 * gradients the explicit backward hands to FusedAdam's flat gradient buffer: torch.autograd's AccumulateGrad nodes in the
 * reference).  cols == 0: src dense, n elements.  cols > 0: dst is a dense [n / cols, cols] matrix and receives the
 * TRANSPOSE of src, element (r, c) from src[c * ld_src + r] (the depthwise conv weight, kept [taps, channels] by the
 * kernels).  cols < 0: dst is a dense [n / C, C] matrix, C = -cols, and src has the same rows ld_src >= C elements apart
 * (a product written with a padded row stride).  The table is read on the host and travels in the kernel arguments, so
 * a captured launch replays it. */
typedef struct csship_grad_job {
  const float* src;
  float* dst;
  int64_t n, ld_src;
  int32_t cols, pad_;
} csship_grad_job;
int csship_grad_accumulate(const csship_grad_job* jobs_host, int32_t n_jobs, csship_stream_t stream);
/* dst = keep * dst + w * src for a table of at most 96 DENSE f32 tensors (cols == 0) in one launch, w = w_head for the
 * first n_head jobs and w_tail for the rest: the running mean / running variance update of every BatchNorm1d of the
 * model after a training forward (torch.nn.BatchNorm1d, momentum m: keep = 1 - m, w = m for the means and
 * m * n / (n - 1) for the variances; conformer.py:293).  Evaluated as fma(w, src, keep * dst). */
int csship_ema_update(const csship_grad_job* jobs_host, int32_t n_jobs, float keep, float w_head, int32_t n_head,
                      float w_tail, csship_stream_t stream);

/* ---- csship_adam_clip (css_amd/csrc/optim.hip): clip_grad_norm_(max_norm) followed by one Adam step
 * (iteration_types.py:17-64) on FLAT f32 buffers of n elements (parameters, gradients, first and second moments), in
 * three launches with no host round trip.  grad_norm (device f32) receives the total 2-norm before clipping; a
 * non-finite norm skips the update.  max_norm <= 0 disables clipping; step is the 1-based Adam step count;
 * weight_decay is torch.optim.Adam's L2 form.  hyper_dev (optional, device f32[3] = {lr, 1 - beta1^step,
 * 1 - beta2^step}) overrides lr / step at execution time, so a captured HIP graph of the training step can be replayed
 * while the schedule advances. */
size_t csship_adam_clip_workspace(int64_t n);
int csship_adam_clip(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                     float max_norm, float lr, float beta1, float beta2, float eps, int64_t step,
                     float weight_decay, const float* hyper_dev, float* grad_norm, void* workspace,
                     size_t workspace_bytes, csship_stream_t stream);

/* ======================================================================================
 * Mask-based MVDR beamformer (css_amd/csrc/mvdr.hip; css/executor/beamformer.py:126-183, arithmetic of asteroid's
 * compute_scm + SoudenMVDRBeamformer, also stated in css/utils/mvdr_util.py:45-66).
 * Spectra are re / im planes: x [B, D, T_x, F] (csship_stft's layout) of which the first T frames are used, masks
 * [B, >=T, F] (row stride ld_mask, batch stride mask_batch_stride), results y [B, F, T] (csship_istft's layout).
 * 2 <= D <= 8.
 * ====================================================================================== */
/* scm[b, f] = sum_t mask x x^H (interleaved complex f32 [B, F, D, D, 2]) and mask_sum[b, f]; normalize != 0 divides by
 * mask_sum (asteroid), diag_load is added to the diagonal afterwards (mvdr_util.py:51). */
int csship_mvdr_scm(const float* x_re, const float* x_im, const float* mask, int64_t ld_mask,
                    int64_t mask_batch_stride, int32_t B, int32_t D, int32_t T, int32_t T_x, int32_t F,
                    float diag_load, int32_t normalize, float* scm, float* mask_sum, csship_stream_t stream);
/* The same for n_masks <= 4 masks over the same spectra in ONE launch (HOST arrays of device pointers; equal strides):
 * the noise and the two speech covariances of a beamformer batch (beamformer.py:151-159). */
int csship_mvdr_scm_multi(const float* x_re, const float* x_im, const float* const* masks, int32_t n_masks, int64_t ld_mask,
                          int64_t mask_batch_stride, int32_t B, int32_t D, int32_t T, int32_t T_x, int32_t F, float diag_load,
                          int32_t normalize, float* const* scms, float* const* mask_sums, csship_stream_t stream);
/* Souden MVDR weights w[bin] = Z[:, ref_mic] / (trace(Z) + trace_eps), Z = (noise_scm [+ noise_scm2])^-1 target_scm,
 * for n_bins = B*F bins (fp64 elimination with partial pivoting); w: interleaved complex f32 [n_bins, D, 2]. */
int csship_mvdr_souden(const float* target_scm, const float* noise_scm, const float* noise_scm2, int64_t n_bins,
                       int32_t D, int32_t ref_mic, float trace_eps, float* w, csship_stream_t stream);
/* y = w^H x per bin; energies[b] = {sum |y|^2, sum |mask * x_0|^2} (mask may be NULL: second sum is 0). */
size_t csship_mvdr_apply_workspace(int32_t B, int32_t T, int32_t F);
int csship_mvdr_apply(const float* x_re, const float* x_im, const float* w, const float* mask, int64_t ld_mask,
                      int64_t mask_batch_stride, int32_t B, int32_t D, int32_t T, int32_t T_x, int32_t F,
                      float* y_re, float* y_im, float* energies, void* workspace, size_t workspace_bytes,
                      csship_stream_t stream);
/* In place on the two sources' outputs: energy rescale (beamformer.py:165-183) and the deduplication gain of
 * beamformer.py:67-80 (15 dB power gap -> quieter source times clamp(|y_quiet| / max(|y_0|, |y_1|), 0.01)). */
int csship_mvdr_finish(float* y0_re, float* y0_im, float* y1_re, float* y1_im, const float* energies0,
                       const float* energies1, int32_t rescale, int32_t dedup, int32_t B, int64_t per_batch,
                       csship_stream_t stream);

/* ---- DOA likelihood of the 7-channel mask merge (css_amd/csrc/doa.hip; css/executor/separator.py:206-250 with
 * softmax = False, as angle_merge calls it).  x re / im planes [B, 7, F, T]; mask f32 [B, F, T, 2] (binarised at
 * binarize_thr); steering vectors re / im [F, n_angles, 7] (n_angles = 30, separator.py:52); band [lo, hi).
 * likelihood: f32 [B, 2, n_angles]; masked_energy: f32 [B, 2] = sum_{f,t} binmask * |x_0| (separator.py:196-200). */
size_t csship_doa_workspace(int32_t B, int32_t F, int32_t lo, int32_t hi, int32_t n_angles);
int csship_doa_likelihood(const float* x_re, const float* x_im, const float* mask, const float* sv_re,
                          const float* sv_im, int32_t B, int32_t F, int32_t T, int32_t n_angles, int32_t lo,
                          int32_t hi, float binarize_thr, float compression, float* likelihood,
                          float* masked_energy, void* workspace, size_t workspace_bytes, csship_stream_t stream);

/* ---- diagnostics (css_amd/csrc/forward.hip; not on the product path: bench.py's in-step roofline and the stamp tools).
 * csship_debug_timeline(1) starts a fresh per-launch timeline: instrumented entry points then bracket each kernel launch
 * with a HIP event pair on the launch stream (not usable during graph capture); (0) stops it and frees the events.
 * csship_debug_timeline_read: after the caller synchronised the stream, total milliseconds and number of launches
 * recorded under `tag` (CSSHIP_TL_* in css_amd/csrc/common.hpp: 0 chain, 1 chain edge, 2 attention, 3 out-proj+conv,
 * 4 embed, 5 head, 6 stft, 7 pit sse, 8 pit finalize, 9 mvn, 10 other).
 * csship_debug_timeline_calibrate: records `pairs` EMPTY event pairs (nothing launched between the two records) on
 * `stream` under tag 11: their mean elapsed time is what a bracket adds to every launch it times (two timestamp packets
 * the launch is serialised behind); bench.py subtracts it, so that the per-launch times add up to no more than the step.
 * csship_debug_stamps: device buffer that -D*_STAMPS diagnostic builds fill with s_memtime phase stamps; NULL = off. */
void csship_debug_timeline(int enable);
int csship_debug_timeline_read(int tag, double* total_ms, int* count);
void csship_debug_timeline_calibrate(csship_stream_t stream, int pairs);
void csship_debug_stamps(void* device_buffer);

#ifdef __cplusplus
}
#endif
#endif /* CSSHIP_H */
