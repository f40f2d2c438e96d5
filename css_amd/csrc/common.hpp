// Shared device helpers for libcsship (gfx950 only: wave = 64 lanes, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/csship.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DEV static __device__ __forceinline__

DEV unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
DEV float bf2f(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
// two f32 -> packed bf16 pair (round to nearest even): ONE v_cvt_pk_bf16_f32 (two scalar casts + shift/or took 3-4)
typedef __attribute__((ext_vector_type(2))) float f32x2_cvt;
DEV unsigned pack_bf2(float lo, float hi) {
  // ONE vector fptrunc: two scalar casts + a build_vector are fused into v_cvt_pk_bf16_f32 only some of the time (values
  // coming out of MFMA accumulators were converted one by one against a zero and merged with v_perm_b32: 3 instructions)
  const f32x2_cvt f = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
// max(x, 0) on a packed bf16 pair: as signed 16-bit integers every negative bf16 (sign bit set, -0 included) is < 0
DEV unsigned relu_bf2(unsigned p) {
  const s16x2 z = {0, 0};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), z));
}

// Cross-lane reductions without the LDS crossbar: __shfl_xor compiles to ds_bpermute_b32 (an LDS round trip of
// ~120 cycles per step in a dependent chain); DPP quad permutes / mirrors and v_permlane16/32_swap are plain VALU.
// All lanes of an already-reduced group hold the same value, so a mirror (lane i <-> 7-i, 15-i) pairs the same
// partial sums as xor 4 / xor 8 would: the results are bit-identical to the xor-butterfly they replace.
template <int CTRL>
DEV float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
DEV float sum8(float v) {  // over aligned groups of 8 lanes, result in all of them
  v += dpp_mov<DPP_XOR1>(v);
  v += dpp_mov<DPP_XOR2>(v);
  v += dpp_mov<DPP_HALF_MIRROR>(v);
  return v;
}
DEV float max8(float v) {
  v = fmaxf(v, dpp_mov<DPP_XOR1>(v));
  v = fmaxf(v, dpp_mov<DPP_XOR2>(v));
  v = fmaxf(v, dpp_mov<DPP_HALF_MIRROR>(v));
  return v;
}
DEV float rows_sum4(float x) {  // sum over the four 16-lane rows of the wave (same column), in all lanes
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const unsigned a0 = r[0], a1 = r[1];
  const float y = __uint_as_float(a0) + __uint_as_float(a1);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  const unsigned b0 = q[0], b1 = q[1];
  return __uint_as_float(b0) + __uint_as_float(b1);
}
DEV float rows_max4(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const unsigned a0 = r[0], a1 = r[1];
  const float y = fmaxf(__uint_as_float(a0), __uint_as_float(a1));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  const unsigned b0 = q[0], b1 = q[1];
  return fmaxf(__uint_as_float(b0), __uint_as_float(b1));
}
DEV float wave_sum(float v) {
  v = sum8(v);
  v += dpp_mov<DPP_MIRROR>(v);
  return rows_sum4(v);
}
DEV float wave_max(float v) {
  v = max8(v);
  v = fmaxf(v, dpp_mov<DPP_MIRROR>(v));
  return rows_max4(v);
}

// LDS tile of bf16 rows of 64 elements (128 B, eight 16-byte chunks).  Chunk c of row r
// lives at r*128 + ((c ^ ((r>>1)&7))<<4): 16 consecutive rows reading one logical chunk
// with ds_read_b128 touch 16 distinct 16-byte slots of the 256-byte bank row.
DEV int tile_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

static inline hipStream_t as_stream(csship_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define CSSHIP_CHECK_LAUNCH()                          \
  do {                                                 \
    if (hipGetLastError() != hipSuccess) return CSSHIP_ELAUNCH; \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) == 0; }

// out[r, c] = alpha * sum_s slabs[s][r*N + c] (+ bias[c]) (+ add[r*ldadd + c]), fixed order (bgemm.hip)
// deferrable: the output is a parameter gradient nobody reads before the caller's csship_reduce_flush() - between
// csship_reduce_defer(1) and that flush such reductions are queued (per host thread) and run as ONE launch; the slabs
// must stay untouched until then.
int launch_slab_reduce(const float* slabs, int S, int64_t n, int N, int64_t ldc, float alpha, const float* bias,
                       const float* add, int64_t ldadd, float* out, hipStream_t s, bool deferrable = false);

// csship_outproj_conv_train (convmod.hip) = this launch (panel.hip: output projection + dropout + residual, LayerNorm, GLU
// and the depthwise convolution in one kernel, batch-statistics partials) + the BatchNorm reduce / apply kernels
int launch_outproj_conv_train(const uint16_t* attn_out, const uint16_t* wo_packed, const float* bo, const float* x,
                              float* x2_out, const csship_conv_weights& conv, int B, int T, csship_dropout_cfg drop,
                              float* zbuf, float* partial, int* nblk_out, hipStream_t s);

// Optional side stream for launches that yield PARAMETER gradients only (csship_reduce_side_stream, bgemm.hip; off unless the
// caller sets it): side_fork makes the side stream wait for everything `main` has been given so far and returns it (or `main`
// itself when the mode is off); side_join makes `main` wait for everything the side stream has been given (no-op when nothing
// is outstanding).  Per host thread, like the reduction queues.
hipStream_t side_fork(hipStream_t main);
void side_join(hipStream_t main);

// floats from the start of csship_convmod's training workspace (= the pre-BatchNorm activations z [B, T, D]) to its
// statistics block scale | shift | mean | variance (convmod.hip owns the layout; csship_ffn_bwd_conv reads both)
size_t convmod_fwd_stats_offset(int B, int T, int D);

// Optional in-kernel phase stamps (diagnostic only; nullptr in normal operation): thread 0 of every
// workgroup stores s_memrealtime (100 MHz) at up to 8 phase boundaries.  Set by csship_debug_stamps().
extern unsigned long long* g_csship_stamps;
DEV void stamp(unsigned long long* st, int slot) {
  if (st != nullptr && threadIdx.x == 0) {
    const size_t wg = (size_t)blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
    st[wg * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---- dropout (training mode only; css/models/conformer.py:127,249,254,302,317,319, css/models/blstm.py:147,158).
// Counter-based: the four elements 4q .. 4q+3 (row-major indices in their tensor) of dropout site `stream` share ONE
// hash evaluation: w0 = hash32(key + q), w1 = hash_derive(w0); element 4q + f is kept iff field f of (w0 lo16, w0 hi16,
// w1 lo16, w1 hi16) is >= thresh; key = hash32(seed ^ stream * 0x9E3779B9), thresh = round(p * 65536).
// (Rounds 2-5 spent one hash32 on every PAIR of elements: the two quarter-rate 32-bit multiplies of the mixer per pair
// were 0.25 ms of the 2.2 ms train-mode forward, bench.py --dropout 0; the second word now costs one full-rate 24-bit
// multiply.  tools/dropout_stats.py: keep rate, serial / row / column / cross-stream / intra-group statistics of the
// new masks are those of the old ones.)
// Forward and backward regenerate the same mask from (seed, stream, e); nothing is stored.  The seed lives in DEVICE
// memory so that a captured training step can be replayed with a fresh mask (the host bumps it before each replay).
// oracle/dropout.py is the numpy statement of the same function (the oracle applies identical masks in the tests).
DEV unsigned hash32(unsigned x) {  // "lowbias32" avalanche mixer
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
struct Drop {  // device-side view of csship_dropout_cfg; thresh == 0 means "no dropout"
  unsigned key, thresh;
  float scale;
  unsigned tb;   // (thresh ^ 0x8000) in both halves: the threshold of the packed signed comparison (drop_mask_bf4)
};
DEV Drop drop_make(const unsigned* seed, unsigned stream, float p) {
  Drop d{0u, 0u, 1.f, 0x80008000u};
  if (seed != nullptr && p > 0.f) {
    // wave-uniform by construction: kept in SCALAR registers (a chain kernel carries four of these; as vector registers
    // they were 16 of the 256 a wave has, and the pipelined training form spilled 7)
    d.key = (unsigned)__builtin_amdgcn_readfirstlane((int)hash32(*seed ^ (stream * 0x9E3779B9u)));
    d.thresh = (unsigned)(p * 65536.f + 0.5f);
    d.scale = 65536.f / (float)(65536u - d.thresh);  // 1 / (1 - p_effective)
    d.tb = ((d.thresh ^ 0x8000u) & 0xffffu) * 0x10001u;
  }
  return d;
}
// second word of a four-element group: one 24-bit multiply (v_mul_u32_u24, full rate) + shift-add + xorshift of the
// first - not a hash on its own, but its two fields are statistically independent of w0's (tools/dropout_stats.py)
DEV unsigned hash_derive(unsigned w) {
  unsigned v = __umul24(w ^ 0x5bd1e995u, 0x85ebcau) + (w >> 11);
  return v ^ (v >> 14);
}
// keep-bits of elements e0 .. e0+3 (e0 a multiple of 4): bit j set = element e0 + j survives
DEV unsigned drop_keep4(const Drop& d, unsigned e0) {
  const unsigned w0 = hash32(d.key + (e0 >> 2)), w1 = hash_derive(w0);
  return ((w0 & 0xffffu) >= d.thresh ? 1u : 0u) | ((w0 >> 16) >= d.thresh ? 2u : 0u) |
         ((w1 & 0xffffu) >= d.thresh ? 4u : 0u) | ((w1 >> 16) >= d.thresh ? 8u : 0u);
}
// the same four decisions as AND-masks for two packed bf16 pairs (0xffff over a kept half): field >= thresh as a packed
// SIGNED comparison of the biased values - v_xor, v_pk_sub_i16 clamp (saturating: the sign of the difference is exact),
// v_pk_ashrrev_i16 15, and the and-not is the caller's one v_bitop3: 4 instructions per pair where a compare + select +
// re-pack per bf16 element took ~14 (the attention's probability dropout: 440 of its 1 335 instructions per key tile)
DEV unsigned drop_keep_halves(unsigned w, unsigned tb) {
  const s16x2 f = __builtin_bit_cast(s16x2, w ^ 0x80008000u);
  const s16x2 d = __builtin_elementwise_sub_sat(f, __builtin_bit_cast(s16x2, tb));
  const s16x2 neg = d >> (s16x2){15, 15};
  return ~__builtin_bit_cast(unsigned, neg);
}
DEV uint2 drop_mask_bf4(const Drop& d, unsigned e0) {   // e0 a multiple of 4; .x covers elements e0, e0 + 1, .y e0 + 2, e0 + 3
  const unsigned w0 = hash32(d.key + (e0 >> 2)), w1 = hash_derive(w0);
  return make_uint2(drop_keep_halves(w0, d.tb), drop_keep_halves(w1, d.tb));
}
// dropout(relu(v)) of four consecutive elements as two packed bf16 pairs: scale, convert, ReLU and mask on the PAIRS
// (same bits as relu -> select per element -> scale -> convert: a negative value times the positive scale stays negative)
DEV uint2 drop_relu_pack4(const Drop& d, unsigned e0, float v0, float v1, float v2, float v3) {
  if (d.thresh == 0u) return make_uint2(relu_bf2(pack_bf2(v0, v1)), relu_bf2(pack_bf2(v2, v3)));   // (wave-uniform)
  const uint2 m = drop_mask_bf4(d, e0);
  return make_uint2(relu_bf2(pack_bf2(v0 * d.scale, v1 * d.scale)) & m.x, relu_bf2(pack_bf2(v2 * d.scale, v3 * d.scale)) & m.y);
}
DEV bool drop_keep1(const Drop& d, unsigned e) {
  const unsigned w0 = hash32(d.key + (e >> 2));
  const unsigned w = (e & 2u) ? hash_derive(w0) : w0;
  return ((e & 1u) ? (w >> 16) : (w & 0xffffu)) >= d.thresh;
}
DEV float4 drop_apply4(const Drop& d, unsigned e0, float4 v) {  // v * keep / (1 - p)
  if (d.thresh == 0u) return v;
  const unsigned k = drop_keep4(d, e0);
  return make_float4((k & 1u) ? v.x * d.scale : 0.f, (k & 2u) ? v.y * d.scale : 0.f, (k & 4u) ? v.z * d.scale : 0.f,
                     (k & 8u) ? v.w * d.scale : 0.f);
}

// Optional per-launch timeline (diagnostic / bench.py's in-step roofline; off in normal operation): when switched on
// with csship_debug_timeline(1), instrumented entry points bracket their launches with a pair of HIP events on the
// launch stream, tagged with a kernel family; csship_debug_timeline_read() sums the elapsed times per tag.  The events
// sit directly around the launch in submission order, so host-side gaps between calls are not inside the bracket.
// Not usable while a HIP graph is being captured (events are recorded eagerly): bench.py runs an un-captured step.
enum {
  CSSHIP_TL_CHAIN = 0,      /* chain_kernel<8,0,1,1,1>: FFN_out + LN -> FFN_in -> LN + q/k/v (interior blocks) */
  CSSHIP_TL_CHAIN_EDGE = 1, /* first / last chain launches (one FeedForward only) */
  CSSHIP_TL_ATTN = 2,
  CSSHIP_TL_OUTPROJ_CONV = 3,
  CSSHIP_TL_EMBED = 4,
  CSSHIP_TL_HEAD = 5,
  CSSHIP_TL_STFT = 6,
  CSSHIP_TL_PIT_SSE = 7,
  CSSHIP_TL_PIT_FINALIZE = 8,
  CSSHIP_TL_MVN = 9,
  CSSHIP_TL_OTHER = 10,
  CSSHIP_TL_EMPTY = 11,     /* csship_debug_timeline_calibrate: event pairs with nothing between them */
  CSSHIP_TL_NTAGS = 12
};
extern bool g_csship_timeline_on;
extern int g_csship_timeline_depth;   // only the OUTERMOST scope brackets (entry points that call entry points)
void csship_timeline_push(int tag, hipStream_t s, bool stop);
struct TimelineScope {
  int tag;
  hipStream_t s;
  bool counted, open;
  TimelineScope(int tag_, hipStream_t s_) : tag(tag_), s(s_), counted(false), open(false) {
    if (g_csship_timeline_on) {
      counted = true;
      if (g_csship_timeline_depth++ == 0) {
        open = true;
        csship_timeline_push(tag, s, false);
      }
    }
  }
  ~TimelineScope() {
    if (counted) --g_csship_timeline_depth;
    if (open) csship_timeline_push(tag, s, true);
  }
};
