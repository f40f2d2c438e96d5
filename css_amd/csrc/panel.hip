// Row-panel kernels for the d_model = 256 encoder: fused FeedForward and LN-prologue linears.
//
// At the headline shape (12 800 tokens) every Linear has K or N = 256, so a classic tiled GEMM
// is all prologue/epilogue.  Here one workgroup owns a PANEL of 64 token rows for a whole
// operator chain and the chip is filled by row panels (200 workgroups on 256 CUs):
//
//   * the panel's activations ([64 rows][256] bf16, 32 KB) live in LDS in a 16-byte-chunk
//     XOR swizzle (chunk ^ (row & 15)): conflict-free ds_read_b128 for MFMA operands;
//   * the products are computed TRANSPOSED, D[feature][row] = W[feature][k] . X^T[k][row], so
//     the WEIGHTS are the MFMA A operand.  They are pre-packed on the host in fragment order
//     (block (16-feature tile, 32-k step) = 64 lanes x 16 B contiguous), so each wave streams
//     its private weight fragments global -> VGPR with perfectly coalesced 1 KB loads, two
//     k-steps ahead, continuing across pass boundaries; weights never touch LDS;
//   * in the transposed accumulator a lane owns ONE token row and 4 consecutive features per
//     tile: residual loads / output stores are 16-byte vectors and LayerNorm statistics are
//     in-lane sums + 2 shuffles + one small LDS exchange between waves;
//   * FeedForward (css/models/conformer.py:322-326 + the 0.5 residual of :355/:358 and the
//     block-final LayerNorm of :360) is ONE kernel: LN -> W1 -> ReLU -> W2 -> residual (-> LN).
//     The 1024-wide hidden activation exists only as four 256-column chunks in LDS.
#include "common.hpp"

namespace {

constexpr int BM = 64;         // token rows per workgroup
constexpr int DM = 256;        // d_model
constexpr int KS = DM / 32;    // MFMA k-steps per 256-deep product
constexpr int PANEL_BYTES = BM * DM * 2;

DEV int panel_off(int row, int chunk) { return row * 512 + ((chunk ^ (row & 15)) << 4); }

// Two transposed-accumulator tiles (16 adjacent columns each) as bf16: a lane (row n, g) holds 4 columns of each (8 bytes
// + 8 bytes, 16 columns apart).  v_permlane16_swap_b32 trades the odd 16-lane rows of one register for the even rows of
// another: after it a lane holds EIGHT CONSECUTIVE columns of ONE tile - columns 8 (g >> 1) .. + 7 of tile (g & 1) - and
// the global store is one 16-byte piece per lane (64 contiguous bytes per token row and instruction) instead of two
// 8-byte ones.  No change to any layout in memory.
DEV uint4 pair_tiles_bf16(uint2 t0, uint2 t1) {
  const auto lo = __builtin_amdgcn_permlane16_swap(t0.x, t1.x, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap(t0.y, t1.y, false, false);
  return make_uint4(lo[0], hi[0], lo[1], hi[1]);
}

constexpr int PD = 4;  // weight prefetch distance in k-steps (ring of PD fragment sets per wave)
constexpr int CS_FFMAX = 2048;  // static forms (chain_s_kernel, ffn_bwd_kernel<FULL>): widest FeedForward whose biases fit their LDS block
extern "C" int csship_ffn_static_max_ffn(void) { return CS_FFMAX; }
constexpr int CS_DUMP = 256;    // floats behind a parameter block where staging lanes without work write
#ifdef P_STAMPS  // diagnostic build: wave 0 of every workgroup stores s_memtime at phase boundaries -> g_csship_stamps[wg*32 + i]
#define P_T(i)                                                                                                  \
  {                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    if (g_stamps_dev != nullptr && threadIdx.x == 0) {                                                          \
      unsigned long long t_;                                                                                    \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                \
      g_stamps_dev[(size_t)blockIdx.x * 32 + (i)] = t_;                                                         \
    }                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  }
#define P_TQ(v)                                                                       \
  if (st0 == 1) {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");         \
    __builtin_amdgcn_sched_barrier(0);                                                \
  }
#else
#define P_T(i)
#define P_TQ(v)
#endif
#ifndef P_ABL
#define P_ABL 0  // diagnostic builds only: bit 0 no weight loads in the k-loop, bit 1 no MFMAs, bit 2 no LDS operand reads,
                 // bit 3 no qkv stores, bit 4 no x stores, bit 5 no hidden-bias loads (chain kernel)
#endif

// acc[t][rt] += W(tile0+t)[16 x 256] . panel^T[256 x 16 rows of tile rt]
//   w: this lane's pointer at block (tile0, k-step 0); block (t, ks) is at w[(t*tstride + ks)*64].
//   wq: weight fragments of k-steps 0..PD-1, prefetched by the caller; on return it holds k-steps
//   0..PD-1 of the NEXT stream (wn / nstride), so the global-load pipeline never drains.
//   hipcc otherwise sinks the prefetch loads down to their first use (and waits on them at once):
//   sched_barrier(0) pins "issue loads for k-step ks+PD" ahead of the LDS reads + MFMAs of ks.
template <int NT>
DEV void mma_pass(f32x4 (&acc)[NT][4], const unsigned char* panel, int n, int g, int nt, const bf16x8* w, int tstride,
                  const bf16x8* wn, int nstride, int nnt, bf16x8 (&wq)[PD][NT]) {
  // LDS operand fragments are double buffered by hand: the reads of k-step ks+1 are issued before the
  // MFMAs of k-step ks, so one wave alone keeps the matrix pipe fed while its reads are in flight.
  bf16x8 xa[2][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) xa[0][rt] = *reinterpret_cast<const bf16x8*>(panel + panel_off(rt * 16 + n, g));
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 wc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wc[t] = wq[ks % PD][t];
    if (P_ABL & 1) {
    } else if (ks + PD < KS) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) wq[ks % PD][t] = w[(size_t)(t * tstride + ks + PD) * 64];
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nnt) wq[ks % PD][t] = wn[(size_t)(t * nstride + ks + PD - KS) * 64];
    }
    if (ks + 1 < KS && !(P_ABL & 4)) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        xa[(ks + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(panel + panel_off(rt * 16 + n, (ks + 1) * 4 + g));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt) {
        if (P_ABL & 2) {
          asm volatile("" ::"v"(wc[t]));
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) asm volatile("" ::"v"(xa[ks & 1][rt]));
          continue;
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], xa[(P_ABL & 4) ? 0 : (ks & 1)][rt], acc[t][rt], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NT>
DEV void prefetch_first(bf16x8 (&wq)[PD][NT], const bf16x8* w, int tstride, int nt) {
#pragma unroll
  for (int k = 0; k < PD; ++k)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt) wq[k][t] = w[(size_t)(t * tstride + k) * 64];
  __builtin_amdgcn_sched_barrier(0);
}

// ---- panel prologues.  Thread (row = tid>>3, part = tid&7) owns 32 elements of one row: float4 pieces
// (i*8 + part), i = 0..7, so the 8 lanes of a row read one full 128-byte line per load instruction and
// the LayerNorm statistics need only 3 shuffle steps (xor 1, 2, 4).  Requires NW*64 == 8*BM threads.
template <bool LN>
DEV void load_panel_rows(unsigned char* panel, const float* x, int m0, int M, const float* gam, const float* bet,
                         int tid) {
  const int r = tid >> 3, part = tid & 7;
  const int row = min(m0 + r, M - 1);
  const float* xr = x + (size_t)row * DM;
  float4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (i * 8 + part) * 4);
  if constexpr (LN) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = sum8(s) * (1.f / DM);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 gm = *reinterpret_cast<const float4*>(gam + (i * 8 + part) * 4);
      const float4 bt = *reinterpret_cast<const float4*>(bet + (i * 8 + part) * 4);
      v[i].x = v[i].x * rstd * gm.x + bt.x; v[i].y = v[i].y * rstd * gm.y + bt.y;
      v[i].z = v[i].z * rstd * gm.z + bt.z; v[i].w = v[i].w * rstd * gm.w + bt.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = i * 8 + part;  // float4 index within the row = 8-byte bf16 piece index
    *reinterpret_cast<uint2*>(panel + panel_off(r, c4 >> 1) + (c4 & 1) * 8) =
        make_uint2(pack_bf2(v[i].x, v[i].y), pack_bf2(v[i].z, v[i].w));
  }
}
template <int NW>
DEV void load_panel_bf16(unsigned char* panel, const unsigned short* a, int m0, int M, int tid) {
  for (int i = tid; i < BM * 32; i += NW * 64) {
    const int r = i >> 5, c = i & 31;
    const int row = min(m0 + r, M - 1);
    *reinterpret_cast<uint4*>(panel + panel_off(r, c)) = *reinterpret_cast<const uint4*>(a + (size_t)row * DM + c * 8);
  }
}

// ---- staged epilogue.  The transposed accumulators are parked in an fp32 LDS tile Ys[64 rows][YS]
// (one 256-feature block per pass); after a barrier thread (row = tid>>3, part = tid&7) finishes 32
// elements of ONE row: residual, LayerNorm statistics (in-lane sums + 3 shuffles), conversion, and
// global stores in which the 8 lanes of a row cover whole 128-byte lines.
constexpr int YS = 260;                // row stride in floats: 1040 B, consecutive rows shift by one 16-B slot
constexpr int YS_BYTES = BM * YS * 4;  // 66 560 B

template <int NT, bool RELU>
DEV void park_tiles(float* ys, const f32x4 (&acc)[NT][4], int nt, int col0, int g, int n, const float4 (&bias)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nt) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        float4 v = make_float4(acc[t][rt][0] + bias[t].x, acc[t][rt][1] + bias[t].y, acc[t][rt][2] + bias[t].z,
                               acc[t][rt][3] + bias[t].w);
        if constexpr (RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(ys + (rt * 16 + n) * YS + col0 + t * 16 + 4 * g) = v;
      }
    }
}

DEV float4 load_bias4(const float* bias, int f, int N) {
  if (bias == nullptr) return make_float4(0.f, 0.f, 0.f, 0.f);
  if (f + 3 < N) return *reinterpret_cast<const float4*>(bias + f);
  return make_float4(f < N ? bias[f] : 0.f, f + 1 < N ? bias[f + 1] : 0.f, f + 2 < N ? bias[f + 2] : 0.f, 0.f);
}

// =============================================================================== fused FFN
struct FfnArgs {
  const float* x;
  float* out;
  const float *ln_g, *ln_b;
  const bf16x8* w1;  // packed [FF/16][8][64]
  const float* b1;
  const bf16x8* w2;  // packed [16][FF/32][64]
  const float* b2;
  const float *fin_g, *fin_b;
  int M, FF;
  float alpha;
  unsigned long long* stamps;
  csship_dropout_cfg drop_h, drop_o;  // training: dropout of the hidden activation [M, FF] and of the output [M, 256]
  float* pre_out;  // optional, with the final LayerNorm: the rows in front of it (the backward's saved input)
};

template <int NW, bool FINAL_LN>
__global__ __launch_bounds__(NW * 64) void ffn_kernel(FfnArgs a) {
  static_assert(NW * 64 == 8 * BM, "row phases assume 8 lanes per panel row");
  constexpr int NT = 16 / NW;  // 16-feature tiles per wave per 256-feature pass
  __shared__ __attribute__((aligned(16))) unsigned char Xs[PANEL_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Hraw[YS_BYTES];  // 2 hidden panels, later the Ys tile
  unsigned char* const Hs0 = Hraw;
  unsigned char* const Hs1 = Hraw + PANEL_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM;
  const int nchunk = a.FF / DM, ks2 = a.FF / 32;
  const Drop dh = drop_make(a.drop_h.seed, a.drop_h.stream, a.drop_h.p);
  const Drop dout = drop_make(a.drop_o.seed, a.drop_o.stream, a.drop_o.p);

  // weight streams of this wave: W1 chunk c -> hidden tiles c*16 + wave*NT + t (tile stride KS blocks);
  // W2 chunk c -> output tiles wave*NT + t, k-steps 8c.. (tile stride FF/32 blocks)
  const bf16x8* w1s = a.w1 + (size_t)(wave * NT) * KS * 64 + lane;
  const bf16x8* w2s = a.w2 + (size_t)(wave * NT) * ks2 * 64 + lane;
  bf16x8 wq[PD][NT];
  stamp(a.stamps, 0);
  prefetch_first<NT>(wq, w1s + (size_t)(blockIdx.x % nchunk) * 16 * KS * 64, KS, NT);  // in flight during the LN prologue

  load_panel_rows<true>(Xs, a.x, m0, a.M, a.ln_g, a.ln_b, tid);
  __syncthreads();
  stamp(a.stamps, 1);

  f32x4 o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // every workgroup walks the hidden chunks in a different rotation, so the 200 panels do not all pull
  // the same weight lines through the same L2 channel at the same moment (the sum over chunks commutes)
  const int rot = blockIdx.x % nchunk;
  for (int ci = 0; ci < nchunk; ++ci) {
    const int c = (ci + rot) % nchunk;
    f32x4 h[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {  // the bias is the accumulators' initial value
      const float4 bb = *reinterpret_cast<const float4*>(a.b1 + c * DM + (wave * NT + t) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    const bf16x8* w1c = w1s + (size_t)c * 16 * KS * 64;
    const bf16x8* w2c = w2s + (size_t)c * KS * 64;
    mma_pass<NT>(h, Xs, n, g, NT, w1c, KS, w2c, ks2, NT, wq);
    // hidden = relu(h) -> bf16 -> Hs[c&1][row][wave's 16*NT columns]; ReLU on the packed pairs
    unsigned char* hs = (ci & 1) ? Hs1 : Hs0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int hc = (wave * NT + t) * 16 + 4 * g;  // column within the chunk
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        uint2 pk;
        if (dh.thresh != 0u) {  // training: relu, then the hidden activation's dropout (conformer.py:317), element (row, c*256 + hc)
          pk = drop_relu_pack4(dh, (unsigned)((m0 + rt * 16 + n) * a.FF + c * DM + hc), h[t][rt][0], h[t][rt][1], h[t][rt][2], h[t][rt][3]);
        } else {
          pk = make_uint2(relu_bf2(pack_bf2(h[t][rt][0], h[t][rt][1])), relu_bf2(pack_bf2(h[t][rt][2], h[t][rt][3])));
        }
        *reinterpret_cast<uint2*>(hs + panel_off(rt * 16 + n, hc >> 3) + (hc & 7) * 2) = pk;
      }
    }
    __syncthreads();
    const bf16x8* w1n = w1s + (size_t)((c + 1) % nchunk) * 16 * KS * 64;  // after the last chunk: harmless re-read
    mma_pass<NT>(o, hs, n, g, NT, w2c, ks2, w1n, KS, NT, wq);
  }
  stamp(a.stamps, 2);

  // ---- epilogue: out = x + alpha * (o + b2)  [-> LayerNorm], row-wise through the Ys tile
  const int r = tid >> 3, part = tid & 7;
  const int row = m0 + r, rowc = min(row, a.M - 1);
  float4 xr[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) xr[i] = *reinterpret_cast<const float4*>(a.x + (size_t)rowc * DM + (i * 8 + part) * 4);
  float4 b2v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) b2v[t] = *reinterpret_cast<const float4*>(a.b2 + (wave * NT + t) * 16 + 4 * g);
  __syncthreads();  // every wave is done reading the hidden panels
  float* ys = reinterpret_cast<float*>(Hraw);
  park_tiles<NT, false>(ys, o, NT, wave * NT * 16, g, n, b2v);
  __syncthreads();
  stamp(a.stamps, 3);
  float4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // (training: the output's dropout, conformer.py:319, element (row, column) of [M, 256])
    const float4 y = drop_apply4(dout, (unsigned)((m0 + r) * DM + (i * 8 + part) * 4),
                                 *reinterpret_cast<const float4*>(ys + r * YS + (i * 8 + part) * 4));
    v[i] = make_float4(xr[i].x + a.alpha * y.x, xr[i].y + a.alpha * y.y, xr[i].z + a.alpha * y.z, xr[i].w + a.alpha * y.w);
  }
  if constexpr (FINAL_LN) {
    if (a.pre_out != nullptr && row < a.M) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(a.pre_out + (size_t)row * DM + (i * 8 + part) * 4) = v[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = sum8(s) * (1.f / DM);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 gm = *reinterpret_cast<const float4*>(a.fin_g + (i * 8 + part) * 4);
      const float4 bt = *reinterpret_cast<const float4*>(a.fin_b + (i * 8 + part) * 4);
      v[i].x = v[i].x * rstd * gm.x + bt.x; v[i].y = v[i].y * rstd * gm.y + bt.y;
      v[i].z = v[i].z * rstd * gm.z + bt.z; v[i].w = v[i].w * rstd * gm.w + bt.w;
    }
  }
  stamp(a.stamps, 4);
  if (row < a.M) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(a.out + (size_t)row * DM + (i * 8 + part) * 4) = v[i];
  }
  stamp(a.stamps, 5);
}

// =============================================================================== fused FFN backward (dX side)
// out = x + alpha * (relu(LN(x) W1^T + b1) W2^T + b2)  (ffn_kernel above).  Given dy = d(out), ONE kernel per row panel
//   a  = LN(x)                                  -> Xs (bf16 panel) and global a_bf   (operand of dW1 = dh^T a)
//   per hidden chunk c (three MFMA passes, weights streamed like the forward):
//     h  = relu(a W1_c^T + b1_c)                -> global h_bf (operand of dW2 = dy^T h); its sign = the ReLU mask
//     dh = alpha * (dy W2[:, c]) o [h > 0]      -> LDS hidden panel (bf16) and global dh_bf (operand of dW1, db1)
//     da += dh W1_c
//   dx = dy + LayerNorm-backward(da) in the staged row epilogue; dgamma / dbeta partials per workgroup.
// Replaces, per FeedForward: LayerNorm + tiled GEMM (h recompute) + two tiled backward GEMMs + the LayerNorm-backward
// kernel.  The weight-gradient GEMMs (reductions over the B*T rows) stay in csship_bgemm.
struct FfnBwdArgs {
  const float* x;
  const float* dy;
  const float *ln_g, *ln_b;
  const bf16x8* w1;   // packed from W1  [FF, 256]  (as the forward)
  const float* b1;
  const bf16x8* w2t;  // packed from W2^T [FF, 256]
  const bf16x8* w1t;  // packed from W1^T [256, FF]
  unsigned short* a_bf;   // [M, 256]
  unsigned short* h_bf;   // [M, FF]
  unsigned short* dh_bf;  // [M, FF]
  float* dx;              // [M, 256]
  float* part;            // [nwg][2][256]  dgamma | dbeta partial sums
  int M, FF;
  float alpha;
  csship_dropout_cfg drop_h, drop_o;  // the forward's dropout sites (hidden [M, FF], output [M, 256])
  unsigned short* dyd_bf;             // optional [M, 256]: alpha * dy o output-dropout mask / (1 - p): operand of dW2 / db2
  // LN2 (optional): x = LayerNorm2(x2) is itself the output of a LayerNorm (the previous block's final one,
  // conformer.py:360): the epilogue carries dx on through it - dx <- LayerNorm2-backward(dx) - and leaves that
  // LayerNorm's dgamma / dbeta partials; its own launch (csship_layernorm_bwd) read x2 and dx and wrote dx once more
  const float* ln2_x;                 // [M, 256] the LayerNorm's input rows
  const float* ln2_g;
  float* part2;                       // [nwg][2][256]
  // CS (optional): dx is the output gradient of the conv module in front of this FeedForward (the block's feed_forward_out,
  // conformer.py:357-358): the statistics pass of that module's backward (conv_bn_bwd_stats_kernel, convmod.hip) rides in the
  // epilogue - per channel S1 = sum dzh, S2 = sum dzh * zn, and the scalars d pw2_w, d pw2_b, as partials per workgroup
  const float* cs_z;                  // [M, 256] the module's pre-BatchNorm activations
  const float* cs_stats;              // [4][256] scale | shift | mean | variance
  const float* cs_pw2_w;
  csship_dropout_cfg cs_drop;         // the module's output dropout site
  float* cs_part;                     // [nwg][2 * 256 + 2]
};

// FULL (M % 64 == 0, d_ffn <= 2048: what the encoder runs): no row predicates on loads or stores, and b1 / gamma / beta
// come from an LDS copy staged once - every global memory instruction of the chunk loop is unconditional, so the
// s_waitcnt counts of the weight ring are exact (a predicated store in the loop made hipcc assume the path without it
// and wait for younger loads than the ones it needs; round 5, see chain_s_kernel).
// LDS of the kernel: XD = a = LN(x) | alpha * dy (two bf16 panels; later 64 KB of column sums), Graw = 2 dh panels (later the Ys
// tile), Pb = FULL: b1 | gamma | beta | gamma2 | dump.  DYREG: dy arrives in registers (this thread's pieces of row tid >> 3, the
// row mapping of the prologue) from the stage in front of it in the same kernel (qkv_ffn_bwd_kernel) instead of from memory.
constexpr int FB_XD_BYTES = 2 * PANEL_BYTES, FB_GRAW_BYTES = 2 * PANEL_BYTES + 1024, FB_PB_FLOATS = CS_FFMAX + 3 * DM + CS_DUMP;
// KEEPOUT (with LN2): the final dx - the gradient through the second LayerNorm - goes back into dyio instead of memory, for a
// third stage behind this one in the same kernel (qkv_ffn2_bwd_kernel).
template <int NW, bool FULL, bool LN2, bool DYREG, bool CS = false, bool KEEPOUT = false>
DEV void ffn_bwd_body(const FfnBwdArgs& a, unsigned char* const XD, unsigned char* const Graw, float* const Pb,
                      float4 (&dyio)[8]) {
  static_assert(!KEEPOUT || LN2, "the hand-over sits in the second LayerNorm's epilogue");
  static_assert(FULL || !LN2, "the fused second LayerNorm backward is built for whole panels");
  static_assert(!CS || (FULL && !LN2), "the conv module's statistics ride behind a whole-panel FeedForward without a second LayerNorm");
  static_assert(FULL || !DYREG, "the chained form is built for whole panels");
  static_assert(NW * 64 == 8 * BM, "row phases assume 8 lanes per panel row");
  constexpr int NT = 16 / NW;
  constexpr bool KEEP_DY = FULL;
  unsigned char* const Xs = XD;
  unsigned char* const Ds = XD + PANEL_BYTES;
  static_assert(2 * PANEL_BYTES + 1024 >= YS_BYTES, "Ys tile must fit the dh panels");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM;
  const int nchunk = a.FF / DM, ks2 = a.FF / 32;
  const bf16x8* w1s = a.w1 + (size_t)(wave * NT) * KS * 64 + lane;
  const bf16x8* w2ts = a.w2t + (size_t)(wave * NT) * KS * 64 + lane;
  const bf16x8* w1ts = a.w1t + (size_t)(wave * NT) * ks2 * 64 + lane;
  const int rot = blockIdx.x % nchunk;
  const Drop dh = drop_make(a.drop_h.seed, a.drop_h.stream, a.drop_h.p);
  const Drop dout = drop_make(a.drop_o.seed, a.drop_o.stream, a.drop_o.p);
  bf16x8 wq[PD][NT];
  const float* b1p = a.b1;
  const float* lgp = a.ln_g;
  const float* lbp = a.ln_b;
  if constexpr (FULL) {  // parameters -> LDS: float4 j of b1 (FF/4) | gamma (64) | beta (64), two per thread, branch-free
    const int q1 = a.FF >> 2;
    const float* const dummy = a.x + (size_t)m0 * DM + lane * 4;
    float4 pvv[2];
    int pdd[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int j = tid + 512 * k;
      // CS: the conv module's scale | shift | mean | variance (256 float4) behind the dump block (FF / 4 + 192 + 256 <= 1024 slots)
      const bool csj = CS && j >= q1 + 192 && j < q1 + 448;
      const float* const base = j < q1 ? a.b1 : j < q1 + 64 ? a.ln_g : j < q1 + 128 ? a.ln_b : (LN2 && j < q1 + 192) ? a.ln2_g
                                : csj ? a.cs_stats : nullptr;
      const int off = j < q1 ? j : j < q1 + 64 ? j - q1 : j < q1 + 128 ? j - q1 - 64 : csj ? j - q1 - 192 : j - q1 - 128;
      pdd[k] = csj ? FB_PB_FLOATS + (j - q1 - 192) * 4
                   : base != nullptr ? (j < q1 ? j * 4 : CS_FFMAX + (j - q1) * 4) : CS_FFMAX + 3 * DM + lane * 4;
      pvv[k] = *reinterpret_cast<const float4*>(base != nullptr ? base + off * 4 : dummy);
    }
    *reinterpret_cast<float4*>(Pb + pdd[0]) = pvv[0];
    *reinterpret_cast<float4*>(Pb + pdd[1]) = pvv[1];
    b1p = Pb; lgp = Pb + CS_FFMAX; lbp = Pb + CS_FFMAX + DM;
    __syncthreads();
  }
  prefetch_first<NT>(wq, w1s + (size_t)rot * 16 * KS * 64, KS, NT);

  // ---- prologue: the two bf16 panels; LN(x) also goes to global for the weight-gradient GEMM
  float4 dyk[KEEP_DY ? 8 : 1];  // KEEP_DY: this thread's dy row pieces stay in registers for the dx epilogue (no second read)
  {
    const int r = tid >> 3, part = tid & 7;
    const int row = m0 + r, rowc = FULL ? row : min(row, a.M - 1);
    float4 v[8], d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = *reinterpret_cast<const float4*>(a.x + (size_t)rowc * DM + (i * 8 + part) * 4);
      if constexpr (DYREG) d[i] = dyio[i];
      else d[i] = *reinterpret_cast<const float4*>(a.dy + (size_t)rowc * DM + (i * 8 + part) * 4);
      if constexpr (KEEP_DY) dyk[i] = d[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = sum8(s) * (1.f / DM);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = i * 8 + part;
      const float4 gm = *reinterpret_cast<const float4*>(lgp + c4 * 4);
      const float4 bt = *reinterpret_cast<const float4*>(lbp + c4 * 4);
      const uint2 av = make_uint2(pack_bf2(v[i].x * rstd * gm.x + bt.x, v[i].y * rstd * gm.y + bt.y),
                                  pack_bf2(v[i].z * rstd * gm.z + bt.z, v[i].w * rstd * gm.w + bt.w));
      *reinterpret_cast<uint2*>(Xs + panel_off(r, c4 >> 1) + (c4 & 1) * 8) = av;
      if (FULL || row < a.M) *reinterpret_cast<uint2*>(a.a_bf + (size_t)row * DM + c4 * 4) = av;
      // gradient of the FeedForward's (dropped) output: alpha * dy o mask / (1 - p)
      const float4 dd = drop_apply4(dout, (unsigned)((m0 + r) * DM + c4 * 4),
                                    make_float4(a.alpha * d[i].x, a.alpha * d[i].y, a.alpha * d[i].z, a.alpha * d[i].w));
      const uint2 dv = make_uint2(pack_bf2(dd.x, dd.y), pack_bf2(dd.z, dd.w));
      *reinterpret_cast<uint2*>(Ds + panel_off(r, c4 >> 1) + (c4 & 1) * 8) = dv;
      if (a.dyd_bf != nullptr && (FULL || row < a.M)) *reinterpret_cast<uint2*>(a.dyd_bf + (size_t)row * DM + c4 * 4) = dv;
    }
  }
  __syncthreads();

  f32x4 o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int ci = 0; ci < nchunk; ++ci) {
    const int c = (ci + rot) % nchunk;
    const bf16x8* w1c = w1s + (size_t)c * 16 * KS * 64;
    const bf16x8* w2tc = w2ts + (size_t)c * 16 * KS * 64;
    const bf16x8* w1tc = w1ts + (size_t)c * KS * 64;
    const bf16x8* w1n = w1s + (size_t)((c + 1) % nchunk) * 16 * KS * 64;
    f32x4 h[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {  // the bias is the accumulators' initial value, as in the forward
      const float4 bb = *reinterpret_cast<const float4*>(b1p + c * DM + (wave * NT + t) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    // pass A: h = dropout(relu(a W1_c^T + b1_c)); keep the (ReLU and dropout) mask, store h
    mma_pass<NT>(h, Xs, n, g, NT, w1c, KS, w2tc, KS, NT, wq);
    unsigned mask = 0u;
    uint2 hpk[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int fcol = c * DM + (wave * NT + t) * 16 + 4 * g;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = m0 + rt * 16 + n;
        // dropout(relu(.)) on the packed pairs (drop_relu_pack4; thresh 0 keeps everything); the ReLU-and-dropout mask
        // of the backward is "the stored bf16 value is not zero"
        const uint2 pk = drop_relu_pack4(dh, (unsigned)(row * a.FF + fcol), h[t][rt][0], h[t][rt][1], h[t][rt][2], h[t][rt][3]);
        mask |= (((pk.x & 0xffffu) ? 1u : 0u) | ((pk.x >> 16) ? 2u : 0u) | ((pk.y & 0xffffu) ? 4u : 0u) | ((pk.y >> 16) ? 8u : 0u)) << ((t * 4 + rt) * 4);
        hpk[t][rt] = pk;
        h[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if constexpr (NT == 2) {   // the wave's two tiles -> one 16-byte store per lane and row tile
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const uint4 pk = pair_tiles_bf16(hpk[0][rt], hpk[1][rt]);
        const int row = m0 + rt * 16 + n;
        if (FULL || row < a.M) *reinterpret_cast<uint4*>(a.h_bf + (size_t)row * a.FF + c * DM + (wave * NT + (g & 1)) * 16 + (g >> 1) * 8) = pk;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = m0 + rt * 16 + n;
          if (FULL || row < a.M) *reinterpret_cast<uint2*>(a.h_bf + (size_t)row * a.FF + c * DM + (wave * NT + t) * 16 + 4 * g) = hpk[t][rt];
        }
    }
    // pass B: dh = (alpha dy) W2[:, c] masked  -> LDS panel + global
    mma_pass<NT>(h, Ds, n, g, NT, w2tc, KS, w1tc, ks2, NT, wq);
    unsigned char* gs = Graw + (ci & 1) * PANEL_BYTES;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int hc = (wave * NT + t) * 16 + 4 * g;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const unsigned mb = mask >> ((t * 4 + rt) * 4);  // survivors of ReLU AND dropout; they carry the 1 / (1 - p)
        const float hs_ = dh.scale;
        const uint2 pk = make_uint2(pack_bf2((mb & 1u) ? h[t][rt][0] * hs_ : 0.f, (mb & 2u) ? h[t][rt][1] * hs_ : 0.f),
                                    pack_bf2((mb & 4u) ? h[t][rt][2] * hs_ : 0.f, (mb & 8u) ? h[t][rt][3] * hs_ : 0.f));
        *reinterpret_cast<uint2*>(gs + panel_off(rt * 16 + n, hc >> 3) + (hc & 7) * 2) = pk;
        hpk[t][rt] = pk;
      }
    }
    if constexpr (NT == 2) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const uint4 pk = pair_tiles_bf16(hpk[0][rt], hpk[1][rt]);
        const int row = m0 + rt * 16 + n;
        if (FULL || row < a.M) *reinterpret_cast<uint4*>(a.dh_bf + (size_t)row * a.FF + c * DM + (wave * NT + (g & 1)) * 16 + (g >> 1) * 8) = pk;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = m0 + rt * 16 + n;
          if (FULL || row < a.M) *reinterpret_cast<uint2*>(a.dh_bf + (size_t)row * a.FF + c * DM + (wave * NT + t) * 16 + 4 * g) = hpk[t][rt];
        }
    }
    __syncthreads();
    // pass C: da += dh W1_c
    mma_pass<NT>(o, gs, n, g, NT, w1tc, ks2, w1n, KS, NT, wq);
  }

  // ---- epilogue: da rows through the Ys tile; LayerNorm backward; gamma / beta partials
  __syncthreads();  // every wave is done reading the dh panels
  float* ys = reinterpret_cast<float*>(Graw);
  {
    float4 zero[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) zero[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    park_tiles<NT, false>(ys, o, NT, wave * NT * 16, g, n, zero);
  }
  __syncthreads();
  const int r = tid >> 3, part = tid & 7;
  const int row = m0 + r, rowc = FULL ? row : min(row, a.M - 1);
  const bool rvalid = FULL || row < a.M;
  float4 v[8], da[8], dyv[8];
  float4 x2[LN2 ? 8 : 1];
  float4 zr[CS ? 8 : 1];   // CS: this thread's pieces of the conv module's z row, requested with the x row
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = *reinterpret_cast<const float4*>(a.x + (size_t)rowc * DM + (i * 8 + part) * 4);
    if constexpr (CS) zr[i] = *reinterpret_cast<const float4*>(a.cs_z + (size_t)rowc * DM + (i * 8 + part) * 4);
    if constexpr (LN2) x2[i] = *reinterpret_cast<const float4*>(a.ln2_x + (size_t)rowc * DM + (i * 8 + part) * 4);
    // dy is needed by the dx rows below: requested here, all eight at once (read next to its use, behind stores it might
    // alias, every piece was a load - s_waitcnt vmcnt(0) - store round trip of its own: eight L2 latencies in a row)
    if constexpr (KEEP_DY) dyv[i] = dyk[i];
    else dyv[i] = *reinterpret_cast<const float4*>(a.dy + (size_t)rowc * DM + (i * 8 + part) * 4);
    da[i] = *reinterpret_cast<const float4*>(ys + r * YS + (i * 8 + part) * 4);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = sum8(s) * (1.f / DM);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
  float s1 = 0.f, s2 = 0.f;
  float4 gg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 gm = *reinterpret_cast<const float4*>(lgp + (i * 8 + part) * 4);
    v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // xhat
    if (!rvalid) da[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    gg[i] = make_float4(da[i].x * gm.x, da[i].y * gm.y, da[i].z * gm.z, da[i].w * gm.w);
    s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
    s2 += (gg[i].x * v[i].x + gg[i].y * v[i].y) + (gg[i].z * v[i].z + gg[i].w * v[i].w);
  }
  s1 = sum8(s1) * (1.f / DM);
  s2 = sum8(s2) * (1.f / DM);
  __syncthreads();  // all da values are in registers: the Ys tile (and Xs / Ds) can be reused for the column sums
  float* cg = reinterpret_cast<float*>(Graw);  // [64][260] da * xhat
  float* cb = reinterpret_cast<float*>(XD);    // [64][256] da       (both panels: 64 KB)
  float4 t1[CS ? 8 : 1], t2[CS ? 8 : 1];   // CS: dzh and dzh * zn of this thread's pieces
  float cs_dw3 = 0.f, cs_db3 = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int col = (i * 8 + part) * 4;
    if (rvalid) {
      const float4 d = dyv[i];
      float4 o4;
      o4.x = d.x + rstd * (gg[i].x - s1 - v[i].x * s2);
      o4.y = d.y + rstd * (gg[i].y - s1 - v[i].y * s2);
      o4.z = d.z + rstd * (gg[i].z - s1 - v[i].z * s2);
      o4.w = d.w + rstd * (gg[i].w - s1 - v[i].w * s2);
      if constexpr (LN2) dyv[i] = o4;   // goes on through the second LayerNorm below
      else *reinterpret_cast<float4*>(a.dx + (size_t)row * DM + col) = o4;
      if constexpr (CS) {   // conv_bn_bwd_stats_kernel's terms on the dx row just formed (its dy), same element order per row
        const Drop cdp = drop_make(a.cs_drop.seed, a.cs_drop.stream, a.cs_drop.p);
        const float* const cst = Pb + FB_PB_FLOATS;   // staged at the kernel's start
        const float4 sc4 = *reinterpret_cast<const float4*>(cst + col), sh4 = *reinterpret_cast<const float4*>(cst + DM + col);
        const float4 mu4 = *reinterpret_cast<const float4*>(cst + 2 * DM + col), va4 = *reinterpret_cast<const float4*>(cst + 3 * DM + col);
        const float w3 = a.cs_pw2_w[0];
        const float4 d4 = drop_apply4(cdp, (unsigned)(row * DM + col), o4);
        const float dd[4] = {d4.x, d4.y, d4.z, d4.w}, zz[4] = {zr[i].x, zr[i].y, zr[i].z, zr[i].w};
        const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w}, muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
        const float rsv[4] = {rsqrtf(va4.x + 1e-5f), rsqrtf(va4.y + 1e-5f), rsqrtf(va4.z + 1e-5f), rsqrtf(va4.w + 1e-5f)};
        float u1[4], u2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float zh = zz[q] * scv[q] + shv[q];
          const float dzh = zh > 0.f ? dd[q] * w3 : 0.f;
          u1[q] = dzh;
          u2[q] = dzh * (zz[q] - muv[q]) * rsv[q];
          cs_dw3 += dd[q] * fmaxf(zh, 0.f);
          cs_db3 += dd[q];
        }
        t1[i] = make_float4(u1[0], u1[1], u1[2], u1[3]);
        t2[i] = make_float4(u2[0], u2[1], u2[2], u2[3]);
      }
    }
    *reinterpret_cast<float4*>(cg + r * YS + col) = make_float4(da[i].x * v[i].x, da[i].y * v[i].y, da[i].z * v[i].z, da[i].w * v[i].w);
    *reinterpret_cast<float4*>(cb + r * DM + col) = da[i];
  }
  __syncthreads();
  {
    const int col = tid & 255, which = tid >> 8;  // 512 threads: 256 columns x {dgamma, dbeta}
    float acc = 0.f;
    if (which == 0) {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) acc += cg[rr * YS + col];
    } else {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) acc += cb[rr * DM + col];
    }
    a.part[(size_t)blockIdx.x * 2 * DM + which * DM + col] = acc;
  }
  if constexpr (CS) {
    // the conv module's S1 | S2 column sums through the two staging tiles once more; its two scalars through the dump block
    cs_dw3 = wave_sum(cs_dw3);
    cs_db3 = wave_sum(cs_db3);
    __syncthreads();  // the first column sums have been read
    float* const sred = Pb + CS_FFMAX + 3 * DM;   // [NW][2] in the dump block (nobody stages parameters any more)
    if (lane == 0) { sred[wave * 2] = cs_dw3; sred[wave * 2 + 1] = cs_db3; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = (i * 8 + part) * 4;
      *reinterpret_cast<float4*>(cg + r * YS + col) = t1[i];
      *reinterpret_cast<float4*>(cb + r * DM + col) = t2[i];
    }
    __syncthreads();
    const int col = tid & 255, which = tid >> 8;
    float acc = 0.f;
    if (which == 0) {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) acc += cg[rr * YS + col];
    } else {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) acc += cb[rr * DM + col];
    }
    float* const out = a.cs_part + (size_t)blockIdx.x * (2 * DM + 2);
    out[which * DM + col] = acc;
    if (tid < 2) {
      float sc = 0.f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) sc += sred[wv * 2 + tid];
      out[2 * DM + tid] = sc;
    }
  }
  if constexpr (LN2) {
    // dx (in dyv) is the gradient of LayerNorm2's output: statistics of its input row, the same three-term formula,
    // and the column sums of dx o xhat2 / dx through the two staging tiles once more
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += (x2[i].x + x2[i].y) + (x2[i].z + x2[i].w);
    const float mean2 = sum8(t) * (1.f / DM);
    float tt = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x2[i].x -= mean2; x2[i].y -= mean2; x2[i].z -= mean2; x2[i].w -= mean2;
      tt += (x2[i].x * x2[i].x + x2[i].y * x2[i].y) + (x2[i].z * x2[i].z + x2[i].w * x2[i].w);
    }
    const float rstd2 = rsqrtf(sum8(tt) * (1.f / DM) + 1e-5f);
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 gm = *reinterpret_cast<const float4*>(Pb + CS_FFMAX + 2 * DM + (i * 8 + part) * 4);
      x2[i].x *= rstd2; x2[i].y *= rstd2; x2[i].z *= rstd2; x2[i].w *= rstd2;  // xhat2
      gg[i] = make_float4(dyv[i].x * gm.x, dyv[i].y * gm.y, dyv[i].z * gm.z, dyv[i].w * gm.w);
      t1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
      t2 += (gg[i].x * x2[i].x + gg[i].y * x2[i].y) + (gg[i].z * x2[i].z + gg[i].w * x2[i].w);
    }
    t1 = sum8(t1) * (1.f / DM);
    t2 = sum8(t2) * (1.f / DM);
    __syncthreads();  // the first column sums have been read
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = (i * 8 + part) * 4;
      float4 o4;
      o4.x = rstd2 * (gg[i].x - t1 - x2[i].x * t2);
      o4.y = rstd2 * (gg[i].y - t1 - x2[i].y * t2);
      o4.z = rstd2 * (gg[i].z - t1 - x2[i].z * t2);
      o4.w = rstd2 * (gg[i].w - t1 - x2[i].w * t2);
      if constexpr (KEEPOUT) dyio[i] = o4;
      else *reinterpret_cast<float4*>(a.dx + (size_t)row * DM + col) = o4;
      *reinterpret_cast<float4*>(cg + r * YS + col) = make_float4(dyv[i].x * x2[i].x, dyv[i].y * x2[i].y, dyv[i].z * x2[i].z, dyv[i].w * x2[i].w);
      *reinterpret_cast<float4*>(cb + r * DM + col) = dyv[i];
    }
    __syncthreads();
    const int col = tid & 255, which = tid >> 8;
    float acc = 0.f;
    if (which == 0) {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) acc += cg[rr * YS + col];
    } else {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) acc += cb[rr * DM + col];
    }
    a.part2[(size_t)blockIdx.x * 2 * DM + which * DM + col] = acc;
  }
}

template <int NW, bool FULL, bool LN2 = false, bool CS = false>
__global__ __launch_bounds__(NW * 64) void ffn_bwd_kernel(FfnBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char XD[FB_XD_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Graw[FB_GRAW_BYTES];
  __shared__ __attribute__((aligned(16))) float Pb[FULL ? FB_PB_FLOATS + (CS ? 4 * DM : 0) : 4];
  float4 none[8];   // (never read: DYREG = false)
  ffn_bwd_body<NW, FULL, LN2, false, CS>(a, XD, Graw, Pb, none);
}

// =============================================================================== panel linear
struct LinArgs {
  const void* a;
  const float *ln_g, *ln_b;
  const bf16x8* w;  // packed [ceil(N/16)][8][64]
  const float* bias;
  const float* res;
  void* out;
  int M, N;
  int64_t ldo, ldr;
  float alpha;
  unsigned long long* stamps;
  // optional conv-module front half fused into the residual epilogue (N == 256):
  //   g = glu(LayerNorm_conv(out_row))   (css/models/conformer.py:291-293)
  const float *cln_g, *cln_b, *pw1_w, *pw1_b;
  float* g_out;
  // optional dropout of the projection's output in front of the residual add (training: x + D(a W^T + b),
  // conformer.py:254), f32 output rows of N columns; element index row * N + column as csship_dropout_add
  const unsigned* dseed;
  unsigned dstream;
  float dp;
};

template <int NW, int PRO, int EPI>
__global__ __launch_bounds__(NW * 64) void panel_linear_kernel(LinArgs a) {
  static_assert(NW * 64 == 8 * BM, "row phases assume 8 lanes per panel row");
  constexpr int NT = 16 / NW;
  constexpr bool RELU = (EPI & CSSHIP_EPI_RELU) != 0, RES = (EPI & CSSHIP_EPI_RESIDUAL) != 0,
                 OBF = (EPI & CSSHIP_EPI_OUT_BF16) != 0;
  __shared__ __attribute__((aligned(16))) unsigned char Xs[PANEL_BYTES];
  __shared__ __attribute__((aligned(16))) float Ys[BM * YS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM;
  const int ntiles = (a.N + 15) / 16;
  const int npass = (ntiles + 15) / 16;  // each pass: the workgroup covers 256 consecutive features

  bf16x8 wq[PD][NT];
  const bf16x8* wbase = a.w + lane;
  stamp(a.stamps, 0);
  prefetch_first<NT>(wq, wbase + (size_t)(wave * NT) * KS * 64, KS, max(0, min(NT, ntiles - wave * NT)));
  if constexpr (PRO == CSSHIP_PRO_LN) load_panel_rows<true>(Xs, reinterpret_cast<const float*>(a.a), m0, a.M, a.ln_g, a.ln_b, tid);
  else if constexpr (PRO == CSSHIP_PRO_F32) load_panel_rows<false>(Xs, reinterpret_cast<const float*>(a.a), m0, a.M, nullptr, nullptr, tid);
  else load_panel_bf16<NW>(Xs, reinterpret_cast<const unsigned short*>(a.a), m0, a.M, tid);
  __syncthreads();
  stamp(a.stamps, 1);

  const int r = tid >> 3, part = tid & 7;
  const int row = m0 + r;
  const bool vec_ok = OBF ? ((a.ldo & 7) == 0) : ((a.ldo & 3) == 0);
  Drop dd{0u, 0u, 1.f};
  if constexpr (RES && !OBF) dd = drop_make(a.dseed, a.dstream, a.dp);

  for (int p = 0; p < npass; ++p) {
    const int tile0 = p * 16 + wave * NT;
    const int nt = max(0, min(NT, ntiles - tile0));
    const int nnt = max(0, min(NT, ntiles - (tile0 + 16)));
    float4 bb[NT];
    f32x4 acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      bb[t] = load_bias4(a.bias, (tile0 + t) * 16 + 4 * g, a.N);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const bf16x8* wc = wbase + (size_t)tile0 * KS * 64;
    mma_pass<NT>(acc, Xs, n, g, nt, wc, KS, wc + (size_t)16 * KS * 64, KS, nnt, wq);
    if (p == 0) stamp(a.stamps, 2);
    park_tiles<NT, RELU>(Ys, acc, nt, wave * NT * 16, g, n, bb);
    __syncthreads();
    const int f0 = p * 256, ncols = min(256, a.N - f0);
    if (vec_ok) {  // rows are 16-byte aligned: vector stores, scalar tail for ncols % 8 != 0
      if (row < a.M) {
        if constexpr (OBF) {
          unsigned short* o = reinterpret_cast<unsigned short*>(a.out) + (size_t)row * a.ldo + f0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int c = (i * 8 + part) * 8;
            if (c + 7 < ncols) {
              float4 y0 = *reinterpret_cast<const float4*>(Ys + r * YS + c);
              float4 y1 = *reinterpret_cast<const float4*>(Ys + r * YS + c + 4);
              if constexpr (RES) {
                const float4 x0 = *reinterpret_cast<const float4*>(a.res + (size_t)row * a.ldr + f0 + c);
                const float4 x1 = *reinterpret_cast<const float4*>(a.res + (size_t)row * a.ldr + f0 + c + 4);
                y0 = make_float4(x0.x + a.alpha * y0.x, x0.y + a.alpha * y0.y, x0.z + a.alpha * y0.z, x0.w + a.alpha * y0.w);
                y1 = make_float4(x1.x + a.alpha * y1.x, x1.y + a.alpha * y1.y, x1.z + a.alpha * y1.z, x1.w + a.alpha * y1.w);
              }
              *reinterpret_cast<uint4*>(o + c) = make_uint4(pack_bf2(y0.x, y0.y), pack_bf2(y0.z, y0.w), pack_bf2(y1.x, y1.y), pack_bf2(y1.z, y1.w));
            } else {
              for (int e = c; e < min(c + 8, ncols); ++e) {
                float y = Ys[r * YS + e];
                if constexpr (RES) y = a.res[(size_t)row * a.ldr + f0 + e] + a.alpha * y;
                o[e] = f2bf(y);
              }
            }
          }
        } else {
          float* o = reinterpret_cast<float*>(a.out) + (size_t)row * a.ldo + f0;
          float4 vrow[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int c = (i * 8 + part) * 4;
            vrow[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c + 3 < ncols) {
              float4 y = *reinterpret_cast<const float4*>(Ys + r * YS + c);
              if constexpr (RES) {
                const float4 x0 = *reinterpret_cast<const float4*>(a.res + (size_t)row * a.ldr + f0 + c);
                y = drop_apply4(dd, (unsigned)((size_t)row * a.N + f0 + c), y);
                y = make_float4(x0.x + a.alpha * y.x, x0.y + a.alpha * y.y, x0.z + a.alpha * y.z, x0.w + a.alpha * y.w);
              }
              *reinterpret_cast<float4*>(o + c) = y;
              vrow[i] = y;
            } else {
              for (int e = c; e < min(c + 4, ncols); ++e) {
                float y = Ys[r * YS + e];
                if constexpr (RES) y = a.res[(size_t)row * a.ldr + f0 + e] + a.alpha * y;
                o[e] = y;
              }
            }
          }
          if constexpr (RES) {
            if (a.g_out != nullptr) {  // N == 256: vrow is the whole new residual row
              float s = 0.f;
#pragma unroll
              for (int i = 0; i < 8; ++i) s += (vrow[i].x + vrow[i].y) + (vrow[i].z + vrow[i].w);
              const float mean = sum8(s) * (1.f / DM);
              float ss = 0.f;
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                vrow[i].x -= mean; vrow[i].y -= mean; vrow[i].z -= mean; vrow[i].w -= mean;
                ss += (vrow[i].x * vrow[i].x + vrow[i].y * vrow[i].y) + (vrow[i].z * vrow[i].z + vrow[i].w * vrow[i].w);
              }
              const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
              const float w0 = a.pw1_w[0], w1 = a.pw1_w[1], b0 = a.pw1_b[0], b1 = a.pw1_b[1];
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const int c = (i * 8 + part) * 4;
                const float4 gm = *reinterpret_cast<const float4*>(a.cln_g + c);
                const float4 bt = *reinterpret_cast<const float4*>(a.cln_b + c);
                float e[4] = {vrow[i].x * rstd * gm.x + bt.x, vrow[i].y * rstd * gm.y + bt.y,
                              vrow[i].z * rstd * gm.z + bt.z, vrow[i].w * rstd * gm.w + bt.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = (w0 * e[q] + b0) * __builtin_amdgcn_rcpf(1.f + __expf(-(w1 * e[q] + b1)));
                *reinterpret_cast<float4*>(a.g_out + (size_t)row * DM + c) = make_float4(e[0], e[1], e[2], e[3]);
              }
            }
          }
        }
      }
    } else {  // unaligned rows (e.g. the 771-wide mask head): element-wise, still lane-contiguous
      for (int idx = tid; idx < BM * 256; idx += NW * 64) {
        const int rr = idx >> 8, c = idx & 255;
        if (c < ncols && m0 + rr < a.M) {
          float y = Ys[rr * YS + c];
          if constexpr (RES) y = a.res[(size_t)(m0 + rr) * a.ldr + f0 + c] + a.alpha * y;
          if constexpr (OBF) reinterpret_cast<unsigned short*>(a.out)[(size_t)(m0 + rr) * a.ldo + f0 + c] = f2bf(y);
          else reinterpret_cast<float*>(a.out)[(size_t)(m0 + rr) * a.ldo + f0 + c] = y;
        }
      }
    }
    if (p + 1 < npass) __syncthreads();  // Ys is re-parked by the next pass
  }
  stamp(a.stamps, 5);
}

// =============================================================================== encoder chain
// FFN_out(l-1) [+ block LayerNorm]  ->  FFN_in(l)  ->  pre-LN + q/k/v projection(l), all for one 64-row
// panel that never leaves the CU between the stages: the residual rows live in registers (thread
// (row = tid>>3, part = tid&7) owns 32 elements as 8 float4), activations in LDS, only weights stream in.
// Saves, per encoder block, two launches and two panel prologue/epilogue round trips through L2.
struct FfnW {
  const float *ln_g, *ln_b;
  const bf16x8* w1;
  const float* b1;
  const bf16x8* w2;
  const float* b2;
};
struct ConvW {
  const float *dw_w, *dw_b, *bn_g, *bn_b, *bn_mean, *bn_var, *pw2_w, *pw2_b;
};
struct ChainArgs {
  const float* x;
  float* xout;
  const float* g;  // HAS_CONV: glu(LN(x)) rows from the attention output projection [M, 256]
  ConvW cv;
  int T;
  FfnW fa;                      // stage A: FFN + residual + final LayerNorm (fin_g/fin_b)
  const float *fin_g, *fin_b;
  FfnW fb;                      // stage B: FFN + residual
  const float *qln_g, *qln_b;   // stage C: LayerNorm -> [M,768] bf16
  const bf16x8* wqkv;
  const float* bqkv;
  unsigned short* qkv;
  int M, FF, NQKV;
  float alpha;
  unsigned long long* stamps;  // -DP_STAMPS diagnostic builds only
  int qkv_il8;                 // wqkv / bqkv in the 8-feature interleaved row order (CSSHIP_CHAIN_QKV_IL8)
  // TRAIN (csship_encoder_chain_train): the reference's dropouts of both FeedForwards (hidden [M, FF] and output
  // [M, 256], element indices as csship_ffn_fused_drop) and the rows the backward pass keeps: in front of stage A's
  // final LayerNorm (pre_ln_out) and behind it (ln_out = stage B's input)
  csship_dropout_cfg da_h, da_o, db_h, db_o;
  float* pre_ln_out;
  float* ln_out;
  // BNP (training, behind csship_outproj_conv_train): the kernel's input rows are formed here instead of by a launch of
  // their own - x3 = x2 + dropout(pw2(relu(z * scale + shift))), the tail of the conv module (conformer.py:298-303) with the
  // batch statistics csship_outproj_conv_train reduced - and leave as bn_out (the backward keeps them)
  const float *bn_z, *bn_x2, *bn_ss, *bn_pw2_w, *bn_pw2_b;
  csship_dropout_cfg d_bn;
  float* bn_out;
};

// LayerNorm of the register rows -> bf16 activation panel in LDS.  gam == nullptr: no affine - the caller folded gamma
// into the columns of the Linear that consumes the panel and W.beta into its bias (css_amd/models/conformer.py _pack):
// 16 float4 loads per thread = 128 KB per workgroup less through the texture path, and no wait behind the weight
// prefetch that was issued just before (loads return in order: phase stamps, 4 300 cycles per prologue with them)
DEV void rows_to_panel(unsigned char* panel, const float4 (&v)[8], const float* gam, const float* bet, int r, int part) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = sum8(s) * (1.f / DM);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
    ss += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
  if (gam == nullptr) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = i * 8 + part;
      *reinterpret_cast<uint2*>(panel + panel_off(r, c4 >> 1) + (c4 & 1) * 8) =
          make_uint2(pack_bf2((v[i].x - mean) * rstd, (v[i].y - mean) * rstd), pack_bf2((v[i].z - mean) * rstd, (v[i].w - mean) * rstd));
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = i * 8 + part;
    const float4 gm = *reinterpret_cast<const float4*>(gam + c4 * 4);
    const float4 bt = *reinterpret_cast<const float4*>(bet + c4 * 4);
    *reinterpret_cast<uint2*>(panel + panel_off(r, c4 >> 1) + (c4 & 1) * 8) =
        make_uint2(pack_bf2((v[i].x - mean) * rstd * gm.x + bt.x, (v[i].y - mean) * rstd * gm.y + bt.y),
                   pack_bf2((v[i].z - mean) * rstd * gm.z + bt.z, (v[i].w - mean) * rstd * gm.w + bt.w));
  }
}

// one FeedForward on the register rows: v <- v + alpha * FFN(LN(v))  [-> LayerNorm(fin)]
struct StageTrain {  // what a training stage needs beyond the inference one
  Drop dh, dout;
  int m0, M;
  float* pre_ln_out;
};

template <int NW, bool FINAL_LN, bool TRAIN = false>
DEV void ffn_stage(float4 (&v)[8], unsigned char* Xs, unsigned char* Hraw, const FfnW& f, const float* fin_g,
                   const float* fin_b, int FF, float alpha, int tid, int rot, unsigned long long* g_stamps_dev = nullptr,
                   int st0 = 0, const StageTrain& tr = StageTrain{}) {
  constexpr int NT = 16 / NW;
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15, r = tid >> 3, part = tid & 7;
  const int nchunk = FF / DM, ks2 = FF / 32;
  unsigned char* const Hs0 = Hraw;
  unsigned char* const Hs1 = Hraw + PANEL_BYTES;
  const bf16x8* w1s = f.w1 + (size_t)(wave * NT) * KS * 64 + lane;
  const bf16x8* w2s = f.w2 + (size_t)(wave * NT) * ks2 * 64 + lane;
  bf16x8 wq[PD][NT];
  prefetch_first<NT>(wq, w1s + (size_t)(rot % nchunk) * 16 * KS * 64, KS, NT);
  rows_to_panel(Xs, v, f.ln_g, f.ln_b, r, part);
  __syncthreads();  // also: every wave has left the previous stage's Ys reads
  P_T(st0 + 0)
  f32x4 o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 b2v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) b2v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ci = 0; ci < nchunk; ++ci) {
    const int c = (ci + rot) % nchunk;
    f32x4 h[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {  // the bias is the accumulators' initial value
      const float4 bb = (P_ABL & 32) ? make_float4(0.f, 0.f, 0.f, 0.f)
                                     : *reinterpret_cast<const float4*>(f.b1 + c * DM + (wave * NT + t) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    const bf16x8* w1c = w1s + (size_t)c * 16 * KS * 64;
    const bf16x8* w2c = w2s + (size_t)c * KS * 64;
#ifdef P_STAMPS
    unsigned long long tq0_ = 0, tq1_ = 0, tq2_ = 0, tq3_ = 0, tq4_ = 0;
    if (st0 == 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq0_)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
    mma_pass<NT>(h, Xs, n, g, NT, w1c, KS, w2c, ks2, NT, wq);
#ifdef P_STAMPS
    if (st0 == 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq1_)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
    unsigned char* hs = (ci & 1) ? Hs1 : Hs0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int hc = (wave * NT + t) * 16 + 4 * g;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {  // ReLU on the packed pairs: 2 conversions + 2 integer max per 4 values
        uint2 pk;
        if (TRAIN && tr.dh.thresh != 0u) {  // relu, then the hidden activation's dropout (conformer.py:317), element (row, c*256 + hc)
          pk = drop_relu_pack4(tr.dh, (unsigned)((tr.m0 + rt * 16 + n) * FF + c * DM + hc), h[t][rt][0], h[t][rt][1], h[t][rt][2], h[t][rt][3]);
        } else {
          pk = make_uint2(relu_bf2(pack_bf2(h[t][rt][0], h[t][rt][1])), relu_bf2(pack_bf2(h[t][rt][2], h[t][rt][3])));
        }
        *reinterpret_cast<uint2*>(hs + panel_off(rt * 16 + n, hc >> 3) + (hc & 7) * 2) = pk;
      }
    }
#ifdef P_STAMPS
    if (st0 == 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq2_)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
    __syncthreads();
#ifdef P_STAMPS
    if (st0 == 1) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq3_)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
    const bf16x8* w1n = w1s + (size_t)((c + 1) % nchunk) * 16 * KS * 64;
    // last chunk: nothing follows in this stream (the next stage primes its own), and the output bias is fetched
    // now, ahead of the pass, instead of behind 8 KB per wave of prefetches nobody uses
    if (ci + 1 == nchunk) {
#pragma unroll
      for (int t = 0; t < NT; ++t) b2v[t] = *reinterpret_cast<const float4*>(f.b2 + (wave * NT + t) * 16 + 4 * g);
    }
    mma_pass<NT>(o, hs, n, g, NT, w2c, ks2, w1n, KS, ci + 1 < nchunk ? NT : 0, wq);
#ifdef P_STAMPS
    if (st0 == 1 && g_stamps_dev != nullptr && (threadIdx.x & 63) == 0) {
      __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq4_)::"memory"); __builtin_amdgcn_sched_barrier(0);
      unsigned long long* q_ = g_stamps_dev + (size_t)blockIdx.x * 32 + 12;
      // wave 0 and wave 7 only: slots 12..15 (wave 0), 16..19 (wave 7): W1 pass, hidden epilogue, barrier wait, W2 pass
      const int wv_ = threadIdx.x >> 6;
      if (wv_ == 0 || wv_ == 7) {
        unsigned long long* d_ = q_ + (wv_ == 7 ? 4 : 0);
        d_[0] += tq1_ - tq0_; d_[1] += tq2_ - tq1_; d_[2] += tq3_ - tq2_; d_[3] += tq4_ - tq3_;
      }
    }
#endif
  }
  P_T(st0 + 1)
  __syncthreads();  // every wave is done reading the hidden panels
  float* ys = reinterpret_cast<float*>(Hraw);
  park_tiles<NT, false>(ys, o, NT, wave * NT * 16, g, n, b2v);
  __syncthreads();
  P_T(st0 + 2)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float4 y = *reinterpret_cast<const float4*>(ys + r * YS + (i * 8 + part) * 4);
    if constexpr (TRAIN) y = drop_apply4(tr.dout, (unsigned)((tr.m0 + r) * DM + (i * 8 + part) * 4), y);  // conformer.py:319
    v[i] = make_float4(v[i].x + alpha * y.x, v[i].y + alpha * y.y, v[i].z + alpha * y.z, v[i].w + alpha * y.w);
  }
  if constexpr (FINAL_LN) {
    if constexpr (TRAIN) {
      if (tr.pre_ln_out != nullptr && tr.m0 + r < tr.M) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(tr.pre_ln_out + (size_t)(tr.m0 + r) * DM + (i * 8 + part) * 4) = v[i];
      }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = sum8(s) * (1.f / DM);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 gm = *reinterpret_cast<const float4*>(fin_g + (i * 8 + part) * 4);
      const float4 bt = *reinterpret_cast<const float4*>(fin_b + (i * 8 + part) * 4);
      v[i].x = v[i].x * rstd * gm.x + bt.x; v[i].y = v[i].y * rstd * gm.y + bt.y;
      v[i].z = v[i].z * rstd * gm.z + bt.z; v[i].w = v[i].w * rstd * gm.w + bt.w;
    }
  }
}

template <int NW, bool HAS_CONV, bool HAS_A, bool HAS_B, bool HAS_C, bool TRAIN = false>
__global__ __launch_bounds__(NW * 64) void chain_kernel(ChainArgs a) {
  static_assert(!TRAIN || !HAS_CONV, "the training forward runs the conv module with batch statistics, on its own");
  static_assert(NW * 64 == 8 * BM, "row phases assume 8 lanes per panel row");
  constexpr int NT = 16 / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[PANEL_BYTES + YS_BYTES];
  unsigned char* const Xs = smem;
  unsigned char* const Hraw = smem + PANEL_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int r = tid >> 3, part = tid & 7;
  const int m0 = blockIdx.x * BM, row = m0 + r, rowc = min(row, a.M - 1);
  float4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(a.x + (size_t)rowc * DM + (i * 8 + part) * 4);
  if constexpr (HAS_CONV) {
    // x <- x + pw2(relu(BN(dwconv_33(g))))   (css/models/conformer.py:294-301, 357) for this panel's 64 rows.
    // g rows m0-16 .. m0+79 are staged in LDS (the panel + hidden regions are still free); thread
    // (channel c, half) walks 32 consecutive rows; taps that leave the row's own utterance are masked
    // (rows are flattened (b, t), a panel may straddle a batch boundary).
    constexpr int KSZ = 33, HALO = 16, GROWS = BM + KSZ - 1;  // 96 rows x 1 KB = 96 KB
    static_assert(GROWS * DM * 4 <= PANEL_BYTES + YS_BYTES, "g tile must fit the panel + hidden regions");
    float* Gs = reinterpret_cast<float*>(smem);
    for (int i = tid; i < GROWS * (DM / 4); i += NW * 64) {
      const int gr = i / (DM / 4), c4 = i % (DM / 4);
      const int m = m0 - HALO + gr;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m >= 0 && m < a.M) val = *reinterpret_cast<const float4*>(a.g + (size_t)m * DM + c4 * 4);
      *reinterpret_cast<float4*>(Gs + gr * DM + c4 * 4) = val;
    }
    const int c = tid % DM, half = tid / DM;
    float wk[KSZ];
#pragma unroll
    for (int k = 0; k < KSZ; ++k) wk[k] = a.cv.dw_w[k * DM + c];  // [k][D]: lanes read consecutive floats
    const float cb = a.cv.dw_b[c];
    const float scale = a.cv.bn_g[c] * rsqrtf(a.cv.bn_var[c] + 1e-5f);
    const float shift = a.cv.bn_b[c] - a.cv.bn_mean[c] * scale;
    const float pw2w = a.cv.pw2_w[0], pw2b = a.cv.pw2_b[0];
    __syncthreads();
    float yout[32];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const int tb = half * 32 + blk * 8;      // first output row of the block within the panel
      const int tfirst = (m0 + tb) % a.T;      // its frame index inside the utterance (uniform)
      const bool interior = tfirst >= HALO && tfirst + 7 + HALO < a.T;  // no tap of the block leaves the utterance
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = cb;
      if (interior) {
#pragma unroll
        for (int u = 0; u < 8 + KSZ - 1; ++u) {
          const float gv = Gs[(tb + u) * DM + c];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = u - j;
            if (k >= 0 && k < KSZ) acc[j] += wk[k] * gv;
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 8 + KSZ - 1; ++u) {
          const float gv = Gs[(tb + u) * DM + c];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int k = u - j;
            if (k >= 0 && k < KSZ) {
              int tj = tfirst + j;             // frame of output j (the block itself may wrap into the next utterance)
              if (tj >= a.T) tj -= a.T;
              const int tt = tj + k - HALO;    // frame the tap reads
              if (tt >= 0 && tt < a.T) acc[j] += wk[k] * gv;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) yout[blk * 8 + j] = fmaxf(acc[j] * scale + shift, 0.f) * pw2w + pw2b;
    }
    __syncthreads();  // all taps read: the region is re-used as the [row][channel] exchange tile
    float* ys = reinterpret_cast<float*>(Hraw);
#pragma unroll
    for (int j = 0; j < 32; ++j) ys[(half * 32 + j) * YS + c] = yout[j];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 y = *reinterpret_cast<const float4*>(ys + r * YS + (i * 8 + part) * 4);
      v[i] = make_float4(v[i].x + y.x, v[i].y + y.y, v[i].z + y.z, v[i].w + y.w);
    }
    __syncthreads();  // ys is overwritten by the first FFN stage's hidden panel
  }
#ifdef P_STAMPS
  unsigned long long* const g_stamps_dev = a.stamps;
#endif
  P_T(0)
  if constexpr (TRAIN) {
    if constexpr (HAS_A) {
      const StageTrain ta{drop_make(a.da_h.seed, a.da_h.stream, a.da_h.p), drop_make(a.da_o.seed, a.da_o.stream, a.da_o.p), m0, a.M, a.pre_ln_out};
      ffn_stage<NW, true, true>(v, Xs, Hraw, a.fa, a.fin_g, a.fin_b, a.FF, a.alpha, tid, blockIdx.x, a.stamps, 1, ta);
      if (a.ln_out != nullptr && row < a.M) {  // the next block's input: the backward of its first FeedForward needs it
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(a.ln_out + (size_t)row * DM + (i * 8 + part) * 4) = v[i];
      }
    }
    if constexpr (HAS_B) {
      const StageTrain tb{drop_make(a.db_h.seed, a.db_h.stream, a.db_h.p), drop_make(a.db_o.seed, a.db_o.stream, a.db_o.p), m0, a.M, nullptr};
      ffn_stage<NW, false, true>(v, Xs, Hraw, a.fb, nullptr, nullptr, a.FF, a.alpha, tid, blockIdx.x + 2, a.stamps, 5, tb);
    }
  } else {
  if constexpr (HAS_A) ffn_stage<NW, true>(v, Xs, Hraw, a.fa, a.fin_g, a.fin_b, a.FF, a.alpha, tid, blockIdx.x, a.stamps, 1);
  P_T(4)
  if constexpr (HAS_B) ffn_stage<NW, false>(v, Xs, Hraw, a.fb, nullptr, nullptr, a.FF, a.alpha, tid, blockIdx.x + 2, a.stamps, 5);
  }
  P_T(8)
  if (row < a.M && !(P_ABL & 16)) {  // the residual stream the attention output projection adds to
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(a.xout + (size_t)row * DM + (i * 8 + part) * 4) = v[i];
  }
  if constexpr (HAS_C) {
    const int ntiles = (a.NQKV + 15) / 16, npass = (ntiles + 15) / 16;
    bf16x8 wq[PD][NT];
    const bf16x8* wbase = a.wqkv + lane;
    prefetch_first<NT>(wq, wbase + (size_t)(wave * NT) * KS * 64, KS, max(0, min(NT, ntiles - wave * NT)));
    rows_to_panel(Xs, v, a.qln_g, a.qln_b, r, part);
    __syncthreads();
    P_T(9)
    // The projection is written straight from the transposed accumulators (bias = their initial value): a lane owns
    // one token row per row tile and 4 consecutive features per tile = one 8-byte bf16 store.  No fp32 staging tile
    // and no barrier in this loop (phase stamps: the park / barrier / row read-back / barrier of the staged form cost
    // 4 300 cycles per 256-column pass against 2 048 cycles of MFMA).
    for (int p = 0; p < npass; ++p) {
      const int tile0 = p * 16 + wave * NT;
      const int nt = max(0, min(NT, ntiles - tile0));
      const int nnt = max(0, min(NT, ntiles - (tile0 + 16)));
      f32x4 acc[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 bb = load_bias4(a.bqkv, (tile0 + t) * 16 + 4 * g, a.NQKV);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
      }
      const bf16x8* wc = wbase + (size_t)tile0 * KS * 64;
      mma_pass<NT>(acc, Xs, n, g, nt, wc, KS, wc + (size_t)16 * KS * 64, KS, nnt, wq);
      if (NT == 2 && a.qkv_il8) {
        // CSSHIP_CHAIN_QKV_IL8 weights: rows of the wave's two tiles interleaved so that a lane's 2 x 4 results are
        // EIGHT CONSECUTIVE features of its token row: one 16-byte store per row tile instead of two 8-byte ones
        // (half the store instructions; every instruction writes 64 contiguous bytes of 16 rows)
        const int f = (p * 16 + wave * NT) * 16 + 8 * g;
        if (nt == NT && f + 7 < a.NQKV) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            const int orow = m0 + rt * 16 + n;
            if (orow < a.M && (!(P_ABL & 8) || acc[0][rt][0] == 123.456f))
              *reinterpret_cast<uint4*>(a.qkv + (size_t)orow * a.NQKV + f) =
                  make_uint4(pack_bf2(acc[0][rt][0], acc[0][rt][1]), pack_bf2(acc[0][rt][2], acc[0][rt][3]),
                             pack_bf2(acc[NT - 1][rt][0], acc[NT - 1][rt][1]), pack_bf2(acc[NT - 1][rt][2], acc[NT - 1][rt][3]));
          }
        }
        continue;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int f = (tile0 + t) * 16 + 4 * g;
        if (t < nt && f + 3 < a.NQKV) {  // NQKV % 8 == 0 (checked on the host)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            const int orow = m0 + rt * 16 + n;
            if (orow < a.M && (!(P_ABL & 8) || acc[t][rt][0] == 123.456f))
              *reinterpret_cast<uint2*>(a.qkv + (size_t)orow * a.NQKV + f) =
                  make_uint2(pack_bf2(acc[t][rt][0], acc[t][rt][1]), pack_bf2(acc[t][rt][2], acc[t][rt][3]));
          }
        }
      }
    }
    P_T(10)
  }
}

// =============================================================================== chain, static form
// Same operator chain and the same arithmetic as chain_kernel above, for the shapes the encoder actually runs (whole
// 64-row panels: M % 64 == 0; q | k | v of 768 columns; d_ffn a multiple of 256 up to 2048): every global memory
// instruction of the kernel is UNCONDITIONAL and its count is known at compile time.  Why that matters (round 5, the
// ISA of the first form): hipcc's s_waitcnt insertion merges the pending-load state of the two sides of every branch
// pessimistically.  The first form had three loads / load groups behind run-time conditions inside its matrix passes -
// "no prefetch behind the last chunk" (t < nnt with nnt = 0 or NT), "output bias in front of the last W2 pass" and the
// bounds-checked bias / weight loads of the q | k | v stage - and behind every one of them the compiler had to assume
// the FEWER-loads path: the W2 pass of every hidden chunk ended on s_waitcnt vmcnt(3) / vmcnt(1) where vmcnt(9) /
// vmcnt(8) was meant - a wait for weight fragments requested only a k-step earlier, i.e. a full L2 round trip per pass,
// 8 per launch (the 700-1 000 cycles the W2 passes ran behind the W1 passes in the phase stamps) - and every pass of the
// q | k | v stage began with s_waitcnt vmcnt(0), draining the whole prefetch ring (14 800 cycles for 3 x 2 048 of MFMA).
// Here:
//   * all per-channel parameters (four LayerNorm affines when they are not folded into the weights, hidden / output
//     biases of both FeedForwards, the q | k | v bias: 13-29 KB) are staged into LDS ONCE at kernel entry by four
//     unconditional 16-byte loads per thread; accumulators start from ds_read values (no global load - and no wait behind
//     the in-order weight prefetches - at the start of a pass), LayerNorm prologues read gamma / beta from LDS;
//   * ONE weight ring for the whole kernel: the last W2 pass of a FeedForward refills the ring with the first four
//     k-steps of the NEXT stage's stream (a pointer select, not a predicated load), so the global -> register pipeline
//     never drains between stages and no stage primes its own ring;
//   * tile counts are template constants (mma_pass_s has no t < nt tests), stores are unpredicated.
// Anything else (ragged last panel, other widths, the fused conv front end) runs the first form.
constexpr int CS_LN = 8 * DM;                                     // A gamma | beta, final gamma | beta, B gamma | beta, q/k/v gamma | beta
constexpr int CS_PARAMS = CS_LN + 2 * CS_FFMAX + 2 * DM + 768;    // floats: + b1 A, b2 A, b1 B, b2 B, b q|k|v

template <int NT, bool NEXT>
DEV void mma_pass_s(f32x4 (&acc)[NT][4], const unsigned char* panel, int n, int g, const bf16x8* w, int tstride,
                    const bf16x8* wn, int nstride, bf16x8 (&wq)[PD][NT]) {
  bf16x8 xa[2][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) xa[0][rt] = *reinterpret_cast<const bf16x8*>(panel + panel_off(rt * 16 + n, g));
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 wc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wc[t] = wq[ks % PD][t];
    if (ks + PD < KS) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wq[ks % PD][t] = w[(size_t)(t * tstride + ks + PD) * 64];
    } else if constexpr (NEXT) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wq[ks % PD][t] = wn[(size_t)(t * nstride + ks + PD - KS) * 64];
    }
    if (ks + 1 < KS) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        xa[(ks + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(panel + panel_off(rt * 16 + n, (ks + 1) * 4 + g));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], xa[ks & 1][rt], acc[t][rt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// rows_to_panel with gamma / beta in LDS (gam == nullptr: folded into the consuming Linear)
DEV void rows_to_panel_s(unsigned char* panel, const float4 (&v)[8], const float* gam, const float* bet, int r, int part) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = sum8(s) * (1.f / DM);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
    ss += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
  }
  const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
  if (gam == nullptr) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = i * 8 + part;
      *reinterpret_cast<uint2*>(panel + panel_off(r, c4 >> 1) + (c4 & 1) * 8) =
          make_uint2(pack_bf2((v[i].x - mean) * rstd, (v[i].y - mean) * rstd), pack_bf2((v[i].z - mean) * rstd, (v[i].w - mean) * rstd));
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = i * 8 + part;
    const float4 gm = *reinterpret_cast<const float4*>(gam + c4 * 4);
    const float4 bt = *reinterpret_cast<const float4*>(bet + c4 * 4);
    *reinterpret_cast<uint2*>(panel + panel_off(r, c4 >> 1) + (c4 & 1) * 8) =
        make_uint2(pack_bf2((v[i].x - mean) * rstd * gm.x + bt.x, (v[i].y - mean) * rstd * gm.y + bt.y),
                   pack_bf2((v[i].z - mean) * rstd * gm.z + bt.z, (v[i].w - mean) * rstd * gm.w + bt.w));
  }
}

// one FeedForward on the register rows, every parameter from LDS; the ring arrives primed with this stage's first four
// k-steps and leaves primed with next_w's (lane-adjusted pointer at block (tile 0 of this wave, k-step 0), tile stride KS)
template <int NW, bool FINAL_LN, bool TRAIN>
DEV void ffn_stage_s(float4 (&v)[8], unsigned char* Xs, unsigned char* Hraw, const float* lng, const float* fing,
                     const float* b1s, const float* b2s, const FfnW& f, int FF, float alpha, int tid, int rot,
                     bf16x8 (&wq)[PD][16 / NW], const bf16x8* next_w, unsigned long long* g_stamps_dev, int st0,
                     const StageTrain& tr) {
  constexpr int NT = 16 / NW;
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15, r = tid >> 3, part = tid & 7;
  const int nchunk = FF / DM, ks2 = FF / 32;
  unsigned char* const Hs0 = Hraw;
  unsigned char* const Hs1 = Hraw + PANEL_BYTES;
  const bf16x8* w1s = f.w1 + (size_t)(wave * NT) * KS * 64 + lane;
  const bf16x8* w2s = f.w2 + (size_t)(wave * NT) * ks2 * 64 + lane;
  rows_to_panel_s(Xs, v, lng, lng != nullptr ? lng + DM : nullptr, r, part);
  __syncthreads();  // also: every wave has left the previous stage's Ys reads
  P_T(st0 + 0)
  f32x4 o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ci = 0; ci < nchunk; ++ci) {
    const int c = (ci + rot) % nchunk;
#ifdef P_STAMPS
    unsigned long long tq0_ = 0, tq1_ = 0, tq2_ = 0, tq3_ = 0;
#endif
    f32x4 h[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {  // the bias is the accumulators' initial value
      const float4 bb = *reinterpret_cast<const float4*>(b1s + c * DM + (wave * NT + t) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    const bf16x8* w1c = w1s + (size_t)c * 16 * KS * 64;
    const bf16x8* w2c = w2s + (size_t)c * KS * 64;
    P_TQ(tq0_)
    mma_pass_s<NT, true>(h, Xs, n, g, w1c, KS, w2c, ks2, wq);
    P_TQ(tq1_)
    unsigned char* hs = (ci & 1) ? Hs1 : Hs0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int hc = (wave * NT + t) * 16 + 4 * g;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {  // ReLU on the packed pairs: 2 conversions + 2 integer max per 4 values
        uint2 pk;
        if (TRAIN && tr.dh.thresh != 0u) {  // relu, then the hidden activation's dropout (conformer.py:317), element (row, c*256 + hc)
          pk = drop_relu_pack4(tr.dh, (unsigned)((tr.m0 + rt * 16 + n) * FF + c * DM + hc), h[t][rt][0], h[t][rt][1], h[t][rt][2], h[t][rt][3]);
        } else {
          pk = make_uint2(relu_bf2(pack_bf2(h[t][rt][0], h[t][rt][1])), relu_bf2(pack_bf2(h[t][rt][2], h[t][rt][3])));
        }
        *reinterpret_cast<uint2*>(hs + panel_off(rt * 16 + n, hc >> 3) + (hc & 7) * 2) = pk;
      }
    }
    P_TQ(tq2_)
    __syncthreads();
    P_TQ(tq3_)
    // behind the last chunk the ring takes the NEXT stage's first k-steps: a pointer select, the loads stay unconditional
    const bf16x8* wn = (ci + 1 < nchunk) ? w1s + (size_t)((c + 1) % nchunk) * 16 * KS * 64 : next_w;
    mma_pass_s<NT, true>(o, hs, n, g, w2c, ks2, wn, KS, wq);
#ifdef P_STAMPS
    if (st0 == 1 && g_stamps_dev != nullptr && (threadIdx.x & 63) == 0) {
      unsigned long long tq4_;
      P_TQ(tq4_)
      const int wv_ = threadIdx.x >> 6;  // wave 0 and wave 7 only: slots 12..15 / 16..19: W1 pass, hidden epilogue, barrier wait, W2 pass
      if (wv_ == 0 || wv_ == 7) {
        unsigned long long* d_ = g_stamps_dev + (size_t)blockIdx.x * 32 + 12 + (wv_ == 7 ? 4 : 0);
        d_[0] += tq1_ - tq0_; d_[1] += tq2_ - tq1_; d_[2] += tq3_ - tq2_; d_[3] += tq4_ - tq3_;
      }
    }
#endif
  }
  P_T(st0 + 1)
  __syncthreads();  // every wave is done reading the hidden panels
  float* ys = reinterpret_cast<float*>(Hraw);
  float4 b2v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) b2v[t] = *reinterpret_cast<const float4*>(b2s + (wave * NT + t) * 16 + 4 * g);
  park_tiles<NT, false>(ys, o, NT, wave * NT * 16, g, n, b2v);
  __syncthreads();
  P_T(st0 + 2)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float4 y = *reinterpret_cast<const float4*>(ys + r * YS + (i * 8 + part) * 4);
    if constexpr (TRAIN) y = drop_apply4(tr.dout, (unsigned)((tr.m0 + r) * DM + (i * 8 + part) * 4), y);  // conformer.py:319
    v[i] = make_float4(v[i].x + alpha * y.x, v[i].y + alpha * y.y, v[i].z + alpha * y.z, v[i].w + alpha * y.w);
  }
  if constexpr (FINAL_LN) {
    if constexpr (TRAIN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(tr.pre_ln_out + (size_t)(tr.m0 + r) * DM + (i * 8 + part) * 4) = v[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = sum8(s) * (1.f / DM);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 gm = *reinterpret_cast<const float4*>(fing + (i * 8 + part) * 4);
      const float4 bt = *reinterpret_cast<const float4*>(fing + DM + (i * 8 + part) * 4);
      v[i].x = v[i].x * rstd * gm.x + bt.x; v[i].y = v[i].y * rstd * gm.y + bt.y;
      v[i].z = v[i].z * rstd * gm.z + bt.z; v[i].w = v[i].w * rstd * gm.w + bt.w;
    }
  }
}

// ---- pipelined form of the stage (chain_s_kernel<..., PIPE>): the hidden epilogue of chunk c (ReLU, bf16 pack, 8-byte LDS
// stores: 450-600 cycles of VALU / LDS work per chunk with the matrix pipe idle) rides INSIDE the W2 pass of chunk c - 1,
// one accumulator tile per k-step, so that only the first chunk's epilogue of a FeedForward is exposed.  Pass order
//   W1(0) | epi(0) B W1(1) | [W2(0) + epi(1)] B W1(2) | [W2(1) + epi(2)] B W1(3) | [W2(2) + epi(3)] B | W2(3)      (B = barrier)
// - same barrier count, same double-buffered hidden panels (W2(c - 1) reads panel (c - 1) & 1 while epi(c) fills panel
// c & 1, which W2(c - 2) left before the previous barrier), same registers at the peak (h stays live through a W2 pass,
// where the first form had it dead; the W1 pass holds h and o either way).  The weight ring follows the new order.
template <int NT, bool TRAIN>
DEV void hidden_tile(unsigned char* hs, const f32x4& hv, int row, int hc, const StageTrain& tr, int FF, int c) {
  uint2 pk;
  if (TRAIN && tr.dh.thresh != 0u) {  // relu, then the hidden activation's dropout (conformer.py:317), element (row, c*256 + hc)
    pk = drop_relu_pack4(tr.dh, (unsigned)((tr.m0 + row) * FF + c * DM + hc), hv[0], hv[1], hv[2], hv[3]);
  } else {
    pk = make_uint2(relu_bf2(pack_bf2(hv[0], hv[1])), relu_bf2(pack_bf2(hv[2], hv[3])));
  }
  *reinterpret_cast<uint2*>(hs + panel_off(row, hc >> 3) + (hc & 7) * 2) = pk;
}

// mma_pass_s with the hidden epilogue of the NEXT chunk as a side job: behind the MFMAs of k-step ks, accumulator tile
// (t = ks / 4, row tile ks % 4) of h goes to the panel hs_out
template <int NT, bool TRAIN>
DEV void mma_pass_epi(f32x4 (&acc)[NT][4], const unsigned char* panel, int n, int g, const bf16x8* w, int tstride,
                      const bf16x8* wn, int nstride, bf16x8 (&wq)[PD][NT], unsigned char* hs_out, const f32x4 (&h)[NT][4],
                      int wave, const StageTrain& tr, int FF, int c) {
  static_assert(NT * 4 == KS, "one accumulator tile per k-step");
  bf16x8 xa[2][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) xa[0][rt] = *reinterpret_cast<const bf16x8*>(panel + panel_off(rt * 16 + n, g));
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 wc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wc[t] = wq[ks % PD][t];
    if (ks + PD < KS) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wq[ks % PD][t] = w[(size_t)(t * tstride + ks + PD) * 64];
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) wq[ks % PD][t] = wn[(size_t)(t * nstride + ks + PD - KS) * 64];
    }
    if (ks + 1 < KS) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        xa[(ks + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(panel + panel_off(rt * 16 + n, (ks + 1) * 4 + g));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], xa[ks & 1][rt], acc[t][rt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    hidden_tile<NT, TRAIN>(hs_out, h[ks >> 2][ks & 3], (ks & 3) * 16 + n, (wave * NT + (ks >> 2)) * 16 + 4 * g, tr, FF, c);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// rrow (training form): this thread's pieces of its input row in MEMORY (the stage's input rows are kept for the backward
// anyway: the thread wrote them itself a stage ago) - the 32 registers of v are dead during the passes, which is what lets the
// pipelined form (h AND o live through a W2 pass) fit the 256 a wave has; the rows come back behind the last pass's final
// weight request, under park + barrier.
template <int NW, bool FINAL_LN, bool TRAIN>
DEV void ffn_stage_p(float4 (&v)[8], unsigned char* Xs, unsigned char* Hraw, const float* lng, const float* fing,
                     const float* b1s, const float* b2s, const FfnW& f, int FF, float alpha, int tid, int rot,
                     bf16x8 (&wq)[PD][16 / NW], const bf16x8* next_w, unsigned long long* g_stamps_dev, int st0,
                     const StageTrain& tr, const float* rrow = nullptr) {
  constexpr int NT = 16 / NW;
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15, r = tid >> 3, part = tid & 7;
  const int nchunk = FF / DM, ks2 = FF / 32;   // nchunk >= 2 (checked on the host)
  unsigned char* const Hs0 = Hraw;
  unsigned char* const Hs1 = Hraw + PANEL_BYTES;
  const bf16x8* w1s = f.w1 + (size_t)(wave * NT) * KS * 64 + lane;
  const bf16x8* w2s = f.w2 + (size_t)(wave * NT) * ks2 * 64 + lane;
  rows_to_panel_s(Xs, v, lng, lng != nullptr ? lng + DM : nullptr, r, part);
  __syncthreads();  // also: every wave has left the previous stage's Ys reads
  P_T(st0 + 0)
  f32x4 o[NT][4], h[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto chunk_of = [&](int ci) { return (ci + rot) % nchunk; };
  auto w1_of = [&](int c) { return w1s + (size_t)c * 16 * KS * 64; };
  auto w2_of = [&](int c) { return w2s + (size_t)c * KS * 64; };
#define FS_HINIT(C)                                                                                        \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                         \
    const float4 bb = *reinterpret_cast<const float4*>(b1s + (C) * DM + (wave * NT + t) * 16 + 4 * g);     \
    _Pragma("unroll") for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};             \
  }
  const int c0 = chunk_of(0), c1 = chunk_of(1);
  FS_HINIT(c0)
  mma_pass_s<NT, true>(h, Xs, n, g, w1_of(c0), KS, w1_of(c1), KS, wq);
#pragma unroll
  for (int q = 0; q < NT * 4; ++q)   // the first chunk's epilogue has no W2 pass to ride in
    hidden_tile<NT, TRAIN>(Hs0, h[q >> 2][q & 3], (q & 3) * 16 + n, (wave * NT + (q >> 2)) * 16 + 4 * g, tr, FF, c0);
  __syncthreads();
  FS_HINIT(c1)
  mma_pass_s<NT, true>(h, Xs, n, g, w1_of(c1), KS, w2_of(c0), ks2, wq);
  for (int ci = 1; ci + 1 < nchunk; ++ci) {   // [W2(ci - 1) + epi(ci)] B W1(ci + 1)
    const int cp = chunk_of(ci - 1), c = chunk_of(ci), cn = chunk_of(ci + 1);
    mma_pass_epi<NT, TRAIN>(o, ((ci - 1) & 1) ? Hs1 : Hs0, n, g, w2_of(cp), ks2, w1_of(cn), KS, wq, (ci & 1) ? Hs1 : Hs0, h, wave, tr, FF, c);
    __syncthreads();
    FS_HINIT(cn)
    mma_pass_s<NT, true>(h, Xs, n, g, w1_of(cn), KS, w2_of(c), ks2, wq);
  }
  {
    const int ci = nchunk - 1, cp = chunk_of(ci - 1), c = chunk_of(ci);
    mma_pass_epi<NT, TRAIN>(o, ((ci - 1) & 1) ? Hs1 : Hs0, n, g, w2_of(cp), ks2, w2_of(c), ks2, wq, (ci & 1) ? Hs1 : Hs0, h, wave, tr, FF, c);
    __syncthreads();
    mma_pass_s<NT, true>(o, (ci & 1) ? Hs1 : Hs0, n, g, w2_of(c), ks2, next_w, KS, wq);
  }
#undef FS_HINIT
  if constexpr (TRAIN) {   // (unconditional: a run-time choice would keep v alive through the passes)
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(rrow + i * 32);
  }
  P_T(st0 + 1)
  __syncthreads();  // every wave is done reading the hidden panels
  float* ys = reinterpret_cast<float*>(Hraw);
  float4 b2v[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) b2v[t] = *reinterpret_cast<const float4*>(b2s + (wave * NT + t) * 16 + 4 * g);
  park_tiles<NT, false>(ys, o, NT, wave * NT * 16, g, n, b2v);
  __syncthreads();
  P_T(st0 + 2)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float4 y = *reinterpret_cast<const float4*>(ys + r * YS + (i * 8 + part) * 4);
    if constexpr (TRAIN) y = drop_apply4(tr.dout, (unsigned)((tr.m0 + r) * DM + (i * 8 + part) * 4), y);  // conformer.py:319
    v[i] = make_float4(v[i].x + alpha * y.x, v[i].y + alpha * y.y, v[i].z + alpha * y.z, v[i].w + alpha * y.w);
  }
  if constexpr (FINAL_LN) {
    if constexpr (TRAIN) {
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(tr.pre_ln_out + (size_t)(tr.m0 + r) * DM + (i * 8 + part) * 4) = v[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = sum8(s) * (1.f / DM);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 gm = *reinterpret_cast<const float4*>(fing + (i * 8 + part) * 4);
      const float4 bt = *reinterpret_cast<const float4*>(fing + DM + (i * 8 + part) * 4);
      v[i].x = v[i].x * rstd * gm.x + bt.x; v[i].y = v[i].y * rstd * gm.y + bt.y;
      v[i].z = v[i].z * rstd * gm.z + bt.z; v[i].w = v[i].w * rstd * gm.w + bt.w;
    }
  }
}

// one 256-feature pass of the q | k | v projection, written straight from the transposed accumulators
template <int NT, bool NEXT, bool IL8>
DEV void qkv_pass_s(const unsigned char* Xs, const float* bq, const bf16x8* wc, bf16x8 (&wq)[PD][NT], unsigned short* qkv,
                    int m0, int tile0, int n, int g) {
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 bb = *reinterpret_cast<const float4*>(bq + (tile0 + t) * 16 + 4 * g);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
  }
  mma_pass_s<NT, NEXT>(acc, Xs, n, g, wc, KS, wc + (size_t)16 * KS * 64, KS, wq);
  if constexpr (IL8 && NT == 2) {  // a lane's 2 x 4 results are EIGHT CONSECUTIVE features of its token row (ops.pack_frag_il8)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
      *reinterpret_cast<uint4*>(qkv + (size_t)(m0 + rt * 16 + n) * 768 + tile0 * 16 + 8 * g) =
          make_uint4(pack_bf2(acc[0][rt][0], acc[0][rt][1]), pack_bf2(acc[0][rt][2], acc[0][rt][3]),
                     pack_bf2(acc[1][rt][0], acc[1][rt][1]), pack_bf2(acc[1][rt][2], acc[1][rt][3]));
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        *reinterpret_cast<uint2*>(qkv + (size_t)(m0 + rt * 16 + n) * 768 + (tile0 + t) * 16 + 4 * g) =
            make_uint2(pack_bf2(acc[t][rt][0], acc[t][rt][1]), pack_bf2(acc[t][rt][2], acc[t][rt][3]));
  }
}

// float4 j of the bias space b1 A (q1) | b2 A (64) | b1 B (q1) | b2 B (64) | b q|k|v (192): where it is read from and
// where it lands in the LDS parameter block (floats); lanes without work read `dummy` and write the dump block
struct ParamSrc {
  const float* p;
  int dst;
};
template <bool HAS_A, bool HAS_B, bool HAS_C>
DEV ParamSrc param_src(const ChainArgs& a, int j, int q1, const float* dummy, int lane) {
  const int e1 = q1, e2 = q1 + 64, e3 = 2 * q1 + 64, e4 = 2 * q1 + 128, e5 = 2 * q1 + 320;
  const float* const bA1 = HAS_A ? a.fa.b1 : nullptr;
  const float* const bA2 = HAS_A ? a.fa.b2 : nullptr;
  const float* const bB1 = HAS_B ? a.fb.b1 : nullptr;
  const float* const bB2 = HAS_B ? a.fb.b2 : nullptr;
  const float* const bQ = HAS_C ? a.bqkv : nullptr;
  const float* const base = j < e1 ? bA1 : j < e2 ? bA2 : j < e3 ? bB1 : j < e4 ? bB2 : j < e5 ? bQ : nullptr;
  const int off = j < e1 ? j : j < e2 ? j - e1 : j < e3 ? j - e2 : j < e4 ? j - e3 : j - e4;
  const bool ok = base != nullptr;
  return ParamSrc{ok ? base + off * 4 : dummy, ok ? CS_LN + j * 4 : CS_PARAMS + lane * 4};
}

template <int NW, bool HAS_A, bool HAS_B, bool HAS_C, bool TRAIN, bool IL8, bool PIPE = false, bool BNP = false>
__global__ __launch_bounds__(NW * 64) void chain_s_kernel(ChainArgs a) {
  static_assert(!BNP || (TRAIN && HAS_A), "the batch-statistics prologue belongs to the training chain behind a conv module");
  static_assert(NW * 64 == 8 * BM, "row phases assume 8 lanes per panel row");
  static_assert(HAS_A || HAS_B, "at least one FeedForward");
  static_assert(!HAS_C || HAS_B, "the q | k | v stage follows the block's first FeedForward");
  constexpr int NT = 16 / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[PANEL_BYTES + YS_BYTES + (CS_PARAMS + CS_DUMP) * 4];
  unsigned char* const Xs = smem;
  unsigned char* const Hraw = smem + PANEL_BYTES;
  float* const Ps = reinterpret_cast<float*>(smem + PANEL_BYTES + YS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int r = tid >> 3, part = tid & 7;
  const int m0 = blockIdx.x * BM, row = m0 + r;   // whole panels only: row < M
  const int FF = a.FF, nchunk = FF / DM;
  float4 v[8], zr[BNP ? 8 : 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) {   // (BNP: the z rows now, the x2 rows behind the barrier - both at once were the kernel's register peak)
    if constexpr (BNP) zr[i] = *reinterpret_cast<const float4*>(a.bn_z + (size_t)row * DM + (i * 8 + part) * 4);
    else v[i] = *reinterpret_cast<const float4*>(a.x + (size_t)row * DM + (i * 8 + part) * 4);
  }
  // BNP: BatchNorm scale | shift of the 256 channels -> the (still unused) hidden-panel region, one float4 per thread of
  // waves 0-1, published by the parameter block's barrier below
  float4 ssv = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (BNP) ssv = *reinterpret_cast<const float4*>(a.bn_ss + (tid & 127) * 4);

  // ---- parameters -> LDS: float4 index space [0, 512) = the eight LayerNorm vectors (wave w stages vector w), then
  // b1 A (FF/4) | b2 A (64) | b1 B (FF/4) | b2 B (64) | b q|k|v (192), three more float4 per thread.  Missing stages and
  // folded LayerNorms read the residual rows instead (any valid address) and land in the dump block.
  const float* const dummy = a.x + (size_t)m0 * DM + lane * 4;
  // (named scalars and branch-free selects: an array written under the if-chain became an alloca, promoted into LDS)
  const int w8 = __builtin_amdgcn_readfirstlane(wave);
  // training: the four dropout keys first - each reads its seed word and WAITS for it (s_waitcnt vmcnt(0)); taken between
  // the stages they drained the weight ring and waited for the row stores in front of them
  const Drop dAh = TRAIN && HAS_A ? drop_make(a.da_h.seed, a.da_h.stream, a.da_h.p) : Drop{0u, 0u, 1.f};
  const Drop dAo = TRAIN && HAS_A ? drop_make(a.da_o.seed, a.da_o.stream, a.da_o.p) : Drop{0u, 0u, 1.f};
  const Drop dBh = TRAIN && HAS_B ? drop_make(a.db_h.seed, a.db_h.stream, a.db_h.p) : Drop{0u, 0u, 1.f};
  const Drop dBo = TRAIN && HAS_B ? drop_make(a.db_o.seed, a.db_o.stream, a.db_o.p) : Drop{0u, 0u, 1.f};
  const Drop dBn = BNP ? drop_make(a.d_bn.seed, a.d_bn.stream, a.d_bn.p) : Drop{0u, 0u, 1.f};
  const float* p0 = nullptr;
  if (HAS_A && w8 == 0) p0 = a.fa.ln_g;
  if (HAS_A && w8 == 1) p0 = a.fa.ln_b;
  if (HAS_A && w8 == 2) p0 = a.fin_g;
  if (HAS_A && w8 == 3) p0 = a.fin_b;
  if (HAS_B && w8 == 4) p0 = a.fb.ln_g;
  if (HAS_B && w8 == 5) p0 = a.fb.ln_b;
  if (HAS_C && w8 == 6) p0 = a.qln_g;
  if (HAS_C && w8 == 7) p0 = a.qln_b;
  const int pd0 = p0 != nullptr ? tid * 4 : CS_PARAMS + lane * 4;
  const float4 pv0 = *reinterpret_cast<const float4*>(p0 != nullptr ? p0 + lane * 4 : dummy);
  const ParamSrc ps1 = param_src<HAS_A, HAS_B, HAS_C>(a, tid, FF >> 2, dummy, lane);
  const ParamSrc ps2 = param_src<HAS_A, HAS_B, HAS_C>(a, tid + 512, FF >> 2, dummy, lane);
  const ParamSrc ps3 = param_src<HAS_A, HAS_B, HAS_C>(a, tid + 1024, FF >> 2, dummy, lane);
  const float4 pv1 = *reinterpret_cast<const float4*>(ps1.p);
  const float4 pv2 = *reinterpret_cast<const float4*>(ps2.p);
  const float4 pv3 = *reinterpret_cast<const float4*>(ps3.p);
  const float* const lnA = Ps, * const finA = Ps + 2 * DM, * const lnB = Ps + 4 * DM, * const lnQ = Ps + 6 * DM;
  const float* const b1A = Ps + CS_LN, * const b2A = b1A + FF, * const b1B = b2A + DM, * const b2B = b1B + FF, * const bQ = b2B + DM;

  // ---- the ONE weight ring of the kernel, primed with the first stage's first chunk
  const int rotA = blockIdx.x, rotB = blockIdx.x + 2;
  const bf16x8* const wA0 = HAS_A ? a.fa.w1 + (size_t)(wave * NT) * KS * 64 + lane + (size_t)(rotA % nchunk) * 16 * KS * 64 : nullptr;
  const bf16x8* const wB0 = HAS_B ? a.fb.w1 + (size_t)(wave * NT) * KS * 64 + lane + (size_t)(rotB % nchunk) * 16 * KS * 64 : nullptr;
  const bf16x8* const wC0 = HAS_C ? a.wqkv + (size_t)(wave * NT) * KS * 64 + lane : nullptr;
  bf16x8 wq[PD][NT];
  prefetch_first<NT>(wq, HAS_A ? wA0 : wB0, KS, NT);
  *reinterpret_cast<float4*>(Ps + pd0) = pv0;
  *reinterpret_cast<float4*>(Ps + ps1.dst) = pv1;
  *reinterpret_cast<float4*>(Ps + ps2.dst) = pv2;
  *reinterpret_cast<float4*>(Ps + ps3.dst) = pv3;
  if constexpr (BNP) {
    if (tid < 128) *reinterpret_cast<float4*>(Hraw + tid * 16) = ssv;
  }
  __syncthreads();
  if constexpr (BNP) {   // the conv module's tail on the register rows (bn_apply_kernel's arithmetic, element for element)
    const float pw = a.bn_pw2_w[0], pb = a.bn_pw2_b[0];
    const float* const ssl = reinterpret_cast<const float*>(Hraw);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(a.bn_x2 + (size_t)row * DM + (i * 8 + part) * 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = (i * 8 + part) * 4;
      const float4 sc = *reinterpret_cast<const float4*>(ssl + c), sh = *reinterpret_cast<const float4*>(ssl + DM + c);
      zr[i] = drop_apply4(dBn, (unsigned)((size_t)row * DM + c),
                          make_float4(fmaxf(zr[i].x * sc.x + sh.x, 0.f) * pw + pb, fmaxf(zr[i].y * sc.y + sh.y, 0.f) * pw + pb,
                                      fmaxf(zr[i].z * sc.z + sh.z, 0.f) * pw + pb, fmaxf(zr[i].w * sc.w + sh.w, 0.f) * pw + pb));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = make_float4(v[i].x + zr[i].x, v[i].y + zr[i].y, v[i].z + zr[i].z, v[i].w + zr[i].w);
      *reinterpret_cast<float4*>(a.bn_out + (size_t)row * DM + (i * 8 + part) * 4) = v[i];
    }
  }
#ifdef P_STAMPS
  unsigned long long* const g_stamps_dev = a.stamps;
#else
  unsigned long long* const g_stamps_dev = nullptr;
#endif
  P_T(0)
  if constexpr (HAS_A) {
    const StageTrain ta{dAh, dAo, m0, a.M, a.pre_ln_out};
    if constexpr (PIPE)
      ffn_stage_p<NW, true, TRAIN>(v, Xs, Hraw, a.fa.ln_g != nullptr ? lnA : nullptr, finA, b1A, b2A, a.fa, FF, a.alpha, tid, rotA, wq,
                                   HAS_B ? wB0 : wA0, g_stamps_dev, 1, ta,
                                   TRAIN ? (BNP ? a.bn_out : a.x) + (size_t)row * DM + part * 4 : nullptr);
    else
      ffn_stage_s<NW, true, TRAIN>(v, Xs, Hraw, a.fa.ln_g != nullptr ? lnA : nullptr, finA, b1A, b2A, a.fa, FF, a.alpha, tid, rotA, wq,
                                   HAS_B ? wB0 : wA0, g_stamps_dev, 1, ta);
    if constexpr (TRAIN) {  // the next block's input: the backward of its first FeedForward needs it
#pragma unroll
      for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(a.ln_out + (size_t)row * DM + (i * 8 + part) * 4) = v[i];
    }
  }
  P_T(4)
  if constexpr (HAS_B) {
    const StageTrain tb{dBh, dBo, m0, a.M, nullptr};
    if constexpr (PIPE)
      ffn_stage_p<NW, false, TRAIN>(v, Xs, Hraw, a.fb.ln_g != nullptr ? lnB : nullptr, nullptr, b1B, b2B, a.fb, FF, a.alpha, tid, rotB, wq,
                                    HAS_C ? wC0 : wB0, g_stamps_dev, 5, tb,
                                    TRAIN ? (HAS_A ? a.ln_out : a.x) + (size_t)row * DM + part * 4 : nullptr);
    else
      ffn_stage_s<NW, false, TRAIN>(v, Xs, Hraw, a.fb.ln_g != nullptr ? lnB : nullptr, nullptr, b1B, b2B, a.fb, FF, a.alpha, tid, rotB, wq,
                                    HAS_C ? wC0 : wB0, g_stamps_dev, 5, tb);
  }
  P_T(8)
#pragma unroll
  for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(a.xout + (size_t)row * DM + (i * 8 + part) * 4) = v[i];
  if constexpr (HAS_C) {
    rows_to_panel_s(Xs, v, a.qln_g != nullptr ? lnQ : nullptr, a.qln_g != nullptr ? lnQ + DM : nullptr, r, part);
    __syncthreads();
    P_T(9)
    const int t0 = wave * NT;
    qkv_pass_s<NT, true, IL8>(Xs, bQ, wC0, wq, a.qkv, m0, t0, n, g);
    qkv_pass_s<NT, true, IL8>(Xs, bQ, wC0 + (size_t)16 * KS * 64, wq, a.qkv, m0, 16 + t0, n, g);
    qkv_pass_s<NT, false, IL8>(Xs, bQ, wC0 + (size_t)32 * KS * 64, wq, a.qkv, m0, 32 + t0, n, g);
    P_T(10)
  }
}

// =============================================================================== chain, second form
// NOT the default (csship_encoder_chain runs it with CSSHIP_CHAIN=8 or 4): it is 8 % slower than the first form, and
// is kept as the measured counter-example to "the row phases and barriers are what is slow" - tools/chain2_stamps.py.
// What the stamps of both forms say instead: a 256-wide pass over a 64-row panel needs 128 KB of weight fragments
// through the CU's L1 (64 B/clk -> 2 048 cycles), 8 waves x 4 KB of LDS operand reads per k-step (128 B/clk -> 2 048
// cycles per pass) and 2 048 cycles of MFMA on each SIMD: all three at their peak at once, so the passes run at
// ~3 000 cycles whatever is done around them.  The accumulator-layout LayerNorm of this form costs MORE VALU than the
// row form (4 row reductions per lane instead of 1), and with 4 waves (half the LDS operand traffic) a single wave per
// SIMD cannot cover its own load latencies.
//
// Same operator chain, re-laid out after the phase stamps of the first form (tools/chain_stamps.py: of 93 600 cycles per
// workgroup 39 000 are MFMA; 23 000 sat in the row phases - park the accumulators in LDS, barrier, read rows back, LayerNorm,
// write the panel - and 18 000 in the hidden epilogues + barriers between the W1 and W2 passes):
//   * the residual rows never leave the ACCUMULATOR layout (lane = one token row per row tile, 4 consecutive features of each
//     of the wave's 2 feature tiles): the FFN output is added in registers, and LayerNorm needs only the per-row (mean, M2) of
//     the wave's 32 features (in-lane sums + 2 permlane swaps), an 8-byte-per-row exchange between the 8 waves through LDS
//     and Chan's combination - no fp32 staging tile, no row read-back;
//   * the W1 pass of chunk c+1 runs BEFORE the barrier that publishes the hidden panel of chunk c (three hidden panels in
//     rotation), so that barrier finds the stores long retired and every wave arriving together.  Weight streams therefore
//     run W1(c0) W1(c1) W2(c0) W1(c2) W2(c1) ... through the same prefetch ring.
constexpr int C2_STAT_BYTES = 8 * BM * 8;  // [<= 8 waves][64 rows] (mean, M2)

DEV uint2 relu_pack4(const f32x4& h) { return make_uint2(relu_bf2(pack_bf2(h[0], h[1])), relu_bf2(pack_bf2(h[2], h[3]))); }

// hidden epilogue: ReLU + bf16 pack of a wave's accumulators -> operand panel of the W2 pass
template <int NT>
DEV void store_hidden(unsigned char* hs, const f32x4 (&h)[NT][4], int hcol0, int n) {
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int hc = hcol0 + t * 16;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
      *reinterpret_cast<uint2*>(hs + panel_off(rt * 16 + n, hc >> 3) + (hc & 7) * 2) = relu_pack4(h[t][rt]);
  }
}

#ifdef P_STAMPS
#define P_T2(i) if (sbase >= 0) { P_T(sbase + (i)) }
#else
#define P_T2(i)
#endif
// per-row LayerNorm statistics of rows held in accumulator layout; contains ONE workgroup barrier
template <int NT>
DEV void acc_row_stats(const f32x4 (&v)[NT][4], float2* stat, int wave, int g, int n, float (&mean)[4], float (&rstd)[4],
                       unsigned long long* g_stamps_dev = nullptr, int sbase = -1) {
  P_T2(0)
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += (v[t][rt][0] + v[t][rt][1]) + (v[t][rt][2] + v[t][rt][3]);
    const float m = rows_sum4(s) * (1.f / (16 * NT));
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float d0 = v[t][rt][0] - m, d1 = v[t][rt][1] - m, d2 = v[t][rt][2] - m, d3 = v[t][rt][3] - m;
      q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    q = rows_sum4(q);
    if (g == 0) stat[wave * BM + rt * 16 + n] = make_float2(m, q);
  }
  P_T2(1)
  __syncthreads();
  P_T2(2)
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    constexpr int NW = 16 / NT;
    float2 e[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) e[w] = stat[w * BM + rt * 16 + n];
    float mu = 0.f, m2 = 0.f, dev = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { mu += e[w].x; m2 += e[w].y; }
    mu *= 1.f / NW;
#pragma unroll
    for (int w = 0; w < NW; ++w) dev += (e[w].x - mu) * (e[w].x - mu);
    mean[rt] = mu;
    rstd[rt] = rsqrtf((m2 + (16.f * NT) * dev) * (1.f / DM) + 1e-5f);
  }
  P_T2(3)
}

// LayerNorm of the accumulator-layout rows; PANEL: bf16 operand panel in LDS, else in place
template <bool PANEL, int NT>
DEV void acc_layernorm(f32x4 (&v)[NT][4], unsigned char* panel, float2* stat, const float4 (&gm)[NT], const float4 (&bt)[NT],
                       int wave, int g, int n, unsigned long long* g_stamps_dev = nullptr, int sbase = -1) {
  float mean[4], rstd[4];
  acc_row_stats<NT>(v, stat, wave, g, n, mean, rstd, g_stamps_dev, sbase);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = (wave * NT + t) * 16 + 4 * g;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const float y0 = (v[t][rt][0] - mean[rt]) * rstd[rt] * gm[t].x + bt[t].x;
      const float y1 = (v[t][rt][1] - mean[rt]) * rstd[rt] * gm[t].y + bt[t].y;
      const float y2 = (v[t][rt][2] - mean[rt]) * rstd[rt] * gm[t].z + bt[t].z;
      const float y3 = (v[t][rt][3] - mean[rt]) * rstd[rt] * gm[t].w + bt[t].w;
      if constexpr (PANEL)
        *reinterpret_cast<uint2*>(panel + panel_off(rt * 16 + n, col >> 3) + (col & 7) * 2) =
            make_uint2(pack_bf2(y0, y1), pack_bf2(y2, y3));
      else
        v[t][rt] = f32x4{y0, y1, y2, y3};
    }
  }
  P_T2(4)
}

template <int NT, bool FINAL_LN>
DEV void ffn2_stage(f32x4 (&v)[NT][4], unsigned char* Xs, unsigned char* Hs, float2* stat0, float2* stat1, const FfnW& f,
                    const float* fin_g, const float* fin_b, int FF, float alpha, int tid, int rot,
                    unsigned long long* g_stamps_dev = nullptr, int st0 = 0) {
  const int lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int nchunk = FF / DM, ks2 = FF / 32;
  const int fcol = wave * NT * 16 + 4 * g;  // this lane's first feature (tile t adds 16)
  const bf16x8* w1s = f.w1 + (size_t)(wave * NT) * KS * 64 + lane;
  const bf16x8* w2s = f.w2 + (size_t)(wave * NT) * ks2 * 64 + lane;
  int c = rot % nchunk;
  bf16x8 wq[PD][NT];
  prefetch_first<NT>(wq, w1s + (size_t)c * 16 * KS * 64, KS, NT);
  float4 gm[NT], bt[NT], bias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    gm[t] = *reinterpret_cast<const float4*>(f.ln_g + fcol + t * 16);
    bt[t] = *reinterpret_cast<const float4*>(f.ln_b + fcol + t * 16);
    bias[t] = *reinterpret_cast<const float4*>(f.b1 + c * DM + fcol + t * 16);
  }
  acc_layernorm<true, NT>(v, Xs, stat0, gm, bt, wave, g, n, g_stamps_dev, st0 == 4 ? 10 : -1);  // its barrier: every wave is past the previous stage's passes
  __syncthreads();
  P_T(st0 + 0)
  f32x4 o[NT][4], h[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      h[t][rt] = f32x4{bias[t].x, bias[t].y, bias[t].z, bias[t].w};  // the bias is the accumulators' initial value
    }
  {
    const int c1 = (c + 1) % nchunk;
    const bf16x8* nxt = nchunk > 1 ? w1s + (size_t)c1 * 16 * KS * 64 : w2s + (size_t)c * KS * 64;
    if (nchunk > 1) {
#pragma unroll
      for (int t = 0; t < NT; ++t) bias[t] = *reinterpret_cast<const float4*>(f.b1 + c1 * DM + fcol + t * 16);
    }
    mma_pass<NT>(h, Xs, n, g, NT, w1s + (size_t)c * 16 * KS * 64, KS, nxt, nchunk > 1 ? KS : ks2, NT, wq);
    store_hidden<NT>(Hs, h, fcol, n);
  }
#ifdef P_STAMPS
  unsigned long long tq_[5] = {0, 0, 0, 0, 0};
#define P_Q(i) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tq_[i])::"memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define P_Q(i)
#endif
  int slot = 0;  // hidden panel of chunk ci (three in rotation)
  for (int ci = 0; ci < nchunk; ++ci) {
    const int c1 = (c + 1) % nchunk, c2 = (c + 2) % nchunk;
    const int slot1 = slot == 2 ? 0 : slot + 1;
    const bf16x8* w2c = w2s + (size_t)c * KS * 64;
    if (ci + 1 < nchunk) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{bias[t].x, bias[t].y, bias[t].z, bias[t].w};
      if (ci + 2 < nchunk) {
#pragma unroll
        for (int t = 0; t < NT; ++t) bias[t] = *reinterpret_cast<const float4*>(f.b1 + c2 * DM + fcol + t * 16);
      }
      P_Q(0)
      mma_pass<NT>(h, Xs, n, g, NT, w1s + (size_t)c1 * 16 * KS * 64, KS, w2c, ks2, NT, wq);
      P_Q(1)
      // panel slot1 was last read by the W2 pass of chunk ci-2: every wave finished that before the barrier of ci-1
      store_hidden<NT>(Hs + slot1 * PANEL_BYTES, h, fcol, n);
    }
    P_Q(2)
    __syncthreads();
    P_Q(3)
    const bf16x8* nxt = w2c;
    int nstride = ks2, nnt = 0;
    if (ci + 2 < nchunk) { nxt = w1s + (size_t)c2 * 16 * KS * 64; nstride = KS; nnt = NT; }
    else if (ci + 1 < nchunk) { nxt = w2s + (size_t)c1 * KS * 64; nnt = NT; }
    mma_pass<NT>(o, Hs + slot * PANEL_BYTES, n, g, NT, w2c, ks2, nxt, nstride, nnt, wq);
    P_Q(4)
#ifdef P_STAMPS
    if (st0 == 1 && ci == 1 && g_stamps_dev != nullptr && (threadIdx.x & 63) == 0 && (wave == 0 || wave == 7)) {
      unsigned long long* d_ = g_stamps_dev + (size_t)blockIdx.x * 32 + 16 + (wave == 7 ? 5 : 0);
      for (int i_ = 0; i_ < 5; ++i_) d_[i_] = tq_[i_];
    }
#endif
    c = c1;
    slot = slot1;
  }
  P_T(st0 + 1)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 b2 = *reinterpret_cast<const float4*>(f.b2 + fcol + t * 16);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      v[t][rt][0] += alpha * (o[t][rt][0] + b2.x); v[t][rt][1] += alpha * (o[t][rt][1] + b2.y);
      v[t][rt][2] += alpha * (o[t][rt][2] + b2.z); v[t][rt][3] += alpha * (o[t][rt][3] + b2.w);
    }
  }
  if constexpr (FINAL_LN) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      gm[t] = *reinterpret_cast<const float4*>(fin_g + fcol + t * 16);
      bt[t] = *reinterpret_cast<const float4*>(fin_b + fcol + t * 16);
    }
    acc_layernorm<false, NT>(v, nullptr, stat1, gm, bt, wave, g, n);
  }
  P_T(st0 + 2)
}

template <int NW, bool HAS_A, bool HAS_B, bool HAS_C>
__global__ __launch_bounds__(NW * 64) void chain2_kernel(ChainArgs a) {
  constexpr int NT = 16 / NW;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * PANEL_BYTES + 2 * C2_STAT_BYTES];
  unsigned char* const Xs = smem;
  unsigned char* const Hs = smem + PANEL_BYTES;
  float2* const stat0 = reinterpret_cast<float2*>(smem + 4 * PANEL_BYTES);
  float2* const stat1 = stat0 + 8 * BM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM, fcol = wave * NT * 16 + 4 * g;
  f32x4 v[NT][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const float* xr = a.x + (size_t)min(m0 + rt * 16 + n, a.M - 1) * DM + fcol;
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t][rt] = *reinterpret_cast<const f32x4*>(xr + t * 16);
  }
#ifdef P_STAMPS
  unsigned long long* const g_stamps_dev = a.stamps;
#endif
  P_T(0)
  if constexpr (HAS_A)
    ffn2_stage<NT, true>(v, Xs, Hs, stat0, stat1, a.fa, a.fin_g, a.fin_b, a.FF, a.alpha, tid, blockIdx.x, a.stamps, 1);
  if constexpr (HAS_B)
    ffn2_stage<NT, false>(v, Xs, Hs, stat0, stat1, a.fb, nullptr, nullptr, a.FF, a.alpha, tid, blockIdx.x + 2, a.stamps, 4);
  if constexpr (HAS_C) {
    const int ntiles = (a.NQKV + 15) / 16, npass = (ntiles + 15) / 16;
    bf16x8 wq[PD][NT];
    const bf16x8* wbase = a.wqkv + lane;
    prefetch_first<NT>(wq, wbase + (size_t)(wave * NT) * KS * 64, KS, max(0, min(NT, ntiles - wave * NT)));
    float4 gm[NT], bt[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      gm[t] = *reinterpret_cast<const float4*>(a.qln_g + fcol + t * 16);
      bt[t] = *reinterpret_cast<const float4*>(a.qln_b + fcol + t * 16);
    }
    acc_layernorm<true, NT>(v, Xs, stat0, gm, bt, wave, g, n);
    __syncthreads();
    P_T(7)
    for (int p = 0; p < npass; ++p) {
      const int tile0 = p * 16 + wave * NT;
      const int nt = max(0, min(NT, ntiles - tile0));
      const int nnt = max(0, min(NT, ntiles - (tile0 + 16)));
      f32x4 acc[NT][4];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 bb = load_bias4(a.bqkv, (tile0 + t) * 16 + 4 * g, a.NQKV);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
      }
      const bf16x8* wc = wbase + (size_t)tile0 * KS * 64;
      mma_pass<NT>(acc, Xs, n, g, nt, wc, KS, wc + (size_t)16 * KS * 64, KS, nnt, wq);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int f = (tile0 + t) * 16 + 4 * g;
        if (t < nt && f + 3 < a.NQKV) {  // NQKV % 8 == 0 (checked on the host)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) {
            const int orow = m0 + rt * 16 + n;
            if (orow < a.M)
              *reinterpret_cast<uint2*>(a.qkv + (size_t)orow * a.NQKV + f) =
                  make_uint2(pack_bf2(acc[t][rt][0], acc[t][rt][1]), pack_bf2(acc[t][rt][2], acc[t][rt][3]));
          }
        }
      }
    }
    P_T(8)
  }
  // the residual stream the attention output projection adds to (last: nothing in this kernel waits behind these stores)
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int orow = m0 + rt * 16 + n;
    if (orow < a.M) {
#pragma unroll
      for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(a.xout + (size_t)orow * DM + fcol + t * 16) = v[t][rt];
    }
  }
}

// =============================================================================== embed
// x0 = ReLU(LayerNorm(Linear(MVN_time(f))))   (css/models/conformer.py:95 + :124-129)
// f: [B,T,idim] f32 with arbitrary idim (257, 1799): the input MVN is applied while the panel is built,
// K is walked in 256-column chunks (weights packed with K zero-padded to a multiple of 256).
struct EmbedArgs {
  const float* f;
  const float *mean, *rinv;  // [B, idim]
  const bf16x8* w;           // packed [16][Kp/32][64]
  const float* bias;
  const float *ln_g, *ln_b;
  float* out;
  int M, T, idim, Kp;
  int64_t ldf;
  // TRAIN (csship_embed_train): what the backward keeps - the normalised features as the bf16 operand of the weight gradient
  // (fn_out [M, ldfn], zero beyond idim) and the rows in front of the LayerNorm (e_out f32 [M, 256]) - and the embedding's
  // dropout (conformer.py:127; relu(drop(v)) == drop(relu(v)): the scale is positive)
  unsigned short* fn_out;
  int ldfn;
  float* e_out;
  csship_dropout_cfg drop;
};

template <int NW, bool TRAIN = false>
__global__ __launch_bounds__(NW * 64) void embed_kernel(EmbedArgs a) {
  static_assert(NW * 64 == 8 * BM, "row phases assume 8 lanes per panel row");
  constexpr int NT = 16 / NW;
  __shared__ __attribute__((aligned(16))) unsigned char Xs[PANEL_BYTES];
  __shared__ __attribute__((aligned(16))) float Ys[BM * YS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int r = tid >> 3, part = tid & 7;
  const int m0 = blockIdx.x * BM, row = m0 + r, rowc = min(row, a.M - 1);
  const int nkc = a.Kp / DM, kst = a.Kp / 32;
  const float* fr = a.f + (size_t)rowc * a.ldf;
  const size_t sb = (size_t)(rowc / a.T) * a.idim;
  const bf16x8* wbase = a.w + (size_t)(wave * NT) * kst * 64 + lane;
  bf16x8 wq[PD][NT];
  prefetch_first<NT>(wq, wbase, kst, NT);
  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int kc = 0; kc < nkc; ++kc) {
    if (kc > 0) __syncthreads();  // the previous chunk's panel has been consumed
    // rows of f are only 4-byte aligned (idim = 257, 1799): ONE 16-byte load per operand and four columns (unaligned
    // dwordx4 is fine, and the memory pipeline takes the same ~18 cycles per wave-wide load whatever its width); column
    // groups that lie wholly beyond idim (all of the last chunk but its first columns) are zero without any load
    struct __attribute__((packed, aligned(4))) f4u { float v[4]; };
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = i * 8 + part, k = kc * DM + c4 * 4;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (kc * DM + i * 32 < a.idim) {  // wave-uniform
        if (k + 3 < a.idim) {
          const f4u fv = *reinterpret_cast<const f4u*>(fr + k), mv = *reinterpret_cast<const f4u*>(a.mean + sb + k),
                    rv = *reinterpret_cast<const f4u*>(a.rinv + sb + k);
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = (fv.v[q] - mv.v[q]) * rv.v[q];
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) e[q] = (k + q < a.idim) ? (fr[k + q] - a.mean[sb + k + q]) * a.rinv[sb + k + q] : 0.f;
        }
      }
      const uint2 pk = make_uint2(pack_bf2(e[0], e[1]), pack_bf2(e[2], e[3]));
      *reinterpret_cast<uint2*>(Xs + panel_off(r, c4 >> 1) + (c4 & 1) * 8) = pk;
      if constexpr (TRAIN) {
        if (row < a.M && k < a.ldfn) *reinterpret_cast<uint2*>(a.fn_out + (size_t)row * a.ldfn + k) = pk;
      }
    }
    __syncthreads();
    const bf16x8* wc = wbase + (size_t)kc * KS * 64;
    mma_pass<NT>(acc, Xs, n, g, NT, wc, kst, (kc + 1 < nkc) ? wc + (size_t)KS * 64 : wbase, kst, NT, wq);
  }
  float4 bb[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) bb[t] = *reinterpret_cast<const float4*>(a.bias + (wave * NT + t) * 16 + 4 * g);
  park_tiles<NT, false>(Ys, acc, NT, wave * NT * 16, g, n, bb);
  __syncthreads();
  float4 v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = *reinterpret_cast<const float4*>(Ys + r * YS + (i * 8 + part) * 4);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    if constexpr (TRAIN) {
      if (row < a.M) *reinterpret_cast<float4*>(a.e_out + (size_t)row * DM + (i * 8 + part) * 4) = v[i];
    }
  }
  const float mean = sum8(s) * (1.f / DM);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
  const Drop dp = TRAIN ? drop_make(a.drop.seed, a.drop.stream, a.drop.p) : Drop{0u, 0u, 1.f, 0x80008000u};
  if (row < a.M) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = (i * 8 + part) * 4;
      const float4 gm = *reinterpret_cast<const float4*>(a.ln_g + c);
      const float4 bt = *reinterpret_cast<const float4*>(a.ln_b + c);
      float4 y = make_float4(fmaxf(v[i].x * rstd * gm.x + bt.x, 0.f), fmaxf(v[i].y * rstd * gm.y + bt.y, 0.f),
                             fmaxf(v[i].z * rstd * gm.z + bt.z, 0.f), fmaxf(v[i].w * rstd * gm.w + bt.w, 0.f));
      if constexpr (TRAIN) y = drop_apply4(dp, (unsigned)((size_t)row * DM + c), y);
      *reinterpret_cast<float4*>(a.out + (size_t)row * DM + c) = y;
    }
  }
}

// =============================================================================== out-proj + conv module, one kernel
// x2 = x1 + attn_out Wo^T + bo (conformer.py:254,356) followed by the convolution module + residual
// (conformer.py:289-303,357), eval-mode BatchNorm.  One workgroup = (batch b, 50 output frames): it projects the 82
// frames of tile + halo (96-row bf16 panel, 6 row tiles per wave: the halo is recomputed, 1.6x of a 13 MFLOP GEMM),
// finishes the rows through the staged tile (residual -> x2 rows of the own frames go to `out`; LayerNorm -> scalar
// GLU written back IN PLACE, so the staging tile becomes the stencil's input), then two threads per channel run the
// 33-tap depthwise convolution, BatchNorm, ReLU, affine and add onto x2 in `out`.  8 tiles x 32 chunks = 256
// workgroups at T = 400: exactly one round of the chip, one launch per block instead of two, x2 never makes a round
// trip through a separate buffer.  `out` must not alias x1 (neighbouring workgroups read each other's halo rows).
#ifndef OC_RES_AT
#define OC_RES_AT 4   // k-step behind which the residual rows are requested (KS - PD: behind the last weight fragment); -1: behind the loop
#endif
constexpr int OC_TT = 50, OC_HALO = 16, OC_USED = OC_TT + 2 * OC_HALO, OC_RT = 6, OC_ROWS = 16 * OC_RT, OC_KS = 33;
static_assert(OC_USED <= OC_ROWS, "tile + halo must fit the row tiles");

struct OcArgs {
  const unsigned short* ao;  // attention output bf16 [B*T, 256]
  const bf16x8* wo;          // packed [16][8][64]
  const float* bo;
  const float* x1;           // residual stream in  [B*T, 256]
  float* out;                // residual stream out [B*T, 256]
  csship_conv_weights cw;
  int B, T;
  unsigned long long* stamps;
  // TRAIN (csship_outproj_conv_train): the reference's dropout of the projection (conformer.py:254; element index
  // row * 256 + column as csship_panel_linear_drop), `out` receives x2 itself, the depthwise convolution's output goes
  // to zbuf [B*T, 256] and its per-channel sum / sum of squares per (workgroup, half tile) to partial [nblk][2][256]
  // (nblk = 2 * gridDim.x * B): the batch-statistics BatchNorm is finished by bn_reduce8 + bn_apply (convmod.hip)
  csship_dropout_cfg drop;
  float* zbuf;
  float* partial;
};

template <int NW, bool TRAIN = false>
__global__ __launch_bounds__(NW * 64) void outproj_conv_kernel(OcArgs a) {
  static_assert(NW == 8, "row phases: 8 lanes per row, 64 rows per pass; stencil: 2 threads per channel");
  constexpr int NT = 16 / NW, NSTAGE = OC_ROWS * 32 / (NW * 64);
  // Xs: the bf16 operand panel of the projection (96 rows x 512 B), and once the GEMM is done the f32 tile of the
  // workgroup's OWN x2 rows (50 x 1 KB): x2 stays on the CU from the row phase to the final store (it used to be
  // written to `out`, read back by the stencil and written again: the kernel is bound by the rate at which the CU's
  // texture-address unit takes wave memory instructions, ~170 per wave then, ~80 now)
  constexpr int XS_BYTES = OC_TT * DM * 4 > OC_ROWS * 512 ? OC_TT * DM * 4 : OC_ROWS * 512;
  __shared__ __attribute__((aligned(16))) unsigned char Xs[XS_BYTES];
  __shared__ __attribute__((aligned(16))) float Ys[OC_ROWS * YS];
  __shared__ __attribute__((aligned(16))) float Ps[2 * DM];  // conv-module LayerNorm gamma | beta, staged once
  float* const X2 = reinterpret_cast<float*>(Xs);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int b = blockIdx.y, t0 = blockIdx.x * OC_TT;
  const size_t rowb = (size_t)b * a.T;
  const bf16x8* ws_ = a.wo + (size_t)(wave * NT) * KS * 64 + lane;
  bf16x8 wq[PD][NT];
  stamp(a.stamps, 0);
  prefetch_first<NT>(wq, ws_, KS, NT);
  if (tid < 2 * DM / 4)
    *reinterpret_cast<float4*>(Ps + tid * 4) = *reinterpret_cast<const float4*>((tid < DM / 4 ? a.cw.ln_g : a.cw.ln_b - DM) + tid * 4);

  // ---- attention-output rows of tile + halo -> bf16 panel (rows outside the utterance / beyond row 81: zero).
  // The residual rows of the row phase are requested INSIDE the projection's k loop, behind the wave's last weight
  // fragment: a wave's loads return in order, and requested ahead of the loop (rounds 2-3: "they land during the GEMM")
  // they held every weight fragment back until the 21 MB of residual rows of all workgroups were through HBM - the GEMM
  // did not overlap them, it waited for them (16.1 -> 15.4 us per launch).
  uint4 av[NSTAGE];
#pragma unroll
  for (int k = 0; k < NSTAGE; ++k) {
    const int i = tid + k * NW * 64;
    const int r = i >> 5, c = i & 31, t = t0 - OC_HALO + r;
    av[k] = make_uint4(0, 0, 0, 0);
    if (r < OC_USED && t >= 0 && t < a.T) av[k] = *reinterpret_cast<const uint4*>(a.ao + (rowb + t) * DM + c * 8);
  }
  __builtin_amdgcn_sched_barrier(0);
  float4 xres[2][8];  // two passes of 64 rows, 8 lanes per row
  auto request_residual = [&]() {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = pass * 64 + (tid >> 3), t = t0 - OC_HALO + r;
      const bool ok = r < OC_USED && t >= 0 && t < a.T;
      const float* xr = a.x1 + (rowb + (ok ? t : 0)) * DM;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        xres[pass][i] = ok ? *reinterpret_cast<const float4*>(xr + (i * 8 + (tid & 7)) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
#pragma unroll
  for (int k = 0; k < NSTAGE; ++k) {
    const int i = tid + k * NW * 64;
    *reinterpret_cast<uint4*>(Xs + panel_off(i >> 5, i & 31)) = av[k];
  }
  __syncthreads();
  stamp(a.stamps, 1);

  // ---- out-proj: acc[t][rt] = Wo(tile)[16 x 256] . panel^T, weights streamed PD k-steps ahead
  f32x4 acc[NT][OC_RT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < OC_RT; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    bf16x8 xa[2][OC_RT];
#pragma unroll
    for (int rt = 0; rt < OC_RT; ++rt) xa[0][rt] = *reinterpret_cast<const bf16x8*>(Xs + panel_off(rt * 16 + n, g));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 wc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wc[t] = wq[ks % PD][t];
      if (ks + PD < KS) {
#pragma unroll
        for (int t = 0; t < NT; ++t) wq[ks % PD][t] = ws_[(size_t)(t * KS + ks + PD) * 64];
      }
      if (OC_RES_AT >= 0 && ks == OC_RES_AT) request_residual();   // behind the last weight fragment: a wave's loads return in order
      if (ks + 1 < KS) {
#pragma unroll
        for (int rt = 0; rt < OC_RT; ++rt)
          xa[(ks + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(Xs + panel_off(rt * 16 + n, (ks + 1) * 4 + g));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < OC_RT; ++rt)
          acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], xa[ks & 1][rt], acc[t][rt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (OC_RES_AT < 0) request_residual();
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = (wave * NT + t) * 16 + 4 * g;
    const float4 bb = *reinterpret_cast<const float4*>(a.bo + col);
#pragma unroll
    for (int rt = 0; rt < OC_RT; ++rt)
      *reinterpret_cast<float4*>(Ys + (rt * 16 + n) * YS + col) =
          make_float4(acc[t][rt][0] + bb.x, acc[t][rt][1] + bb.y, acc[t][rt][2] + bb.z, acc[t][rt][3] + bb.w);
  }
  // stencil constants of this thread's channel: requested now, they arrive during the row phase
  float wk[OC_KS];
  {
    const int c = tid & (DM - 1);
#pragma unroll
    for (int k = 0; k < OC_KS; ++k) wk[k] = a.cw.dw_w[k * DM + c];
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  stamp(a.stamps, 2);

  // ---- rows: x2 = x1 + proj; own frames -> out; g = GLU(LN(x2)) back into the staging tile (zero outside the utterance)
  {
    const float w0 = a.cw.pw1_w[0], w1 = a.cw.pw1_w[1], b0 = a.cw.pw1_b[0], b1 = a.cw.pw1_b[1];
    const int part = tid & 7;
    const Drop dproj = TRAIN ? drop_make(a.drop.seed, a.drop.stream, a.drop.p) : Drop{0u, 0u, 1.f};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = pass * 64 + (tid >> 3);
      if (r >= OC_USED) continue;
      const int t = t0 - OC_HALO + r;
      float* yrow = Ys + r * YS;
      if (t < 0 || t >= a.T) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<float4*>(yrow + (i * 8 + part) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      const bool own = r >= OC_HALO && r < OC_HALO + OC_TT;
      float4 v[8];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 xv = xres[pass][i];
        float4 pv = *reinterpret_cast<const float4*>(yrow + (i * 8 + part) * 4);
        if constexpr (TRAIN) pv = drop_apply4(dproj, (unsigned)((rowb + t) * DM + (i * 8 + part) * 4), pv);  // halo rows: the same mask as their owner
        v[i] = make_float4(xv.x + pv.x, xv.y + pv.y, xv.z + pv.z, xv.w + pv.w);
        if (own) *reinterpret_cast<float4*>(X2 + (r - OC_HALO) * DM + (i * 8 + part) * 4) = v[i];
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      }
      const float mean = sum8(s) * (1.f / DM);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
      }
      const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = (i * 8 + part) * 4;
        const float4 gm = *reinterpret_cast<const float4*>(Ps + c);
        const float4 bt = *reinterpret_cast<const float4*>(Ps + DM + c);
        float e[4] = {v[i].x * rstd * gm.x + bt.x, v[i].y * rstd * gm.y + bt.y, v[i].z * rstd * gm.z + bt.z,
                      v[i].w * rstd * gm.w + bt.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) e[q] = (w0 * e[q] + b0) * __builtin_amdgcn_rcpf(1.f + __expf(-(w1 * e[q] + b1)));
        *reinterpret_cast<float4*>(yrow + c) = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  }
  __syncthreads();
  stamp(a.stamps, 3);

  // ---- depthwise conv + BatchNorm (running statistics) + ReLU + affine, added onto x2.
  // Two outputs per packed FMA: o2[p] = {out[2p], out[2p+1]} += {w[m], w[m-1]} * g[u] with m = u - 2p; the thread's
  // 57 staging values are read once into registers (no re-reads, no wait per tap).
  {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int c = tid & (DM - 1), half = tid >> 8;
    constexpr int TH = OC_TT / 2, NP = (TH + 1) / 2, NG = TH + OC_KS - 1;
    f2 wp[OC_KS + 1];  // wp[m] = {w[m], w[m-1]}, zero outside 0..32
#pragma unroll
    for (int m = 0; m <= OC_KS; ++m) wp[m] = f2{m < OC_KS ? wk[m] : 0.f, m > 0 ? wk[m - 1] : 0.f};
    const float cb = a.cw.dw_b[c];
    float scale = 0.f, shift = 0.f, pw2w = 0.f, pw2b = 0.f;
    if constexpr (!TRAIN) {
      scale = a.cw.bn_g[c] * rsqrtf(a.cw.bn_var[c] + 1e-5f);
      shift = a.cw.bn_b[c] - a.cw.bn_mean[c] * scale;
      pw2w = a.cw.pw2_w[0];
      pw2b = a.cw.pw2_b[0];
    }
    float gv[NG];
#pragma unroll
    for (int u = 0; u < NG; ++u) gv[u] = Ys[(half * TH + u) * YS + c];
    f2 o2[NP];
#pragma unroll
    for (int p_ = 0; p_ < NP; ++p_) o2[p_] = f2{cb, cb};
#pragma unroll
    for (int u = 0; u < NG; ++u)
#pragma unroll
      for (int p_ = 0; p_ < NP; ++p_) {
        const int m = u - 2 * p_;
        if (m >= 0 && m <= OC_KS) o2[p_] = __builtin_elementwise_fma(wp[m], f2{gv[u], gv[u]}, o2[p_]);
      }
    if constexpr (TRAIN) {
      // batch statistics: the pre-BatchNorm activations leave as they are (lanes = consecutive channels), with this
      // thread's share of the channel's sum and sum of squares; x2 stays untouched
      float lsum = 0.f, lsq = 0.f;
#pragma unroll
      for (int j = 0; j < TH; ++j) {
        const float v = (j & 1) ? o2[j >> 1].y : o2[j >> 1].x;
        const int t = t0 + half * TH + j;
        if (t < a.T) {
          a.zbuf[(rowb + t) * DM + c] = v;
          lsum += v;
          lsq += v * v;
        }
      }
      const size_t blk = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + half;
      a.partial[(blk * 2 + 0) * DM + c] = lsum;
      a.partial[(blk * 2 + 1) * DM + c] = lsq;
    } else {
    // the branch is added onto the x2 tile in place (this thread is the only one that touches these elements) ...
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const float v = (j & 1) ? o2[j >> 1].y : o2[j >> 1].x;
      X2[(half * TH + j) * DM + c] += fmaxf(v * scale + shift, 0.f) * pw2w + pw2b;
    }
    }
  }
  __syncthreads();
  // ... and the finished rows leave as whole 16-byte pieces, 64 lanes = one row (rows beyond the utterance: not stored)
  for (int i = tid; i < OC_TT * (DM / 4); i += NW * 64) {
    const int r = i / (DM / 4), c4 = i % (DM / 4), t = t0 + r;
    if (t < a.T) *reinterpret_cast<float4*>(a.out + (rowb + t) * DM + c4 * 4) = *reinterpret_cast<const float4*>(X2 + r * DM + c4 * 4);
  }
  stamp(a.stamps, 4);
}

constexpr int NWAVES = 8;

template <int PRO, int EPI>
int launch_linear(const LinArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((panel_linear_kernel<NWAVES, PRO, EPI>), dim3((a.M + BM - 1) / BM), dim3(NWAVES * 64), 0, s, a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

}  // namespace

extern "C" int csship_ffn_fused_drop(const float* x, float* out, const float* ln_g, const float* ln_b,
                                     const uint16_t* w1_packed, const float* b1, const uint16_t* w2_packed,
                                     const float* b2, const float* final_ln_g, const float* final_ln_b, int32_t M,
                                     int32_t d_model, int32_t d_ffn, float alpha, const csship_dropout_cfg* drop_hidden,
                                     const csship_dropout_cfg* drop_out, csship_stream_t stream) {
  return csship_ffn_fused_train(x, out, nullptr, ln_g, ln_b, w1_packed, b1, w2_packed, b2, final_ln_g, final_ln_b, M, d_model,
                                d_ffn, alpha, drop_hidden, drop_out, stream);
}

extern "C" int csship_ffn_fused_train(const float* x, float* out, float* pre_ln_out, const float* ln_g, const float* ln_b,
                                      const uint16_t* w1_packed, const float* b1, const uint16_t* w2_packed,
                                      const float* b2, const float* final_ln_g, const float* final_ln_b, int32_t M,
                                      int32_t d_model, int32_t d_ffn, float alpha, const csship_dropout_cfg* drop_hidden,
                                      const csship_dropout_cfg* drop_out, csship_stream_t stream) {
  if (!x || !out || !ln_g || !ln_b || !w1_packed || !b1 || !w2_packed || !b2) return CSSHIP_ENULL;
  if (pre_ln_out && (!final_ln_g || !aligned16(pre_ln_out) || pre_ln_out == out)) return CSSHIP_EBADSHAPE;
  if ((final_ln_g == nullptr) != (final_ln_b == nullptr)) return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (d_model != DM || d_ffn <= 0 || (d_ffn % DM) != 0) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(out) || !aligned16(w1_packed) || !aligned16(w2_packed) || !aligned16(ln_g) ||
      !aligned16(ln_b) || !aligned16(b1) || !aligned16(b2) || (final_ln_g && (!aligned16(final_ln_g) || !aligned16(final_ln_b))))
    return CSSHIP_EALIGN;
  if ((int64_t)((M + BM - 1) / BM * BM) * d_ffn > 0xffffffffLL) return CSSHIP_EBADSHAPE;  // 32-bit dropout element indices
  const csship_dropout_cfg none{nullptr, 0u, 0.f};
  FfnArgs a{x, out, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w1_packed), b1,
            reinterpret_cast<const bf16x8*>(w2_packed), b2, final_ln_g, final_ln_b, M, d_ffn, alpha, g_csship_stamps,
            drop_hidden ? *drop_hidden : none, drop_out ? *drop_out : none, pre_ln_out};
  dim3 grid((M + BM - 1) / BM), block(NWAVES * 64);
  TimelineScope tl(CSSHIP_TL_CHAIN_EDGE, as_stream(stream));   // (the last block's FeedForward of the training forward)
  if (final_ln_g) hipLaunchKernelGGL((ffn_kernel<NWAVES, true>), grid, block, 0, as_stream(stream), a);
  else hipLaunchKernelGGL((ffn_kernel<NWAVES, false>), grid, block, 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_ffn_fused(const float* x, float* out, const float* ln_g, const float* ln_b,
                                const uint16_t* w1_packed, const float* b1, const uint16_t* w2_packed,
                                const float* b2, const float* final_ln_g, const float* final_ln_b, int32_t M,
                                int32_t d_model, int32_t d_ffn, float alpha, csship_stream_t stream) {
  return csship_ffn_fused_drop(x, out, ln_g, ln_b, w1_packed, b1, w2_packed, b2, final_ln_g, final_ln_b, M, d_model, d_ffn,
                               alpha, nullptr, nullptr, stream);
}

extern "C" int csship_outproj_conv(const uint16_t* attn_out, const uint16_t* wo_packed, const float* bo, const float* x,
                                   float* out, const csship_conv_weights* conv, int32_t B, int32_t T, int32_t d_model,
                                   int32_t kernel_size, csship_stream_t stream) {
  if (!attn_out || !wo_packed || !bo || !x || !out || !conv) return CSSHIP_ENULL;
  if (conv->causal) return CSSHIP_EUNSUPPORTED;  // the fused form is built for the centred stencil (csship_convmod serves the causal one)
  if (!conv->ln_g || !conv->ln_b || !conv->pw1_w || !conv->pw1_b || !conv->dw_w || !conv->dw_b || !conv->bn_g ||
      !conv->bn_b || !conv->bn_mean || !conv->bn_var || !conv->pw2_w || !conv->pw2_b)
    return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  if (d_model != DM || kernel_size != OC_KS) return CSSHIP_EUNSUPPORTED;
  if (x == out) return CSSHIP_EBADSHAPE;  // halo rows of neighbouring tiles are read from x
  if (!aligned16(attn_out) || !aligned16(wo_packed) || !aligned16(bo) || !aligned16(x) || !aligned16(out) ||
      !aligned16(conv->ln_g) || !aligned16(conv->ln_b))
    return CSSHIP_EALIGN;
  OcArgs a{attn_out, reinterpret_cast<const bf16x8*>(wo_packed), bo, x, out, *conv, B, T, g_csship_stamps,
           csship_dropout_cfg{nullptr, 0u, 0.f}, nullptr, nullptr};
  hipLaunchKernelGGL((outproj_conv_kernel<NWAVES>), dim3((T + OC_TT - 1) / OC_TT, B), dim3(NWAVES * 64), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

// the TRAIN form's launch, for csship_outproj_conv_train (convmod.hip owns the BatchNorm kernels that finish it)
int launch_outproj_conv_train(const uint16_t* attn_out, const uint16_t* wo_packed, const float* bo, const float* x,
                              float* x2_out, const csship_conv_weights& conv, int B, int T, csship_dropout_cfg drop,
                              float* zbuf, float* partial, int* nblk_out, hipStream_t s) {
  OcArgs a{attn_out, reinterpret_cast<const bf16x8*>(wo_packed), bo, x, x2_out, conv, B, T, g_csship_stamps, drop, zbuf, partial};
  const dim3 grid((T + OC_TT - 1) / OC_TT, B);
  *nblk_out = (int)(grid.x * grid.y * 2);
  hipLaunchKernelGGL((outproj_conv_kernel<NWAVES, true>), grid, dim3(NWAVES * 64), 0, s, a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" size_t csship_ffn_bwd_workspace(int32_t M) { return (size_t)((M + BM - 1) / BM) * 2 * DM * sizeof(float); }

extern "C" int csship_ffn_bwd_drop(const float* x, const float* dy, const float* ln_g, const float* ln_b,
                                   const uint16_t* w1_packed, const float* b1, const uint16_t* w2t_packed,
                                   const uint16_t* w1t_packed, uint16_t* a_bf16, uint16_t* h_bf16, uint16_t* dh_bf16,
                                   float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                                   const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out,
                                   uint16_t* dyd_bf16, void* workspace, size_t workspace_bytes, csship_stream_t stream) {
  return csship_ffn_bwd_ln(x, dy, ln_g, ln_b, w1_packed, b1, w2t_packed, w1t_packed, a_bf16, h_bf16, dh_bf16, dx, dgamma_dbeta, M,
                           d_model, d_ffn, alpha, drop_hidden, drop_out, dyd_bf16, nullptr, nullptr, nullptr, workspace,
                           workspace_bytes, stream);
}

extern "C" size_t csship_ffn_bwd_ln_workspace(int32_t M) { return 2 * csship_ffn_bwd_workspace(M); }

extern "C" int csship_ffn_bwd_ln(const float* x, const float* dy, const float* ln_g, const float* ln_b,
                                 const uint16_t* w1_packed, const float* b1, const uint16_t* w2t_packed,
                                 const uint16_t* w1t_packed, uint16_t* a_bf16, uint16_t* h_bf16, uint16_t* dh_bf16,
                                 float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                                 const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out,
                                 uint16_t* dyd_bf16, const float* ln2_x, const float* ln2_g, float* dgamma_dbeta2,
                                 void* workspace, size_t workspace_bytes, csship_stream_t stream) {
  if (!x || !dy || !ln_g || !ln_b || !w1_packed || !b1 || !w2t_packed || !w1t_packed || !a_bf16 || !h_bf16 || !dh_bf16 ||
      !dx || !dgamma_dbeta || !workspace)
    return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (d_model != DM || d_ffn <= 0 || (d_ffn % DM) != 0) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || !aligned16(ln_g) || !aligned16(ln_b) || !aligned16(b1) ||
      !aligned16(w1_packed) || !aligned16(w2t_packed) || !aligned16(w1t_packed) || !aligned8(a_bf16) || !aligned8(h_bf16) ||
      !aligned8(dh_bf16))
    return CSSHIP_EALIGN;
  const int nwg = (M + BM - 1) / BM;
  if (workspace_bytes < (size_t)nwg * 2 * DM * sizeof(float)) return CSSHIP_EWORKSPACE;
  if ((int64_t)nwg * BM * d_ffn > 0xffffffffLL) return CSSHIP_EBADSHAPE;  // 32-bit dropout element indices
  if (dyd_bf16 && !aligned8(dyd_bf16)) return CSSHIP_EALIGN;
  const bool ln2 = ln2_x != nullptr;
  if (ln2) {  // the second LayerNorm rides in the whole-panel form only
    if (!ln2_g || !dgamma_dbeta2) return CSSHIP_ENULL;
    if ((M % BM) != 0 || d_ffn > CS_FFMAX) return CSSHIP_EUNSUPPORTED;
    if (!aligned16(ln2_x) || !aligned16(ln2_g)) return CSSHIP_EALIGN;
    if (workspace_bytes < (size_t)nwg * 4 * DM * sizeof(float)) return CSSHIP_EWORKSPACE;
  }
  const csship_dropout_cfg none{nullptr, 0u, 0.f};
  float* const part2 = reinterpret_cast<float*>(workspace) + (size_t)nwg * 2 * DM;
  FfnBwdArgs a{x, dy, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w1_packed), b1,
               reinterpret_cast<const bf16x8*>(w2t_packed), reinterpret_cast<const bf16x8*>(w1t_packed), a_bf16, h_bf16,
               dh_bf16, dx, reinterpret_cast<float*>(workspace), M, d_ffn, alpha,
               drop_hidden ? *drop_hidden : none, drop_out ? *drop_out : none, dyd_bf16, ln2_x, ln2_g, ln2 ? part2 : nullptr};
  static const bool use_static = [] { const char* e = getenv("CSSHIP_CHAIN_STATIC"); return e == nullptr || atoi(e) != 0; }();
  if (ln2)
    hipLaunchKernelGGL((ffn_bwd_kernel<NWAVES, true, true>), dim3(nwg), dim3(NWAVES * 64), 0, as_stream(stream), a);
  else if (use_static && (M % BM) == 0 && d_ffn <= CS_FFMAX)
    hipLaunchKernelGGL((ffn_bwd_kernel<NWAVES, true>), dim3(nwg), dim3(NWAVES * 64), 0, as_stream(stream), a);
  else
    hipLaunchKernelGGL((ffn_bwd_kernel<NWAVES, false>), dim3(nwg), dim3(NWAVES * 64), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  const int rc = launch_slab_reduce(reinterpret_cast<const float*>(workspace), nwg, 2 * DM, 2 * DM, 2 * DM, 1.f, nullptr, nullptr, 0,
                                    dgamma_dbeta, as_stream(stream), true);
  if (rc != CSSHIP_OK || !ln2) return rc;
  return launch_slab_reduce(part2, nwg, 2 * DM, 2 * DM, 2 * DM, 1.f, nullptr, nullptr, 0, dgamma_dbeta2, as_stream(stream), true);
}

extern "C" size_t csship_ffn_bwd_conv_workspace(int32_t M) {
  return csship_ffn_bwd_workspace(M) + (size_t)((M + BM - 1) / BM) * (2 * DM + 2) * sizeof(float);
}

extern "C" int csship_ffn_bwd_conv(const float* x, const float* dy, const float* ln_g, const float* ln_b,
                                   const uint16_t* w1_packed, const float* b1, const uint16_t* w2t_packed,
                                   const uint16_t* w1t_packed, uint16_t* a_bf16, uint16_t* h_bf16, uint16_t* dh_bf16,
                                   float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                                   const csship_dropout_cfg* drop_hidden, const csship_dropout_cfg* drop_out,
                                   uint16_t* dyd_bf16, const void* conv_fwd_workspace, int32_t B, int32_t T,
                                   const float* conv_pw2_w, const csship_dropout_cfg* conv_drop, float* conv_grads,
                                   void* workspace, size_t workspace_bytes, csship_stream_t stream) {
  if (!x || !dy || !ln_g || !ln_b || !w1_packed || !b1 || !w2t_packed || !w1t_packed || !a_bf16 || !h_bf16 || !dh_bf16 ||
      !dx || !dgamma_dbeta || !workspace || !conv_fwd_workspace || !conv_pw2_w || !conv_grads)
    return CSSHIP_ENULL;
  if (M <= 0 || B <= 0 || T <= 0 || (int64_t)B * T != M) return CSSHIP_EBADSHAPE;
  if (d_model != DM || d_ffn <= 0 || (d_ffn % DM) != 0 || (M % BM) != 0 || d_ffn > CS_FFMAX) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || !aligned16(ln_g) || !aligned16(ln_b) || !aligned16(b1) ||
      !aligned16(w1_packed) || !aligned16(w2t_packed) || !aligned16(w1t_packed) || !aligned8(a_bf16) || !aligned8(h_bf16) ||
      !aligned8(dh_bf16) || !aligned16(conv_fwd_workspace) || (dyd_bf16 && !aligned8(dyd_bf16)))
    return CSSHIP_EALIGN;
  const int nwg = M / BM;
  if (workspace_bytes < csship_ffn_bwd_conv_workspace(M)) return CSSHIP_EWORKSPACE;
  if ((int64_t)nwg * BM * d_ffn > 0xffffffffLL) return CSSHIP_EBADSHAPE;  // 32-bit dropout element indices
  const csship_dropout_cfg none{nullptr, 0u, 0.f};
  const bool cdropping = conv_drop && conv_drop->seed && conv_drop->p > 0.f;
  float* const wsf = reinterpret_cast<float*>(workspace);
  float* const cs_part = wsf + (size_t)nwg * 2 * DM;
  const float* const z = reinterpret_cast<const float*>(conv_fwd_workspace);
  FfnBwdArgs a{x, dy, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w1_packed), b1,
               reinterpret_cast<const bf16x8*>(w2t_packed), reinterpret_cast<const bf16x8*>(w1t_packed), a_bf16, h_bf16,
               dh_bf16, dx, wsf, M, d_ffn, alpha, drop_hidden ? *drop_hidden : none, drop_out ? *drop_out : none, dyd_bf16,
               nullptr, nullptr, nullptr, z, z + convmod_fwd_stats_offset(B, T, DM), conv_pw2_w, cdropping ? *conv_drop : none, cs_part};
  hipLaunchKernelGGL((ffn_bwd_kernel<NWAVES, true, false, true>), dim3(nwg), dim3(NWAVES * 64), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  // the conv module's sums are read by its own backward right behind this call: reduced at once, like pass A's (never deferred)
  int rc = launch_slab_reduce(cs_part, nwg, 2 * DM + 2, 2 * DM + 2, 2 * DM + 2, 1.f, nullptr, nullptr, 0, conv_grads, as_stream(stream));
  if (rc != CSSHIP_OK) return rc;
  return launch_slab_reduce(wsf, nwg, 2 * DM, 2 * DM, 2 * DM, 1.f, nullptr, nullptr, 0, dgamma_dbeta, as_stream(stream), true);
}

extern "C" int csship_ffn_bwd(const float* x, const float* dy, const float* ln_g, const float* ln_b,
                              const uint16_t* w1_packed, const float* b1, const uint16_t* w2t_packed,
                              const uint16_t* w1t_packed, uint16_t* a_bf16, uint16_t* h_bf16, uint16_t* dh_bf16,
                              float* dx, float* dgamma_dbeta, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                              void* workspace, size_t workspace_bytes, csship_stream_t stream) {
  return csship_ffn_bwd_drop(x, dy, ln_g, ln_b, w1_packed, b1, w2t_packed, w1t_packed, a_bf16, h_bf16, dh_bf16, dx,
                             dgamma_dbeta, M, d_model, d_ffn, alpha, nullptr, nullptr, nullptr, workspace, workspace_bytes,
                             stream);
}

// =============================================================================== q | k | v projection backward (dX side)
// qkv = LayerNorm(x) Wqkv^T + b is the front of the attention block (conformer.py:213-224); out = x + attention(...).
// Given dqkv (bf16 [M, 768]) and dy = d(out), ONE kernel per 64-row panel:
//   da = dqkv Wqkv          24 k-steps over the panel's three 256-column chunks (all three in LDS: 96 KB), the
//                           fragment-packed Wqkv^T [256, 768] streamed through the weight ring without a pause
//   dx = dy + LayerNorm-backward(da);  y = LayerNorm(x) as bf16 (second operand of dWqkv = dqkv^T y);
//   dgamma / dbeta partials per workgroup.
// Replaces the tiled GEMM (17.5 us) + the LayerNorm-backward row kernel (11.8 us) and the f32 round trip of da between
// them.  Whole panels only (M % 64 == 0); every global memory instruction is unconditional (see chain_s_kernel).
struct QkvBwdArgs {
  const unsigned short* dqkv;
  const bf16x8* wt;   // packed from Wqkv^T [256, 768] (the q rows of Wqkv carry the forward's 1/sqrt(d_k))
  const float* x;
  const float* dy;
  const float *ln_g, *ln_b;
  float* dx;
  unsigned short* y_bf;
  float* part;        // [nwg][2][256]
  int M;
};

// LDS: Ps = three bf16 panels (96 KB); later the Ys tile, then da * xhat [64][260] | da [64][256] for the column sums;
// Pb = gamma | beta | dump.  KEEP: dx stays in registers (keep[i] = this thread's pieces of row tid >> 3) for the stage behind
// it in the same kernel (qkv_ffn_bwd_kernel) instead of going to memory.
constexpr int QB_PS_BYTES = YS_BYTES + 2 * PANEL_BYTES, QB_PB_FLOATS = 2 * DM + CS_DUMP;
template <int NW, bool KEEP>
DEV void qkv_in_bwd_body(const QkvBwdArgs& a, unsigned char* const Ps, float* const Pb, float4 (&keep)[8]) {
  static_assert(NW * 64 == 8 * BM, "row phases assume 8 lanes per panel row");
  constexpr int NT = 16 / NW;
  constexpr int KT = 3 * KS;  // 24 k-steps
  static_assert(YS_BYTES + 2 * PANEL_BYTES >= 3 * PANEL_BYTES, "the three panels must fit");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM;
  const bf16x8* ws = a.wt + (size_t)(wave * NT) * KT * 64 + lane;
  bf16x8 wq[PD][NT];
  // the panel's 96 KB of dqkv and the LayerNorm affine: unconditional 16-byte loads
  f32x4 pv0[4], pv1[4], pv2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + NW * 64 * j, r = i >> 5, c = i & 31;
    const unsigned short* src = a.dqkv + (size_t)(m0 + r) * (3 * DM) + c * 8;
    pv0[j] = *reinterpret_cast<const f32x4*>(src);
    pv1[j] = *reinterpret_cast<const f32x4*>(src + DM);
    pv2[j] = *reinterpret_cast<const f32x4*>(src + 2 * DM);
  }
  const float4 pg = *reinterpret_cast<const float4*>((tid < 64 ? a.ln_g : a.ln_b) + lane * 4);
  prefetch_first<NT>(wq, ws, KT, NT);
  *reinterpret_cast<float4*>(Pb + (tid < 128 ? tid * 4 : 2 * DM + lane * 4)) = pg;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + NW * 64 * j, r = i >> 5, c = i & 31;
    *reinterpret_cast<f32x4*>(Ps + panel_off(r, c)) = pv0[j];
    *reinterpret_cast<f32x4*>(Ps + PANEL_BYTES + panel_off(r, c)) = pv1[j];
    *reinterpret_cast<f32x4*>(Ps + 2 * PANEL_BYTES + panel_off(r, c)) = pv2[j];
  }
  __syncthreads();

  f32x4 o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int r = tid >> 3, part = tid & 7;
  const int row = m0 + r;
  float4 v[8], dyv[8];
  {
    bf16x8 xa[2][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) xa[0][rt] = *reinterpret_cast<const bf16x8*>(Ps + panel_off(rt * 16 + n, g));
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
      bf16x8 wc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) wc[t] = wq[ks % PD][t];
      if (ks + PD < KT) {
#pragma unroll
        for (int t = 0; t < NT; ++t) wq[ks % PD][t] = ws[(size_t)(t * KT + ks + PD) * 64];
      }
      if (ks == KT - PD) {  // behind the last weight fragments: the x and dy rows of the epilogue
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          v[i] = *reinterpret_cast<const float4*>(a.x + (size_t)row * DM + (i * 8 + part) * 4);
          dyv[i] = *reinterpret_cast<const float4*>(a.dy + (size_t)row * DM + (i * 8 + part) * 4);
        }
      }
      if (ks + 1 < KT) {
        const unsigned char* pn = Ps + ((ks + 1) >> 3) * PANEL_BYTES;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          xa[(ks + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(pn + panel_off(rt * 16 + n, ((ks + 1) & 7) * 4 + g));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          o[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], xa[ks & 1][rt], o[t][rt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();  // every wave is done reading the panels
  float* ys = reinterpret_cast<float*>(Ps);
  {
    float4 zero[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) zero[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    park_tiles<NT, false>(ys, o, NT, wave * NT * 16, g, n, zero);
  }
  __syncthreads();
  const float* lgp = Pb;
  const float* lbp = Pb + DM;
  float4 da[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) da[i] = *reinterpret_cast<const float4*>(ys + r * YS + (i * 8 + part) * 4);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = sum8(s) * (1.f / DM);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(sum8(ss) * (1.f / DM) + 1e-5f);
  float s1 = 0.f, s2 = 0.f;
  float4 gg[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 gm = *reinterpret_cast<const float4*>(lgp + (i * 8 + part) * 4);
    const float4 bt = *reinterpret_cast<const float4*>(lbp + (i * 8 + part) * 4);
    v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;  // xhat
    *reinterpret_cast<uint2*>(a.y_bf + (size_t)row * DM + (i * 8 + part) * 4) =
        make_uint2(pack_bf2(v[i].x * gm.x + bt.x, v[i].y * gm.y + bt.y), pack_bf2(v[i].z * gm.z + bt.z, v[i].w * gm.w + bt.w));
    gg[i] = make_float4(da[i].x * gm.x, da[i].y * gm.y, da[i].z * gm.z, da[i].w * gm.w);
    s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
    s2 += (gg[i].x * v[i].x + gg[i].y * v[i].y) + (gg[i].z * v[i].z + gg[i].w * v[i].w);
  }
  s1 = sum8(s1) * (1.f / DM);
  s2 = sum8(s2) * (1.f / DM);
  __syncthreads();  // all da values are in registers: the Ys tile can be reused for the column sums
  float* cg = reinterpret_cast<float*>(Ps);             // [64][260] da * xhat
  float* cb = reinterpret_cast<float*>(Ps + YS_BYTES);  // [64][256] da
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int col = (i * 8 + part) * 4;
    const float4 d = dyv[i];
    float4 o4;
    o4.x = d.x + rstd * (gg[i].x - s1 - v[i].x * s2);
    o4.y = d.y + rstd * (gg[i].y - s1 - v[i].y * s2);
    o4.z = d.z + rstd * (gg[i].z - s1 - v[i].z * s2);
    o4.w = d.w + rstd * (gg[i].w - s1 - v[i].w * s2);
    // (KEEP: the two instantiations contract their float expressions differently here and there - the chained kernel agrees
    // with the two launches to an ulp of dy, which a bf16 rounding behind it can amplify to a bf16 ulp in a handful of elements)
    if constexpr (KEEP) keep[i] = o4;
    else *reinterpret_cast<float4*>(a.dx + (size_t)row * DM + col) = o4;
    *reinterpret_cast<float4*>(cg + r * YS + col) = make_float4(da[i].x * v[i].x, da[i].y * v[i].y, da[i].z * v[i].z, da[i].w * v[i].w);
    *reinterpret_cast<float4*>(cb + r * DM + col) = da[i];
  }
  __syncthreads();
  {
    const int col = tid & 255, which = tid >> 8;  // 512 threads: 256 columns x {dgamma, dbeta}
    float acc = 0.f;
    if (which == 0) {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) acc += cg[rr * YS + col];
    } else {
#pragma unroll 8
      for (int rr = 0; rr < BM; ++rr) acc += cb[rr * DM + col];
    }
    a.part[(size_t)blockIdx.x * 2 * DM + which * DM + col] = acc;
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void qkv_in_bwd_kernel(QkvBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char Ps[QB_PS_BYTES];
  __shared__ __attribute__((aligned(16))) float Pb[QB_PB_FLOATS];
  float4 none[8];   // (never written: KEEP = false)
  qkv_in_bwd_body<NW, false>(a, Ps, Pb, none);
}

// The q | k | v projection's input-side backward and the FeedForward backward behind it in the backward pass (the block's
// feed_forward_in, conformer.py:355) as ONE kernel per 64-row panel: dx of the first stage is the dy of the second and has the
// second's row mapping (8 lanes per row), so it stays in registers - no f32 [M, 256] round trip, one launch, and the
// FeedForward's prologue finds half of its rows already there.  The two stages share the LDS (the three dqkv panels and the
// column-sum tiles of the first = the LN(x) | dy panels + the dh panels of the second, byte for byte).
constexpr int QF_RAW_BYTES = QB_PS_BYTES > FB_XD_BYTES + FB_GRAW_BYTES ? QB_PS_BYTES : FB_XD_BYTES + FB_GRAW_BYTES;
template <int NW, bool LN2>
__global__ __launch_bounds__(NW * 64) void qkv_ffn_bwd_kernel(QkvBwdArgs q, FfnBwdArgs f) {
  __shared__ __attribute__((aligned(16))) unsigned char RAW[QF_RAW_BYTES];
  __shared__ __attribute__((aligned(16))) float Pb1[QB_PB_FLOATS];
  __shared__ __attribute__((aligned(16))) float Pb2[FB_PB_FLOATS];
  float4 dxk[8];
  qkv_in_bwd_body<NW, true>(q, RAW, Pb1, dxk);
  __syncthreads();   // the first stage's column sums have been read: its tiles become the second stage's panels
  ffn_bwd_body<NW, true, LN2, true>(f, RAW, RAW + FB_XD_BYTES, Pb2, dxk);
}

// ... and the FeedForward backward of the BLOCK BELOW behind that (its feed_forward_out, conformer.py:358-360: the second stage's
// dx through the block's final LayerNorm is its dy), with the conv module's statistics in its epilogue: three stages, one launch.
template <int NW>
__global__ __launch_bounds__(NW * 64) void qkv_ffn2_bwd_kernel(QkvBwdArgs q, FfnBwdArgs f1, FfnBwdArgs f2) {
  __shared__ __attribute__((aligned(16))) unsigned char RAW[QF_RAW_BYTES];
  __shared__ __attribute__((aligned(16))) float Pb1[QB_PB_FLOATS];
  __shared__ __attribute__((aligned(16))) float Pb2[FB_PB_FLOATS + 4 * DM];
  float4 dxk[8];
  qkv_in_bwd_body<NW, true>(q, RAW, Pb1, dxk);
  __syncthreads();
  ffn_bwd_body<NW, true, true, true, false, true>(f1, RAW, RAW + FB_XD_BYTES, Pb2, dxk);
  __syncthreads();   // the second stage's column sums have been read; its parameter block is restaged by the third
  ffn_bwd_body<NW, true, false, true, true, false>(f2, RAW, RAW + FB_XD_BYTES, Pb2, dxk);
}

extern "C" size_t csship_qkv_in_bwd_workspace(int32_t M) { return (size_t)((M + BM - 1) / BM) * 2 * DM * sizeof(float); }

extern "C" int csship_qkv_in_bwd(const uint16_t* dqkv, const uint16_t* wqkvT_packed, const float* x, const float* dy,
                                 const float* ln_g, const float* ln_b, float* dx, uint16_t* y_bf16, float* dgamma_dbeta,
                                 int32_t M, int32_t d_model, void* workspace, size_t workspace_bytes,
                                 csship_stream_t stream) {
  if (!dqkv || !wqkvT_packed || !x || !dy || !ln_g || !ln_b || !dx || !y_bf16 || !dgamma_dbeta || !workspace) return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (d_model != DM || (M % BM) != 0) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(dqkv) || !aligned16(wqkvT_packed) || !aligned16(x) || !aligned16(dy) || !aligned16(ln_g) ||
      !aligned16(ln_b) || !aligned16(dx) || !aligned8(y_bf16))
    return CSSHIP_EALIGN;
  const int nwg = M / BM;
  if (workspace_bytes < (size_t)nwg * 2 * DM * sizeof(float)) return CSSHIP_EWORKSPACE;
  QkvBwdArgs a{dqkv, reinterpret_cast<const bf16x8*>(wqkvT_packed), x, dy, ln_g, ln_b, dx, y_bf16,
               reinterpret_cast<float*>(workspace), M};
  hipLaunchKernelGGL((qkv_in_bwd_kernel<NWAVES>), dim3(nwg), dim3(NWAVES * 64), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return launch_slab_reduce(reinterpret_cast<const float*>(workspace), nwg, 2 * DM, 2 * DM, 2 * DM, 1.f, nullptr, nullptr, 0,
                            dgamma_dbeta, as_stream(stream), true);
}

extern "C" size_t csship_qkv_ffn_bwd_workspace(int32_t M) { return (size_t)((M + BM - 1) / BM) * 6 * DM * sizeof(float); }

extern "C" int csship_qkv_ffn_bwd(const uint16_t* dqkv, const uint16_t* wqkvT_packed, const float* x_attn, const float* dy,
                                  const float* attn_ln_g, const float* attn_ln_b, uint16_t* y_bf16, float* attn_dgamma_dbeta,
                                  const float* x, const float* ln_g, const float* ln_b, const uint16_t* w1_packed,
                                  const float* b1, const uint16_t* w2t_packed, const uint16_t* w1t_packed, uint16_t* a_bf16,
                                  uint16_t* h_bf16, uint16_t* dh_bf16, float* dx, float* dgamma_dbeta, int32_t M,
                                  int32_t d_model, int32_t d_ffn, float alpha, const csship_dropout_cfg* drop_hidden,
                                  const csship_dropout_cfg* drop_out, uint16_t* dyd_bf16, const float* ln2_x,
                                  const float* ln2_g, float* dgamma_dbeta2, void* workspace, size_t workspace_bytes,
                                  csship_stream_t stream) {
  if (!dqkv || !wqkvT_packed || !x_attn || !dy || !attn_ln_g || !attn_ln_b || !y_bf16 || !attn_dgamma_dbeta || !x || !ln_g ||
      !ln_b || !w1_packed || !b1 || !w2t_packed || !w1t_packed || !a_bf16 || !h_bf16 || !dh_bf16 || !dx || !dgamma_dbeta ||
      !dyd_bf16 || !workspace)
    return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (d_model != DM || (M % BM) != 0 || d_ffn <= 0 || (d_ffn % DM) != 0 || d_ffn > CS_FFMAX) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(dqkv) || !aligned16(wqkvT_packed) || !aligned16(x_attn) || !aligned16(dy) || !aligned16(attn_ln_g) ||
      !aligned16(attn_ln_b) || !aligned8(y_bf16) || !aligned16(x) || !aligned16(dx) || !aligned16(ln_g) || !aligned16(ln_b) ||
      !aligned16(b1) || !aligned16(w1_packed) || !aligned16(w2t_packed) || !aligned16(w1t_packed) || !aligned8(a_bf16) ||
      !aligned8(h_bf16) || !aligned8(dh_bf16) || !aligned8(dyd_bf16))
    return CSSHIP_EALIGN;
  const int nwg = M / BM;
  if (workspace_bytes < csship_qkv_ffn_bwd_workspace(M)) return CSSHIP_EWORKSPACE;
  if ((int64_t)nwg * BM * d_ffn > 0xffffffffLL) return CSSHIP_EBADSHAPE;  // 32-bit dropout element indices
  const bool ln2 = ln2_x != nullptr;
  if (ln2) {
    if (!ln2_g || !dgamma_dbeta2) return CSSHIP_ENULL;
    if (!aligned16(ln2_x) || !aligned16(ln2_g)) return CSSHIP_EALIGN;
  }
  float* const wsf = reinterpret_cast<float*>(workspace);   // [nwg][2][256] x 3: the FeedForward's, its second LayerNorm's, the q | k | v LayerNorm's
  float* const part2 = wsf + (size_t)nwg * 2 * DM;
  float* const partq = wsf + (size_t)nwg * 4 * DM;
  const csship_dropout_cfg none{nullptr, 0u, 0.f};
  QkvBwdArgs qa{dqkv, reinterpret_cast<const bf16x8*>(wqkvT_packed), x_attn, dy, attn_ln_g, attn_ln_b, nullptr, y_bf16, partq, M};
  FfnBwdArgs fa{x, nullptr, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w1_packed), b1,
                reinterpret_cast<const bf16x8*>(w2t_packed), reinterpret_cast<const bf16x8*>(w1t_packed), a_bf16, h_bf16,
                dh_bf16, dx, wsf, M, d_ffn, alpha, drop_hidden ? *drop_hidden : none, drop_out ? *drop_out : none, dyd_bf16,
                ln2_x, ln2_g, ln2 ? part2 : nullptr};
  if (ln2) hipLaunchKernelGGL((qkv_ffn_bwd_kernel<NWAVES, true>), dim3(nwg), dim3(NWAVES * 64), 0, as_stream(stream), qa, fa);
  else hipLaunchKernelGGL((qkv_ffn_bwd_kernel<NWAVES, false>), dim3(nwg), dim3(NWAVES * 64), 0, as_stream(stream), qa, fa);
  CSSHIP_CHECK_LAUNCH();
  int rc = launch_slab_reduce(partq, nwg, 2 * DM, 2 * DM, 2 * DM, 1.f, nullptr, nullptr, 0, attn_dgamma_dbeta, as_stream(stream), true);
  if (rc != CSSHIP_OK) return rc;
  rc = launch_slab_reduce(wsf, nwg, 2 * DM, 2 * DM, 2 * DM, 1.f, nullptr, nullptr, 0, dgamma_dbeta, as_stream(stream), true);
  if (rc != CSSHIP_OK || !ln2) return rc;
  return launch_slab_reduce(part2, nwg, 2 * DM, 2 * DM, 2 * DM, 1.f, nullptr, nullptr, 0, dgamma_dbeta2, as_stream(stream), true);
}

extern "C" size_t csship_qkv_ffn2_bwd_workspace(int32_t M) { return (size_t)((M + BM - 1) / BM) * (10 * DM + 2) * sizeof(float); }

extern "C" int csship_qkv_ffn2_bwd(const csship_qkv_bwd_args* q, const csship_ffn_bwd_args* fi, const csship_ffn_bwd_args* fo,
                                   const csship_conv_stats_args* cv, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                                   void* workspace, size_t workspace_bytes, csship_stream_t stream) {
  if (!q || !fi || !fo || !cv || !workspace) return CSSHIP_ENULL;
  if (!q->dqkv || !q->wqkvT_packed || !q->x || !q->dy || !q->ln_g || !q->ln_b || !q->y_bf16 || !q->dgamma_dbeta) return CSSHIP_ENULL;
  for (const csship_ffn_bwd_args* f : {fi, fo}) {
    if (!f->x || !f->ln_g || !f->ln_b || !f->w1_packed || !f->b1 || !f->w2t_packed || !f->w1t_packed || !f->a_bf16 || !f->h_bf16 ||
        !f->dh_bf16 || !f->dgamma_dbeta || !f->dyd_bf16)
      return CSSHIP_ENULL;
    if (!aligned16(f->x) || !aligned16(f->ln_g) || !aligned16(f->ln_b) || !aligned16(f->b1) || !aligned16(f->w1_packed) ||
        !aligned16(f->w2t_packed) || !aligned16(f->w1t_packed) || !aligned8(f->a_bf16) || !aligned8(f->h_bf16) ||
        !aligned8(f->dh_bf16) || !aligned8(f->dyd_bf16))
      return CSSHIP_EALIGN;
  }
  if (!fi->ln2_x || !fi->ln2_g || !fi->dgamma_dbeta2 || !fo->dx || fo->ln2_x) return CSSHIP_ENULL;   // the chain runs through that LayerNorm
  if (!cv->fwd_workspace || !cv->pw2_w || !cv->grads) return CSSHIP_ENULL;
  if (M <= 0 || cv->B <= 0 || cv->T <= 0 || (int64_t)cv->B * cv->T != M) return CSSHIP_EBADSHAPE;
  if (d_model != DM || (M % BM) != 0 || d_ffn <= 0 || (d_ffn % DM) != 0 || d_ffn > CS_FFMAX) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(q->dqkv) || !aligned16(q->wqkvT_packed) || !aligned16(q->x) || !aligned16(q->dy) || !aligned16(q->ln_g) ||
      !aligned16(q->ln_b) || !aligned8(q->y_bf16) || !aligned16(fi->ln2_x) || !aligned16(fi->ln2_g) || !aligned16(fo->dx) ||
      !aligned16(cv->fwd_workspace))
    return CSSHIP_EALIGN;
  const int nwg = M / BM;
  if (workspace_bytes < csship_qkv_ffn2_bwd_workspace(M)) return CSSHIP_EWORKSPACE;
  if ((int64_t)nwg * BM * d_ffn > 0xffffffffLL) return CSSHIP_EBADSHAPE;  // 32-bit dropout element indices
  float* const wsf = reinterpret_cast<float*>(workspace);   // [nwg][2][256] x 4 (FeedForward, its second LayerNorm, q | k | v, the FeedForward below) + [nwg][514]
  float* const part_i = wsf, * const part_i2 = wsf + (size_t)nwg * 2 * DM, * const part_q = wsf + (size_t)nwg * 4 * DM;
  float* const part_o = wsf + (size_t)nwg * 6 * DM, * const part_cs = wsf + (size_t)nwg * 8 * DM;
  const csship_dropout_cfg none{nullptr, 0u, 0.f};
  auto cfg = [&](const csship_dropout_cfg* d) { return d && d->seed && d->p > 0.f ? *d : none; };
  const float* const z = reinterpret_cast<const float*>(cv->fwd_workspace);
  QkvBwdArgs qa{q->dqkv, reinterpret_cast<const bf16x8*>(q->wqkvT_packed), q->x, q->dy, q->ln_g, q->ln_b, nullptr, q->y_bf16, part_q, M};
  FfnBwdArgs f1{fi->x, nullptr, fi->ln_g, fi->ln_b, reinterpret_cast<const bf16x8*>(fi->w1_packed), fi->b1,
                reinterpret_cast<const bf16x8*>(fi->w2t_packed), reinterpret_cast<const bf16x8*>(fi->w1t_packed), fi->a_bf16, fi->h_bf16,
                fi->dh_bf16, nullptr, part_i, M, d_ffn, alpha, cfg(fi->drop_hidden), cfg(fi->drop_out), fi->dyd_bf16, fi->ln2_x, fi->ln2_g,
                part_i2};
  FfnBwdArgs f2{fo->x, nullptr, fo->ln_g, fo->ln_b, reinterpret_cast<const bf16x8*>(fo->w1_packed), fo->b1,
                reinterpret_cast<const bf16x8*>(fo->w2t_packed), reinterpret_cast<const bf16x8*>(fo->w1t_packed), fo->a_bf16, fo->h_bf16,
                fo->dh_bf16, fo->dx, part_o, M, d_ffn, alpha, cfg(fo->drop_hidden), cfg(fo->drop_out), fo->dyd_bf16, nullptr, nullptr,
                nullptr, z, z + convmod_fwd_stats_offset(cv->B, cv->T, DM), cv->pw2_w, cfg(cv->drop), part_cs};
  hipLaunchKernelGGL((qkv_ffn2_bwd_kernel<NWAVES>), dim3(nwg), dim3(NWAVES * 64), 0, as_stream(stream), qa, f1, f2);
  CSSHIP_CHECK_LAUNCH();
  // the conv module's sums are read by its own backward right behind this call: reduced at once; the rest are parameter gradients
  int rc = launch_slab_reduce(part_cs, nwg, 2 * DM + 2, 2 * DM + 2, 2 * DM + 2, 1.f, nullptr, nullptr, 0, cv->grads, as_stream(stream));
  if (rc != CSSHIP_OK) return rc;
  const struct { const float* part; float* out; } red[4] = {{part_q, q->dgamma_dbeta}, {part_i, fi->dgamma_dbeta},
                                                            {part_i2, fi->dgamma_dbeta2}, {part_o, fo->dgamma_dbeta}};
  for (const auto& r : red) {
    rc = launch_slab_reduce(r.part, nwg, 2 * DM, 2 * DM, 2 * DM, 1.f, nullptr, nullptr, 0, r.out, as_stream(stream), true);
    if (rc != CSSHIP_OK) return rc;
  }
  return CSSHIP_OK;
}

// out bf16 [M, 256] = a bf16 [M, 256] W^T with no bias, whole panels: the input gradient of a 256 -> 256 Linear against
// its fragment-packed transpose (the attention output projection's dao = dy Wo).  Accumulators go to global as bf16
// straight from the MFMA layout (16-byte pieces after the tile pairing), no staged epilogue.
template <int NW>
__global__ __launch_bounds__(NW * 64) void panel_dx_kernel(const unsigned short* __restrict__ a, const bf16x8* __restrict__ w,
                                                           unsigned short* __restrict__ out, int64_t ldo) {
  constexpr int NT = 16 / NW;
  static_assert(NT == 2, "tile pairing");
  __shared__ __attribute__((aligned(16))) unsigned char Xs[PANEL_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM;
  const bf16x8* ws = w + (size_t)(wave * NT) * KS * 64 + lane;
  bf16x8 wq[PD][NT];
  uint4 pv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + NW * 64 * j, r = i >> 5, c = i & 31;
    pv[j] = *reinterpret_cast<const uint4*>(a + (size_t)(m0 + r) * DM + c * 8);
  }
  prefetch_first<NT>(wq, ws, KS, NT);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + NW * 64 * j, r = i >> 5, c = i & 31;
    *reinterpret_cast<uint4*>(Xs + panel_off(r, c)) = pv[j];
  }
  __syncthreads();
  f32x4 o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  mma_pass_s<NT, false>(o, Xs, n, g, ws, KS, ws, KS, wq);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const uint4 pk = pair_tiles_bf16(make_uint2(pack_bf2(o[0][rt][0], o[0][rt][1]), pack_bf2(o[0][rt][2], o[0][rt][3])),
                                     make_uint2(pack_bf2(o[1][rt][0], o[1][rt][1]), pack_bf2(o[1][rt][2], o[1][rt][3])));
    *reinterpret_cast<uint4*>(out + (size_t)(m0 + rt * 16 + n) * ldo + (wave * NT + (g & 1)) * 16 + (g >> 1) * 8) = pk;
  }
}

extern "C" int csship_panel_linear(const void* a, int32_t prologue, const float* ln_g, const float* ln_b,
                                   const uint16_t* w_packed, const float* bias, const float* residual, int64_t ldr,
                                   float alpha, void* out, int64_t ldo, int32_t M, int32_t N, int32_t d_model,
                                   int32_t epilogue, csship_stream_t stream) {
  return csship_panel_linear_drop(a, prologue, ln_g, ln_b, w_packed, bias, residual, ldr, alpha, out, ldo, M, N, d_model,
                                  epilogue, nullptr, stream);
}

extern "C" int csship_panel_linear_drop(const void* a, int32_t prologue, const float* ln_g, const float* ln_b,
                                        const uint16_t* w_packed, const float* bias, const float* residual, int64_t ldr,
                                        float alpha, void* out, int64_t ldo, int32_t M, int32_t N, int32_t d_model,
                                        int32_t epilogue, const csship_dropout_cfg* drop, csship_stream_t stream) {
  if (!a || !w_packed || !out) return CSSHIP_ENULL;
  const bool dropping = drop != nullptr && drop->seed != nullptr && drop->p > 0.f;
  // the mask lives in the vector form of the f32 residual epilogue: whole 4-column groups, 16-byte aligned output rows
  if (dropping && (epilogue != CSSHIP_EPI_RESIDUAL || (N & 3) != 0 || (ldo & 3) != 0 || (int64_t)M * N > 0xffffffffLL))
    return CSSHIP_EUNSUPPORTED;
  if (M <= 0 || N <= 0 || ldo < N) return CSSHIP_EBADSHAPE;
  if (d_model != DM) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(a) || !aligned16(w_packed) || !aligned16(out)) return CSSHIP_EALIGN;
  if (prologue == CSSHIP_PRO_LN && (!ln_g || !ln_b || !aligned16(ln_g) || !aligned16(ln_b))) return CSSHIP_ENULL;
  if ((epilogue & CSSHIP_EPI_RESIDUAL) && (!residual || !aligned16(residual) || (ldr & 3) != 0 || ldr < N || (N & 3) != 0))
    return CSSHIP_EALIGN;
  LinArgs la{a, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w_packed), bias, residual, out, M, N, ldo, ldr, alpha,
             g_csship_stamps, nullptr, nullptr, nullptr, nullptr, nullptr,
             dropping ? drop->seed : nullptr, dropping ? drop->stream : 0u, dropping ? drop->p : 0.f};
  hipStream_t s = as_stream(stream);
  const int key = prologue * 16 + epilogue;
  // (bench.py's train-mode timeline: the mask head is the one row-panel Linear whose rows are not a multiple of 256 wide)
  TimelineScope tl((N % DM) != 0 ? CSSHIP_TL_HEAD : CSSHIP_TL_OTHER, s);
  static const bool use_static = [] { const char* e = getenv("CSSHIP_CHAIN_STATIC"); return e == nullptr || atoi(e) != 0; }();
  if (use_static && key == CSSHIP_PRO_BF16 * 16 + CSSHIP_EPI_OUT_BF16 && !bias && N == DM && (M % BM) == 0 && (ldo & 7) == 0) {
    hipLaunchKernelGGL((panel_dx_kernel<NWAVES>), dim3(M / BM), dim3(NWAVES * 64), 0, s,
                       reinterpret_cast<const unsigned short*>(a), reinterpret_cast<const bf16x8*>(w_packed),
                       reinterpret_cast<unsigned short*>(out), ldo);
    CSSHIP_CHECK_LAUNCH();
    return CSSHIP_OK;
  }
  switch (key) {
    case CSSHIP_PRO_LN * 16 + CSSHIP_EPI_OUT_BF16: return launch_linear<CSSHIP_PRO_LN, CSSHIP_EPI_OUT_BF16>(la, s);
    case CSSHIP_PRO_LN * 16 + 0: return launch_linear<CSSHIP_PRO_LN, 0>(la, s);
    case CSSHIP_PRO_BF16 * 16 + CSSHIP_EPI_RESIDUAL: return launch_linear<CSSHIP_PRO_BF16, CSSHIP_EPI_RESIDUAL>(la, s);
    case CSSHIP_PRO_BF16 * 16 + 0: return launch_linear<CSSHIP_PRO_BF16, 0>(la, s);
    case CSSHIP_PRO_BF16 * 16 + CSSHIP_EPI_OUT_BF16: return launch_linear<CSSHIP_PRO_BF16, CSSHIP_EPI_OUT_BF16>(la, s);
    case CSSHIP_PRO_F32 * 16 + CSSHIP_EPI_RELU: return launch_linear<CSSHIP_PRO_F32, CSSHIP_EPI_RELU>(la, s);
    case CSSHIP_PRO_F32 * 16 + 0: return launch_linear<CSSHIP_PRO_F32, 0>(la, s);
    default: return CSSHIP_EUNSUPPORTED;
  }
}


static int encoder_chain_impl(const float* x, float* x_out, const float* conv_g, const csship_conv_weights* conv_w,
                              int32_t T, const csship_ffn_weights* ffn_a, const float* final_ln_g,
                              const float* final_ln_b, const csship_ffn_weights* ffn_b, const float* qkv_ln_g,
                              const float* qkv_ln_b, const uint16_t* wqkv_packed, const float* bqkv,
                              uint16_t* qkv_out, int32_t M, int32_t d_model, int32_t d_ffn, int32_t n_qkv,
                              int32_t flags, csship_stream_t stream, const csship_chain_train* train);

extern "C" int csship_encoder_chain(const float* x, float* x_out, const float* conv_g, const csship_conv_weights* conv_w,
                                    int32_t T, const csship_ffn_weights* ffn_a, const float* final_ln_g,
                                    const float* final_ln_b, const csship_ffn_weights* ffn_b, const float* qkv_ln_g,
                                    const float* qkv_ln_b, const uint16_t* wqkv_packed, const float* bqkv,
                                    uint16_t* qkv_out, int32_t M, int32_t d_model, int32_t d_ffn, int32_t n_qkv,
                                    int32_t flags, csship_stream_t stream) {
  return encoder_chain_impl(x, x_out, conv_g, conv_w, T, ffn_a, final_ln_g, final_ln_b, ffn_b, qkv_ln_g, qkv_ln_b, wqkv_packed,
                            bqkv, qkv_out, M, d_model, d_ffn, n_qkv, flags, stream, nullptr);
}

extern "C" int csship_encoder_chain_train(const float* x, float* x_out, const csship_ffn_weights* ffn_a,
                                          const float* final_ln_g, const float* final_ln_b, const csship_ffn_weights* ffn_b,
                                          const float* qkv_ln_g, const float* qkv_ln_b, const uint16_t* wqkv_packed,
                                          const float* bqkv, uint16_t* qkv_out, int32_t M, int32_t d_model, int32_t d_ffn,
                                          int32_t n_qkv, const csship_chain_train* train, csship_stream_t stream) {
  if (!train) return CSSHIP_ENULL;
  if (!x && train->bn_x2) x = train->bn_x2;   // (with the batch-statistics prologue the rows come from bn_x2 / bn_z)
  if ((int64_t)((M + BM - 1) / BM * BM) * d_ffn > 0xffffffffLL) return CSSHIP_EBADSHAPE;  // 32-bit dropout element indices
  if ((train->pre_ln_out && !aligned16(train->pre_ln_out)) || (train->ln_out && !aligned16(train->ln_out))) return CSSHIP_EALIGN;
  TimelineScope tl(ffn_a ? CSSHIP_TL_CHAIN : CSSHIP_TL_CHAIN_EDGE, as_stream(stream));   // (bench.py's train-mode timeline)
  return encoder_chain_impl(x, x_out, nullptr, nullptr, 0, ffn_a, final_ln_g, final_ln_b, ffn_b, qkv_ln_g, qkv_ln_b, wqkv_packed,
                            bqkv, qkv_out, M, d_model, d_ffn, n_qkv, 0, stream, train);
}

static int encoder_chain_impl(const float* x, float* x_out, const float* conv_g, const csship_conv_weights* conv_w,
                              int32_t T, const csship_ffn_weights* ffn_a, const float* final_ln_g,
                              const float* final_ln_b, const csship_ffn_weights* ffn_b, const float* qkv_ln_g,
                              const float* qkv_ln_b, const uint16_t* wqkv_packed, const float* bqkv,
                              uint16_t* qkv_out, int32_t M, int32_t d_model, int32_t d_ffn, int32_t n_qkv,
                              int32_t flags, csship_stream_t stream, const csship_chain_train* train) {
  if (!x || !x_out) return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (d_model != DM || d_ffn <= 0 || (d_ffn % DM) != 0) return CSSHIP_EUNSUPPORTED;
  const bool A = ffn_a != nullptr, Bs = ffn_b != nullptr, Cq = wqkv_packed != nullptr;
  if (A && (!final_ln_g || !final_ln_b)) return CSSHIP_ENULL;
  if (Cq && (((qkv_ln_g == nullptr) != (qkv_ln_b == nullptr)) || !qkv_out || n_qkv <= 0 || (n_qkv % 8) != 0)) return CSSHIP_EBADSHAPE;
  if ((flags & CSSHIP_CHAIN_QKV_IL8) && (!Cq || (n_qkv % 32) != 0)) return CSSHIP_EBADSHAPE;
  if (!aligned16(x) || !aligned16(x_out) || (Cq && (!aligned16(wqkv_packed) || !aligned16(qkv_out)))) return CSSHIP_EALIGN;
  const bool CV = conv_g != nullptr;
  if (CV && (!conv_w || T <= 0 || !aligned16(conv_g) || conv_g == x_out)) return CSSHIP_EBADSHAPE;
  if (CV && conv_w->causal) return CSSHIP_EUNSUPPORTED;
  ChainArgs a{};
  a.x = x; a.xout = x_out; a.M = M; a.FF = d_ffn; a.NQKV = n_qkv; a.alpha = 0.5f; a.g = conv_g; a.T = T;
  a.stamps = g_csship_stamps;
  a.qkv_il8 = (flags & CSSHIP_CHAIN_QKV_IL8) ? 1 : 0;
  if (CV) a.cv = ConvW{conv_w->dw_w, conv_w->dw_b, conv_w->bn_g, conv_w->bn_b, conv_w->bn_mean, conv_w->bn_var,
                       conv_w->pw2_w, conv_w->pw2_b};
  auto conv = [](const csship_ffn_weights* w) {
    return FfnW{w->ln_g, w->ln_b, reinterpret_cast<const bf16x8*>(w->w1), w->b1, reinterpret_cast<const bf16x8*>(w->w2), w->b2};
  };
  if (A) { a.fa = conv(ffn_a); a.fin_g = final_ln_g; a.fin_b = final_ln_b; }
  if (Bs) a.fb = conv(ffn_b);
  if (Cq) { a.qln_g = qkv_ln_g; a.qln_b = qkv_ln_b; a.wqkv = reinterpret_cast<const bf16x8*>(wqkv_packed); a.bqkv = bqkv; a.qkv = qkv_out; }
  dim3 grid((M + BM - 1) / BM), block(NWAVES * 64);
  hipStream_t s = as_stream(stream);
  // the static form (chain_s_kernel): whole panels, q | k | v of 768 columns, d_ffn up to 2048, every parameter 16-byte
  // aligned (they are staged into LDS by float4 loads).  CSSHIP_CHAIN_STATIC=0: the first form everywhere (A/B timing)
  static const bool use_static = [] { const char* e = getenv("CSSHIP_CHAIN_STATIC"); return e == nullptr || atoi(e) != 0; }();
  auto al16 = [](const void* p) { return p == nullptr || aligned16(p); };
  const bool st_ok = use_static && !CV && (M % BM) == 0 && d_ffn <= CS_FFMAX && (!Cq || n_qkv == 768) &&
                     (!A || (al16(ffn_a->ln_g) && al16(ffn_a->ln_b) && aligned16(ffn_a->b1) && aligned16(ffn_a->b2) &&
                             aligned16(final_ln_g) && aligned16(final_ln_b) && (ffn_a->ln_g == nullptr) == (ffn_a->ln_b == nullptr))) &&
                     (!Bs || (al16(ffn_b->ln_g) && al16(ffn_b->ln_b) && aligned16(ffn_b->b1) && aligned16(ffn_b->b2) &&
                              (ffn_b->ln_g == nullptr) == (ffn_b->ln_b == nullptr))) &&
                     (!Cq || (al16(qkv_ln_g) && al16(qkv_ln_b) && aligned16(bqkv) && bqkv != nullptr));
  if (train != nullptr) {
    const csship_dropout_cfg none{nullptr, 0u, 0.f};
    auto cfg = [&](const csship_dropout_cfg* c) { return (c && c->seed && c->p > 0.f) ? *c : none; };
    a.da_h = cfg(train->drop_a_hidden); a.da_o = cfg(train->drop_a_out);
    a.db_h = cfg(train->drop_b_hidden); a.db_o = cfg(train->drop_b_out);
    a.pre_ln_out = train->pre_ln_out; a.ln_out = train->ln_out;
    const bool bnp = train->bn_z != nullptr;
    if (bnp) {
      if (!train->bn_x2 || !train->bn_scale_shift || !train->bn_pw2_w || !train->bn_pw2_b || !train->bn_out) return CSSHIP_ENULL;
      if (!aligned16(train->bn_z) || !aligned16(train->bn_x2) || !aligned16(train->bn_scale_shift) || !aligned16(train->bn_out)) return CSSHIP_EALIGN;
      if ((int64_t)M * DM > 0xffffffffLL) return CSSHIP_EBADSHAPE;
      a.bn_z = train->bn_z; a.bn_x2 = train->bn_x2; a.bn_ss = train->bn_scale_shift; a.bn_pw2_w = train->bn_pw2_w; a.bn_pw2_b = train->bn_pw2_b;
      a.d_bn = cfg(train->drop_bn); a.bn_out = train->bn_out;
    }
    if ((A && !ffn_a->ln_g) || (Bs && !ffn_b->ln_g)) return CSSHIP_EUNSUPPORTED;  // folded LayerNorms belong to the inference packing
    if (st_ok && Bs && Cq && (!A || (a.pre_ln_out != nullptr && a.ln_out != nullptr))) {
      // the pipelined stages are OFF for the training form: with the dropout hashes riding in the W2 passes the kernel
      // needs 264 registers (8 spilled) and measured 10.92 against 10.87 ms per training step; CSSHIP_CHAIN_PIPE_TRAIN=1 runs it
      static const bool use_pipe_t = [] { const char* e = getenv("CSSHIP_CHAIN_PIPE_TRAIN"); return e != nullptr && atoi(e) != 0; }();
      const bool pipe_t = use_pipe_t && d_ffn >= 2 * DM;
      if (bnp && !A) return CSSHIP_EUNSUPPORTED;
      if (bnp && pipe_t) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, true, true, true, false, true, true>), grid, block, 0, s, a);
      else if (bnp) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, true, true, true, false, false, true>), grid, block, 0, s, a);
      else if (A && pipe_t) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, true, true, true, false, true>), grid, block, 0, s, a);
      else if (pipe_t) hipLaunchKernelGGL((chain_s_kernel<NWAVES, false, true, true, true, false, true>), grid, block, 0, s, a);
      else if (A) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, true, true, true, false>), grid, block, 0, s, a);
      else hipLaunchKernelGGL((chain_s_kernel<NWAVES, false, true, true, true, false>), grid, block, 0, s, a);
      CSSHIP_CHECK_LAUNCH();
      return CSSHIP_OK;
    }
    if (bnp) return CSSHIP_EUNSUPPORTED;   // (the batch-statistics prologue lives in the static form: whole 64-row panels)
    if (A && Bs && Cq) hipLaunchKernelGGL((chain_kernel<NWAVES, false, true, true, true, true>), grid, block, 0, s, a);
    else if (!A && Bs && Cq) hipLaunchKernelGGL((chain_kernel<NWAVES, false, false, true, true, true>), grid, block, 0, s, a);
    else return CSSHIP_EUNSUPPORTED;
    CSSHIP_CHECK_LAUNCH();
    return CSSHIP_OK;
  }
  // CSSHIP_CHAIN=8 / 4 run the second form of the kernel (8 or 4 waves) for A/B timing; measured at M = 12 800:
  // first form 50.1 us, second form 54.0 us (8 waves), 61.4 us (4 waves) - see the note above chain2_kernel
  static const int form = [] { const char* e = getenv("CSSHIP_CHAIN"); return e ? atoi(e) : 1; }();
  if (form != 1 && !CV && (A || Bs)) {
    if ((A && !ffn_a->ln_g) || (Bs && !ffn_b->ln_g) || (Cq && !qkv_ln_g) || a.qkv_il8) return CSSHIP_EUNSUPPORTED;  // second form: plain weights only
#define C2_LAUNCH(W)                                                                                            \
  if (A && Bs && Cq) hipLaunchKernelGGL((chain2_kernel<W, true, true, true>), grid, dim3(W * 64), 0, s, a);      \
  else if (!A && Bs && Cq) hipLaunchKernelGGL((chain2_kernel<W, false, true, true>), grid, dim3(W * 64), 0, s, a); \
  else if (A && !Bs && !Cq) hipLaunchKernelGGL((chain2_kernel<W, true, false, false>), grid, dim3(W * 64), 0, s, a); \
  else return CSSHIP_EUNSUPPORTED;
    if (form == 8) { C2_LAUNCH(8) } else { C2_LAUNCH(4) }
#undef C2_LAUNCH
    CSSHIP_CHECK_LAUNCH();
    return CSSHIP_OK;
  }
  if (st_ok && (A || Bs) && (Bs || !Cq)) {
    const bool il8 = a.qkv_il8 != 0;
    static const bool use_pipe = [] { const char* e = getenv("CSSHIP_CHAIN_PIPE"); return e == nullptr || atoi(e) != 0; }();
    const bool pipe = use_pipe && d_ffn >= 2 * DM;
    if (A && Bs && Cq && il8 && pipe) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, true, true, false, true, true>), grid, block, 0, s, a);
    else if (!A && Bs && Cq && il8 && pipe) hipLaunchKernelGGL((chain_s_kernel<NWAVES, false, true, true, false, true, true>), grid, block, 0, s, a);
    else if (A && !Bs && !Cq && pipe) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, false, false, false, false, true>), grid, block, 0, s, a);
    else if (A && Bs && Cq && il8) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, true, true, false, true>), grid, block, 0, s, a);
    else if (A && Bs && Cq) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, true, true, false, false>), grid, block, 0, s, a);
    else if (!A && Bs && Cq && il8) hipLaunchKernelGGL((chain_s_kernel<NWAVES, false, true, true, false, true>), grid, block, 0, s, a);
    else if (!A && Bs && Cq) hipLaunchKernelGGL((chain_s_kernel<NWAVES, false, true, true, false, false>), grid, block, 0, s, a);
    else if (A && !Bs && !Cq) hipLaunchKernelGGL((chain_s_kernel<NWAVES, true, false, false, false, false>), grid, block, 0, s, a);
    else return CSSHIP_EUNSUPPORTED;
    CSSHIP_CHECK_LAUNCH();
    return CSSHIP_OK;
  }
  if (!CV && A && Bs && Cq) hipLaunchKernelGGL((chain_kernel<NWAVES, false, true, true, true>), grid, block, 0, s, a);
  else if (!CV && !A && Bs && Cq) hipLaunchKernelGGL((chain_kernel<NWAVES, false, false, true, true>), grid, block, 0, s, a);
  else if (!CV && A && !Bs && !Cq) hipLaunchKernelGGL((chain_kernel<NWAVES, false, true, false, false>), grid, block, 0, s, a);
  else if (CV && A && Bs && Cq) hipLaunchKernelGGL((chain_kernel<NWAVES, true, true, true, true>), grid, block, 0, s, a);
  else if (CV && A && !Bs && !Cq) hipLaunchKernelGGL((chain_kernel<NWAVES, true, true, false, false>), grid, block, 0, s, a);
  else return CSSHIP_EUNSUPPORTED;
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_embed(const float* f, int64_t ldf, const float* mvn_mean, const float* mvn_rinv,
                            const uint16_t* w_packed, const float* bias, const float* ln_g, const float* ln_b, float* out,
                            int32_t B, int32_t T, int32_t idim, int32_t d_model, csship_stream_t stream) {
  if (!f || !mvn_mean || !mvn_rinv || !w_packed || !bias || !ln_g || !ln_b || !out) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || idim <= 0 || ldf < idim) return CSSHIP_EBADSHAPE;
  if (d_model != DM) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(w_packed) || !aligned16(out) || !aligned16(bias) || !aligned16(ln_g) || !aligned16(ln_b)) return CSSHIP_EALIGN;
  const int Kp = (idim + DM - 1) / DM * DM;
  EmbedArgs a{f, mvn_mean, mvn_rinv, reinterpret_cast<const bf16x8*>(w_packed), bias, ln_g, ln_b, out, B * T, T, idim, Kp, ldf,
              nullptr, 0, nullptr, csship_dropout_cfg{nullptr, 0u, 0.f}};
  hipLaunchKernelGGL((embed_kernel<NWAVES>), dim3((B * T + BM - 1) / BM), dim3(NWAVES * 64), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_embed_train(const float* f, int64_t ldf, const float* mvn_mean, const float* mvn_rinv,
                                  const uint16_t* w_packed, const float* bias, const float* ln_g, const float* ln_b, float* out,
                                  uint16_t* fn_bf16, int32_t ld_fn, float* e_out, int32_t B, int32_t T, int32_t idim,
                                  int32_t d_model, const csship_dropout_cfg* drop, csship_stream_t stream) {
  if (!f || !mvn_mean || !mvn_rinv || !w_packed || !bias || !ln_g || !ln_b || !out || !fn_bf16 || !e_out) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || idim <= 0 || ldf < idim || ld_fn < idim || (ld_fn & 7) != 0 || (int64_t)B * T * DM > 0xffffffffLL)
    return CSSHIP_EBADSHAPE;
  if (d_model != DM) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(w_packed) || !aligned16(out) || !aligned16(bias) || !aligned16(ln_g) || !aligned16(ln_b) || !aligned16(fn_bf16) ||
      !aligned16(e_out))
    return CSSHIP_EALIGN;
  const int Kp = (idim + DM - 1) / DM * DM;
  if (ld_fn > Kp) return CSSHIP_EBADSHAPE;   // (the kernel writes fn while it builds its K panels)
  const bool dropping = drop && drop->seed && drop->p > 0.f;
  EmbedArgs a{f, mvn_mean, mvn_rinv, reinterpret_cast<const bf16x8*>(w_packed), bias, ln_g, ln_b, out, B * T, T, idim, Kp, ldf,
              fn_bf16, ld_fn, e_out, dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f}};
  TimelineScope tl(CSSHIP_TL_EMBED, as_stream(stream));
  hipLaunchKernelGGL((embed_kernel<NWAVES, true>), dim3((B * T + BM - 1) / BM), dim3(NWAVES * 64), 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_outproj_glu(const uint16_t* attn_out, const uint16_t* wo_packed, const float* bo, const float* x,
                                  float* x_out, const csship_conv_weights* conv_w, float* g_out, int32_t M,
                                  int32_t d_model, csship_stream_t stream) {
  if (!attn_out || !wo_packed || !x || !x_out || !conv_w || !g_out) return CSSHIP_ENULL;
  if (!conv_w->ln_g || !conv_w->ln_b || !conv_w->pw1_w || !conv_w->pw1_b) return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (d_model != DM) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(attn_out) || !aligned16(wo_packed) || !aligned16(x) || !aligned16(x_out) || !aligned16(g_out) ||
      !aligned16(conv_w->ln_g) || !aligned16(conv_w->ln_b))
    return CSSHIP_EALIGN;
  LinArgs la{attn_out, nullptr, nullptr, reinterpret_cast<const bf16x8*>(wo_packed), bo, x, x_out, M, DM, DM, DM, 1.f,
             g_csship_stamps, conv_w->ln_g, conv_w->ln_b, conv_w->pw1_w, conv_w->pw1_b, g_out};
  return launch_linear<CSSHIP_PRO_BF16, CSSHIP_EPI_RESIDUAL>(la, as_stream(stream));
}
