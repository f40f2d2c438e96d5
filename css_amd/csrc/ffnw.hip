// FeedForward as ONE row-panel kernel for d_model 256 or 512 (Conformer-large: d_model 512, d_ffn 2048), in two operand
// formats (css/models/conformer.py:322-326 with the 0.5 residual of :355/:358 and the block-final LayerNorm of :360):
//
//   ffnw_kernel   bf16 operands, v_mfma_f32_16x16x32_bf16     - the parity path of the any-width encoder
//   ffnmx_kernel  MXFP8 operands (OCP e4m3 elements + one e8m0 scale per 32 consecutive K elements, BOTH operands),
//                 v_mfma_scale_f32_16x16x128_f8f6f4: twice the bf16 MFMA rate, half the weight bytes through the CU's
//                 L1 and half the LDS operand bytes.  Opt-in (csship_conformer_weights.fp8_ffn): on seeded random
//                 weights block-scaled fp8 FeedForwards cost 2.6e-3 mask MSE, so this is not the parity path.
//
// Same structure as panel.hip's ffn_kernel (64 token rows per workgroup, 8 waves, transposed products with the WEIGHTS
// as the MFMA A operand streamed global -> VGPR in fragment order, the hidden activation only as 256-column chunks in
// LDS), generalised over the row width: the W1 pass reduces over D (hidden tiles: 2 per wave), the W2 pass over a
// 256-wide hidden chunk (output tiles: D / 128 per wave).
//
// MXFP8 operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 (found with tools/micro/mx_probe.hip, pinned with exact
// integer data by tools/micro/mx_layout.hip): lane l = (row-or-column l & 15, g = l >> 4) holds TWO 16-byte pieces of
// its row, k = 16 g + [0, 16) and k = 64 + 16 g + [0, 16); the scale of 32-block b of row i is one byte (selected by
// op_sel) of the scale VGPR of lane i + 16 b.  So a lane reads its row's bytes [16g, 16g+16) and [64+16g, 64+16g+16)
// of every 128-byte k-step, and the scale dword of (row, block b) sits with lane (row, b).
#include "common.hpp"

namespace {

#ifndef MX_ABL
#define MX_ABL 0   // diagnostic builds only (tools/ffn_mx_ablate.sh): bit 0 no scaled MFMAs, 1 no weight / scale loads in the loop,
                   // 2 no hidden quantisation (max, scale, convert, LDS store), 3 no LayerNorm + quantisation prologue,
                   // 4 no row epilogue (finish_rows), 5 no operand fragment reads from LDS
#endif
constexpr int BM = 64;   // token rows per workgroup
constexpr int PDM = 2;   // MXFP8 weight prefetch distance (128-deep k-steps)
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int ROWB>
DEV int poff(int row, int c16) { return row * ROWB + ((c16 ^ (row & 15)) << 4); }

// ------------------------------------------------------------------------------------------------ row phases
// Thread (r = tid >> 3, part = tid & 7) owns the float4 pieces (i * 8 + part), i < D / 32, of row r: the 8 lanes of a
// row read one full 128-byte line per load instruction, LayerNorm statistics are in-lane sums + 3 DPP steps.
template <int D>
DEV void load_rows(float4 (&v)[D / 32], const float* x, int row, int part) {
  const float* xr = x + (size_t)row * D;
#pragma unroll
  for (int i = 0; i < D / 32; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (i * 8 + part) * 4);
}
template <int D>
DEV void layernorm_rows(float4 (&v)[D / 32], const float* gam, const float* bet, int part) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < D / 32; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = sum8(s) * (1.f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < D / 32; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(sum8(ss) * (1.f / D) + 1e-5f);
#pragma unroll
  for (int i = 0; i < D / 32; ++i) {
    const float4 gm = *reinterpret_cast<const float4*>(gam + (i * 8 + part) * 4);
    const float4 bt = *reinterpret_cast<const float4*>(bet + (i * 8 + part) * 4);
    v[i].x = v[i].x * rstd * gm.x + bt.x; v[i].y = v[i].y * rstd * gm.y + bt.y;
    v[i].z = v[i].z * rstd * gm.z + bt.z; v[i].w = v[i].w * rstd * gm.w + bt.w;
  }
}

// out = x + alpha * (Ys row) [-> LayerNorm]; Ys: f32 [64][D + 4] holding FFN output + b2 (parked accumulators)
template <int D, bool FINAL_LN>
DEV void finish_rows(const float* ys, const float* x, float* out, const float* fin_g, const float* fin_b, int m0, int M,
                     float alpha, int tid) {
  const int r = tid >> 3, part = tid & 7, row = m0 + r;
  float4 v[D / 32];
  load_rows<D>(v, x, min(row, M - 1), part);
#pragma unroll
  for (int i = 0; i < D / 32; ++i) {
    const float4 y = *reinterpret_cast<const float4*>(ys + r * (D + 4) + (i * 8 + part) * 4);
    v[i] = make_float4(v[i].x + alpha * y.x, v[i].y + alpha * y.y, v[i].z + alpha * y.z, v[i].w + alpha * y.w);
  }
  if constexpr (FINAL_LN) layernorm_rows<D>(v, fin_g, fin_b, part);
  if (row < M) {
#pragma unroll
    for (int i = 0; i < D / 32; ++i) *reinterpret_cast<float4*>(out + (size_t)row * D + (i * 8 + part) * 4) = v[i];
  }
}

template <int NTO, int D>
DEV void park_out(float* ys, const f32x4 (&o)[NTO][4], const float4 (&b2v)[NTO], int wave, int g, int n) {
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
      *reinterpret_cast<float4*>(ys + (rt * 16 + n) * (D + 4) + (wave * NTO + t) * 16 + 4 * g) =
          make_float4(o[t][rt][0] + b2v[t].x, o[t][rt][1] + b2v[t].y, o[t][rt][2] + b2v[t].z, o[t][rt][3] + b2v[t].w);
}

// ================================================================================================ bf16
struct FfnwArgs {
  const float* x;
  float* out;
  const float *ln_g, *ln_b;
  const bf16x8* w1;  // fragment order [FF/16][D/32][64]  (css_amd.ops.pack_frag)
  const float* b1;
  const bf16x8* w2;  // fragment order [D/16][FF/32][64]
  const float* b2;
  const float *fin_g, *fin_b;
  int M, FF;
  float alpha;
};

// acc[t][rt] += W(tile t)[16 x 32*KSTEPS] . panel^T;  weights of k-steps 0..PDW-1 arrive in wq, on return wq holds
// k-steps 0..PDW-1 of the NEXT stream (wn / nstride, NTN tiles), as in panel.hip's mma_pass.  PDW = prefetch distance
// in k-steps; it must divide KSTEPS of every pass (a k-step lives in ring slot k-step % PDW across pass boundaries)
template <int NT, int NTN, int KSTEPS, int ROWB, int PDW, int NTM>
DEV void pass_bf16(f32x4 (&acc)[NT][4], const unsigned char* panel, int n, int g, const bf16x8* w, int tstride,
                   const bf16x8* wn, int nstride, bool has_next, bf16x8 (&wq)[PDW][NTM]) {
  bf16x8 xa[2][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) xa[0][rt] = *reinterpret_cast<const bf16x8*>(panel + poff<ROWB>(rt * 16 + n, g));
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    bf16x8 wc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wc[t] = wq[ks % PDW][t];
    if (ks + PDW < KSTEPS) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wq[ks % PDW][t] = w[(size_t)(t * tstride + ks + PDW) * 64];
    } else if (has_next) {
#pragma unroll
      for (int t = 0; t < NTN; ++t) wq[ks % PDW][t] = wn[(size_t)(t * nstride + ks + PDW - KSTEPS) * 64];
    }
    if (ks + 1 < KSTEPS) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        xa[(ks + 1) & 1][rt] = *reinterpret_cast<const bf16x8*>(panel + poff<ROWB>(rt * 16 + n, (ks + 1) * 4 + g));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], xa[ks & 1][rt], acc[t][rt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int D, bool FINAL_LN>
__global__ __launch_bounds__(512) void ffnw_kernel(FfnwArgs a) {
  constexpr int KS1 = D / 32, NTO = D / 128, XROWB = 2 * D, HROWB = 512, HBYTES = BM * HROWB;
  constexpr int PDW = 4, NTM = NTO > 2 ? NTO : 2;  // PDW must divide both k-step counts (ring slot = k-step % PDW across passes)
  constexpr int PANELS = BM * XROWB + 2 * HBYTES, YSB = BM * (D + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[PANELS > YSB ? PANELS : YSB];
  // the hidden bias from LDS (staged once, unconditionally), the ring refilled unconditionally behind the last chunk and
  // the output bias requested after the loop: a run-time predicate around a global load inside the chunk loop makes
  // hipcc's s_waitcnt insertion assume the path without it, and the waits of the following k-steps then cover weight
  // fragments requested a moment ago (round 5, panel.hip chain_s_kernel)
  constexpr int FFMAX = 2048;
  __shared__ __attribute__((aligned(16))) float B1s[FFMAX + 256];
  unsigned char* const Xs = smem;
  unsigned char* const Hs = smem + BM * XROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM, nchunk = a.FF / 256, ks2 = a.FF / 32, rot = blockIdx.x % nchunk;
  const bool b1_lds = a.FF <= FFMAX;   // (wave-uniform; wider FeedForwards keep the global loads)
  {
    const int q1 = a.FF >> 2;
    const bool mine = b1_lds && tid < q1;
    const float4 bv = *reinterpret_cast<const float4*>(a.b1 + (mine ? tid : 0) * 4);
    *reinterpret_cast<float4*>(B1s + (mine ? tid * 4 : FFMAX + (lane & 63) * 4)) = bv;
  }
  const float* const b1p = b1_lds ? B1s : a.b1;
  const bf16x8* w1s = a.w1 + (size_t)(wave * 2) * KS1 * 64 + lane;    // + chunk c: c * 16 tiles
  const bf16x8* w2s = a.w2 + (size_t)(wave * NTO) * ks2 * 64 + lane;  // + chunk c: k-steps 8c ..
  bf16x8 wq[PDW][NTM];
  {
    const bf16x8* w1c = w1s + (size_t)rot * 16 * KS1 * 64;
#pragma unroll
    for (int k = 0; k < PDW; ++k)
#pragma unroll
      for (int t = 0; t < 2; ++t) wq[k][t] = w1c[(size_t)(t * KS1 + k) * 64];
    __builtin_amdgcn_sched_barrier(0);
  }
  {  // LayerNorm -> bf16 operand panel
    const int r = tid >> 3, part = tid & 7;
    float4 v[D / 32];
    load_rows<D>(v, a.x, min(m0 + r, a.M - 1), part);
    layernorm_rows<D>(v, a.ln_g, a.ln_b, part);
#pragma unroll
    for (int i = 0; i < D / 32; ++i) {
      const int c4 = i * 8 + part;
      *reinterpret_cast<uint2*>(Xs + poff<XROWB>(r, c4 >> 1) + (c4 & 1) * 8) = make_uint2(pack_bf2(v[i].x, v[i].y), pack_bf2(v[i].z, v[i].w));
    }
  }
  __syncthreads();
  f32x4 o[NTO][4];
#pragma unroll
  for (int t = 0; t < NTO; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 b2v[NTO];
#pragma unroll
  for (int t = 0; t < NTO; ++t) b2v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ci = 0; ci < nchunk; ++ci) {
    const int c = (ci + rot) % nchunk;
    f32x4 h[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // the bias is the accumulators' initial value
      const float4 bb = *reinterpret_cast<const float4*>(b1p + c * 256 + (wave * 2 + t) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    const bf16x8* w1c = w1s + (size_t)c * 16 * KS1 * 64;
    const bf16x8* w2c = w2s + (size_t)c * 8 * 64;
    pass_bf16<2, NTO, KS1, XROWB, PDW, NTM>(h, Xs, n, g, w1c, KS1, w2c, ks2, true, wq);
    unsigned char* hs = Hs + (ci & 1) * HBYTES;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int hc = (wave * 2 + t) * 16 + 4 * g;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        *reinterpret_cast<uint2*>(hs + poff<HROWB>(rt * 16 + n, hc >> 3) + (hc & 7) * 2) =
            make_uint2(relu_bf2(pack_bf2(h[t][rt][0], h[t][rt][1])), relu_bf2(pack_bf2(h[t][rt][2], h[t][rt][3])));
    }
    __syncthreads();
    const bf16x8* w1n = w1s + (size_t)((c + 1) % nchunk) * 16 * KS1 * 64;
    pass_bf16<NTO, 2, 8, HROWB, PDW, NTM>(o, hs, n, g, w2c, ks2, w1n, KS1, true, wq);  // (behind the last chunk: a harmless re-read)
  }
#pragma unroll
  for (int t = 0; t < NTO; ++t) b2v[t] = *reinterpret_cast<const float4*>(a.b2 + (wave * NTO + t) * 16 + 4 * g);
  __syncthreads();  // every wave is done with the panels: the region becomes the f32 staging tile
  float* ys = reinterpret_cast<float*>(smem);
  park_out<NTO, D>(ys, o, b2v, wave, g, n);
  __syncthreads();
  finish_rows<D, FINAL_LN>(ys, a.x, a.out, a.fin_g, a.fin_b, m0, a.M, a.alpha, tid);
}

// ================================================================================================ wide row-panel Linears
// The two Linears of the attention stage at any width as row-panel kernels (the any-width path ran them as LayerNorm +
// tiled GEMM: at d_model 512 the q | k | v projection is a 12 800 x 512 x 1 536 product whose tiles are 16 k-steps
// deep - prologue / epilogue bound):
//   lnlin_kernel   out_bf16[M, N] = bf16( LayerNorm(x) W^T + b )      conformer.py:222-225, N = 3 d_model, rows of W in
//                  the 8-feature interleave of ops.pack_frag_il8 (a lane's 2 x 4 results = 8 consecutive features:
//                  one 16-byte store per row tile)
//   linres_kernel  out[M, D] = res + a_bf16 W^T + b                    conformer.py:254, 356 (attention output projection)
// Same pass structure as the FeedForward above (NT = 2 tiles per wave and 256-feature pass, weights streamed in
// fragment order PDW k-steps ahead, continuing across passes).
struct LinwArgs {
  const void* a;        // f32 [M, D] (lnlin) or bf16 [M, D] (linres)
  const float *ln_g, *ln_b;
  const bf16x8* w;      // fragment order [N/16][D/32][64]
  const float* bias;
  const float* res;     // linres: f32 [M, N]
  void* out;
  int M, N;
};

// FULL (M % 64 == 0, N <= 2048): bias from an LDS copy, no row predicates, the ring refilled unconditionally behind the
// last pass, and linres requests a pass's residual pieces together, ahead of the pass (load - add - store per piece was a
// round trip per piece): every s_waitcnt of the pass loop is exact (round 5, panel.hip chain_s_kernel)
template <int D, bool LN_IN, bool FULL>
__global__ __launch_bounds__(512) void linw_kernel(LinwArgs a) {
  constexpr int KSN = D / 32, XROWB = 2 * D, PDW = 4;
  __shared__ __attribute__((aligned(16))) unsigned char Xs[BM * XROWB];
  __shared__ __attribute__((aligned(16))) float Bls[FULL ? 2048 : 4];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM, npass = a.N / 256;
  const float* biasp = a.bias;
  if constexpr (FULL) {   // N / 4 <= 512 float4: one per thread, clamped (threads beyond N / 4 rewrite the last piece)
    const int j = min(tid, (a.N >> 2) - 1);
    *reinterpret_cast<float4*>(Bls + j * 4) = *reinterpret_cast<const float4*>(a.bias + j * 4);
    biasp = Bls;
  }
  const bf16x8* ws = a.w + (size_t)(wave * 2) * KSN * 64 + lane;  // + pass p: 16 tiles
  bf16x8 wq[PDW][2];
#pragma unroll
  for (int k = 0; k < PDW; ++k)
#pragma unroll
    for (int t = 0; t < 2; ++t) wq[k][t] = ws[(size_t)(t * KSN + k) * 64];
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (LN_IN) {
    const int r = tid >> 3, part = tid & 7;
    float4 v[D / 32];
    load_rows<D>(v, reinterpret_cast<const float*>(a.a), min(m0 + r, a.M - 1), part);
    layernorm_rows<D>(v, a.ln_g, a.ln_b, part);
#pragma unroll
    for (int i = 0; i < D / 32; ++i) {
      const int c4 = i * 8 + part;
      *reinterpret_cast<uint2*>(Xs + poff<XROWB>(r, c4 >> 1) + (c4 & 1) * 8) = make_uint2(pack_bf2(v[i].x, v[i].y), pack_bf2(v[i].z, v[i].w));
    }
  } else {
    const unsigned short* ab = reinterpret_cast<const unsigned short*>(a.a);
    for (int i = tid; i < BM * (D / 8); i += 512) {
      const int r = i / (D / 8), c = i % (D / 8);
      *reinterpret_cast<uint4*>(Xs + poff<XROWB>(r, c)) = *reinterpret_cast<const uint4*>(ab + (size_t)min(m0 + r, a.M - 1) * D + c * 8);
    }
  }
  __syncthreads();
  for (int p = 0; p < npass; ++p) {
    const int tile0 = p * 16 + wave * 2;
    f32x4 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {  // the bias is the accumulators' initial value
      const float4 bb = *reinterpret_cast<const float4*>(biasp + (tile0 + t) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    const bf16x8* wc = ws + (size_t)p * 16 * KSN * 64;
    float4 rq[2][4];   // FULL linres: the residual pieces of this pass, requested now
    if constexpr (FULL && !LN_IN) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          rq[t][rt] = *reinterpret_cast<const float4*>(a.res + (size_t)(m0 + rt * 16 + n) * a.N + (tile0 + t) * 16 + 4 * g);
    }
    if constexpr (FULL)   // behind the last pass the ring re-reads pass 0 (harmless): the loads stay unconditional
      pass_bf16<2, 2, KSN, XROWB, PDW, 2>(acc, Xs, n, g, wc, KSN, p + 1 < npass ? wc + (size_t)16 * KSN * 64 : ws, KSN, true, wq);
    else
      pass_bf16<2, 2, KSN, XROWB, PDW, 2>(acc, Xs, n, g, wc, KSN, wc + (size_t)16 * KSN * 64, KSN, p + 1 < npass, wq);
    if constexpr (LN_IN) {  // interleaved rows: 8 consecutive features per lane, one 16-byte bf16 store per row tile
      unsigned short* o = reinterpret_cast<unsigned short*>(a.out);
      const int f = tile0 * 16 + 8 * g;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = m0 + rt * 16 + n;
        if (FULL || row < a.M)
          *reinterpret_cast<uint4*>(o + (size_t)row * a.N + f) =
              make_uint4(pack_bf2(acc[0][rt][0], acc[0][rt][1]), pack_bf2(acc[0][rt][2], acc[0][rt][3]),
                         pack_bf2(acc[1][rt][0], acc[1][rt][1]), pack_bf2(acc[1][rt][2], acc[1][rt][3]));
      }
    } else {  // natural rows: 4 consecutive features per lane and tile, f32 residual add
      float* o = reinterpret_cast<float*>(a.out);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int f = (tile0 + t) * 16 + 4 * g;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          const int row = m0 + rt * 16 + n;
          if (FULL || row < a.M) {
            float4 r4;
            if constexpr (FULL) r4 = rq[t][rt];
            else r4 = *reinterpret_cast<const float4*>(a.res + (size_t)row * a.N + f);
            *reinterpret_cast<float4*>(o + (size_t)row * a.N + f) =
                make_float4(r4.x + acc[t][rt][0], r4.y + acc[t][rt][1], r4.z + acc[t][rt][2], r4.w + acc[t][rt][3]);
          }
        }
      }
    }
  }
}

// ================================================================================================ MXFP8
struct FfnMxArgs {
  const float* x;
  float* out;
  const float *ln_g, *ln_b;
  const uint4* w1;       // e4m3 bytes in fragment order [FF/16][D/128][2 halves][64 lanes] x 16 B  (css_amd.ops.pack_mx)
  const unsigned* w1s;   // e8m0 scales [FF/16][D/256][64 lanes] dwords: byte q = k-step 2*kp + q of the lane's block
  const float* b1;
  const uint4* w2;       // [D/16][FF/128][2][64] x 16 B
  const unsigned* w2s;   // [D/16][FF/256][64] dwords
  const float* b2;
  const float *fin_g, *fin_b;
  int M, FF;
  float alpha;
};

// e8m0 scale byte of a block with maximum magnitude amax (OCP MX: shared exponent floor(log2(amax)) - 8 for e4m3) and
// the exact reciprocal 2^-(shared exponent) as a float
DEV unsigned mx_scale_byte(float amax) {
  const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 8;  // biased exponent of amax, minus emax(e4m3) = 8
  return (unsigned)max(e, 0);
}
DEV float mx_inv_scale(unsigned sb) { return __uint_as_float((254u - sb) << 23); }  // 2^(127 - sb); sb <= 246 here
DEV unsigned mx_pack4(float a, float b, float c, float d, float inv) {
  // e4m3 saturates at +-448: clamp first (the conversion's own overflow behaviour is mode dependent)
  a = __builtin_fminf(__builtin_fmaxf(a * inv, -448.f), 448.f);
  b = __builtin_fminf(__builtin_fmaxf(b * inv, -448.f), 448.f);
  c = __builtin_fminf(__builtin_fmaxf(c * inv, -448.f), 448.f);
  d = __builtin_fminf(__builtin_fmaxf(d * inv, -448.f), 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (unsigned)r;
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
DEV i32x8 frag8(const i32x4& lo, const i32x4& hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
// Weight streams are addressed as (wave-uniform pointer, kept in SGPRs) + (this lane's byte offset, ONE VGPR): with
// per-lane 64-bit pointers hipcc keeps a VGPR pair per (tile, k-step) it prefetches - ~100 registers of addresses at
// d_model 512, which is what pushed the kernel into scratch.
// Loads go through a buffer descriptor: address = descriptor base + (wave-uniform byte offset: an SGPR) + (lane * 16:
// one VGPR) - the addressing form cannot degrade into per-lane 64-bit pointers.
typedef __attribute__((ext_vector_type(4))) unsigned v4u_t;
struct WStream {
  __amdgpu_buffer_rsrc_t rs;
  unsigned base;      // wave-uniform byte offset of (tile 0, k-step 0, half 0) of this stream
  unsigned lane_off;  // lane * 16
};
DEV i32x8 load_frag(const WStream& w, unsigned piece) {  // two 16-byte pieces 1 KB apart -> one 8-register operand
  const unsigned so = w.base + piece * 1024u;
  const v4u_t lo = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.lane_off, so, 0);
  const v4u_t hi = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.lane_off, so + 1024u, 0);
  return frag8(__builtin_bit_cast(i32x4, lo), __builtin_bit_cast(i32x4, hi));
}
// fp8 panels: rows of ROWB bytes PADDED by 32 (no XOR swizzle).  Row r, 16-byte piece c sits in bank slot
// (2 r + c) mod 16: for every lane group of a ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... = eight
// rows at lane group g and eight at g + 1) the 16 slots are distinct, i.e. conflict free - and, unlike the swizzle,
// the address of (k-step, half) is the row's base + a compile-time constant: ONE address register per row tile instead
// of one per (row tile, k-step, half), ~80 registers at d_model 512.
template <int ROWB>
DEV int poff8(int row, int byte) { return row * (ROWB + 32) + byte; }
template <int ROWB>
DEV i32x8 lds_frag(const unsigned char* panel8, int row, int ks, int g) {
  if (MX_ABL & 32) { i32x8 z = {row, ks, g, 0, 0, 0, 0, 0}; return z; }
  const unsigned char* p = panel8 + poff8<ROWB>(row, 16 * g);
  return frag8(*reinterpret_cast<const i32x4*>(p + ks * 128), *reinterpret_cast<const i32x4*>(p + ks * 128 + 64));
}
template <int OW, int OX>
DEV f32x4 mx_mfma(const i32x8& w, const i32x8& x, const f32x4& c, unsigned ws, unsigned xs) {
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w, x, c, 0, 0, OW, (int)ws, OX, (int)xs);
}
template <int KS>
DEV f32x4 mx_mfma_ks(const i32x8& w, const i32x8& x, const f32x4& c, unsigned ws, unsigned xs) {
  // weight scale dword: byte (k-step & 1); activation scale dword: byte (k-step & 3)
  if constexpr ((KS & 3) == 0) return mx_mfma<0, 0>(w, x, c, ws, xs);
  else if constexpr ((KS & 3) == 1) return mx_mfma<1, 1>(w, x, c, ws, xs);
  else if constexpr ((KS & 3) == 2) return mx_mfma<0, 2>(w, x, c, ws, xs);
  else return mx_mfma<1, 3>(w, x, c, ws, xs);
}

// One pass: acc[t][rt] += W(tile t)[16 x 128*KSTEPS] . panel8^T, both operands block-scaled.
//   panel8: e4m3 rows of ROWB (+ 32 padding) bytes, see poff8; xsc: dword (row, g) = the scale bytes of
//   blocks g, 4 + g, 8 + g, 12 + g of the row (byte = k-step).
//   w: this lane's pointer at (tile 0, k-step 0, half 0); tile t / k-step ks / half hf at w[(t * tstride + ks * 2 + hf) * 64].
//   wsc[t][kp]: scale dwords of the pass (loaded by the caller a pass ahead).
//   wq: ring of PDM k-steps of 8-register weight operands; k-steps 0..PDM-1 are prefetched by the caller, and on return
//   the ring holds k-steps 0..PDM-1 of the next stream (wn / nstride / NTN).  A slot is refilled right AFTER the MFMAs
//   that read it are issued (the MFMA operands ARE the ring registers: no copies, which at d_model 512 - 64 accumulator
//   registers - is what keeps the kernel inside 256 registers); the data is needed PDM k-steps later.
//   XB == 2: the activation fragments of the next k-step are read before this step's MFMAs; XB == 1 (d_model 512): after
//   them, into the same registers - the partner wave of the SIMD covers the read latency.
template <int NT, int NTN, int KSTEPS, int ROWB, int NTM, int XB, int STEP>
struct PassMx {
  DEV void run(f32x4 (&acc)[NT][4], const unsigned char* panel8, const unsigned (&xs)[4], int n, int g,
                      const WStream& w, int tstride, const WStream& wn, int nstride, bool has_next,
                      const unsigned (&wsc)[NTM][2], i32x8 (&wq)[PDM][NTM], i32x8 (&xa)[XB][4]) {
    if (XB == 2 && STEP + 1 < KSTEPS) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) xa[(STEP + 1) % XB][rt] = lds_frag<ROWB>(panel8, rt * 16 + n, STEP + 1, g);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        if (!(MX_ABL & 1)) acc[t][rt] = mx_mfma_ks<STEP>(wq[STEP % PDM][t], xa[STEP % XB][rt], acc[t][rt], wsc[t][STEP >> 1], xs[rt]);
        else asm volatile("" : "+v"(acc[t][rt]) : "v"(wq[STEP % PDM][t]), "v"(xa[STEP % XB][rt]));
    // hipcc otherwise interchanges the unrolled nest (all k-steps of row tile 0, then of row tile 1, ...), which makes
    // every k-step's weights and fragments live at once (at d_model 512: ~140 registers spilled per chunk).  An empty
    // asm that "touches" all accumulators of the step ties the independent chains together: k-step major order.
#pragma unroll
    for (int t = 0; t < NT; ++t)
      asm volatile("" : "+v"(acc[t][0]), "+v"(acc[t][1]), "+v"(acc[t][2]), "+v"(acc[t][3]));
    __builtin_amdgcn_sched_barrier(0);
    if (MX_ABL & 2) {
    } else if (STEP + PDM < KSTEPS) {
#pragma unroll
      for (int t = 0; t < NT; ++t) wq[STEP % PDM][t] = load_frag(w, (unsigned)(t * tstride + (STEP + PDM) * 2));
    } else if (has_next) {
#pragma unroll
      for (int t = 0; t < NTN; ++t) wq[STEP % PDM][t] = load_frag(wn, (unsigned)(t * nstride + (STEP + PDM - KSTEPS) * 2));
    }
    if (XB == 1 && STEP + 1 < KSTEPS) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) xa[0][rt] = lds_frag<ROWB>(panel8, rt * 16 + n, STEP + 1, g);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (STEP + 1 < KSTEPS)
      PassMx<NT, NTN, KSTEPS, ROWB, NTM, XB, STEP + 1>::run(acc, panel8, xs, n, g, w, tstride, wn, nstride, has_next, wsc, wq, xa);
  }
};
template <int NT, int NTN, int KSTEPS, int ROWB, int NTM, int XB>
DEV void pass_mx(f32x4 (&acc)[NT][4], const unsigned char* panel8, const unsigned* xsc, int n, int g, const WStream& w,
                 int tstride, const WStream& wn, int nstride, bool has_next, const unsigned (&wsc)[NTM][2], i32x8 (&wq)[PDM][NTM]) {
  unsigned xs[4];
  i32x8 xa[XB][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    xs[rt] = xsc[(rt * 16 + n) * 4 + g];
    xa[0][rt] = lds_frag<ROWB>(panel8, rt * 16 + n, 0, g);
  }
  PassMx<NT, NTN, KSTEPS, ROWB, NTM, XB, 0>::run(acc, panel8, xs, n, g, w, tstride, wn, nstride, has_next, wsc, wq, xa);
}

template <int D, bool FINAL_LN>
__global__ __launch_bounds__(512) void ffnmx_kernel(FfnMxArgs a) {
  // NH: the W2 pass runs as NH sub-passes of 2 output tiles per wave (d_model 512: the wave's 4 output tiles = 64
  // accumulator registers, and a 4-tile weight ring next to them does not fit 256 registers; the hidden operands are
  // read from LDS once per sub-pass instead)
  constexpr int KS1 = D / 128, KP1 = (KS1 + 1) / 2, NH = D / 256, NTO = 2 * NH, XROWB = D, HROWB = 256;
  constexpr int XBYTES = BM * (XROWB + 32), HBYTES = BM * (HROWB + 32);
  constexpr int NTM = 2, XB = 2;
  constexpr int SC_BYTES = BM * 16;  // [64 rows][4 g] dwords
  // d_model 512: the wave's second pair of output tiles (32 accumulator registers) LIVES IN LDS between its W2
  // sub-passes - [wave][tile][row tile][lane] 16-byte pieces, 64 KB - so that 64 + 32 + 32 + 64 registers of
  // accumulators, weight ring and operand fragments never have to coexist (hipcc spilled ~140 registers per chunk to
  // scratch otherwise: 109 us against 65 us for the bf16 kernel)
  constexpr int OPARK = NH > 1 ? 8 * 2 * 4 * 64 * 16 : 0;
  constexpr int PANELS = XBYTES + SC_BYTES + 2 * (HBYTES + SC_BYTES) + OPARK, YSB = BM * (D + 4) * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[PANELS > YSB ? PANELS : YSB];
  unsigned char* const X8 = smem;
  unsigned* const xsc = reinterpret_cast<unsigned*>(smem + XBYTES);
  unsigned char* const H8 = smem + XBYTES + SC_BYTES;               // buffer b at H8 + b * (HBYTES + SC_BYTES): panel | scales
  f32x4* const opark = reinterpret_cast<f32x4*>(smem + XBYTES + SC_BYTES + 2 * (HBYTES + SC_BYTES));
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, n = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: everything derived from it stays scalar
  const int m0 = blockIdx.x * BM, nchunk = a.FF / 256, ks2 = a.FF / 128, kp2 = a.FF / 256, rot = blockIdx.x % nchunk;
  const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
  // byte offsets (uniform): W1 tile = KS1 * 2 pieces of 1 KB, chunk c = 16 tiles; W2 tile = ks2 * 2 pieces, chunk c = k-steps 2c, 2c + 1
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.w1), 0, a.FF * D, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.w2), 0, a.FF * D, 0x00020000);
  const __amdgpu_buffer_rsrc_t rss1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.w1s), 0, a.FF / 16 * KP1 * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rss2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.w2s), 0, D / 16 * kp2 * 256, 0x00020000);
  const unsigned w1b = (unsigned)(wave * 2) * KS1 * 2 * 1024u, w2b = (unsigned)(wave * NTO) * ks2 * 2 * 1024u;
  const unsigned w1sb = (unsigned)(wave * 2) * KP1 * 256u, w2sb = (unsigned)(wave * NTO) * kp2 * 256u;
  auto sc_load = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned b, unsigned dword_row) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, lane4, b + dword_row * 256u, 0);
  };
  i32x8 wq[PDM][NTM];
  unsigned wsc1[NTM][2], wsc2[NH][NTM][2];
#pragma unroll
  for (int t = 0; t < NTM; ++t) {
    wsc1[t][0] = wsc1[t][1] = 0x7f7f7f7fu;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) wsc2[hf][t][0] = wsc2[hf][t][1] = 0x7f7f7f7fu;
  }
  {
    const WStream w1c{rs1, w1b + (unsigned)rot * 16 * KS1 * 2 * 1024u, lane16};
#pragma unroll
    for (int k = 0; k < PDM; ++k)
#pragma unroll
      for (int t = 0; t < 2; ++t) wq[k][t] = load_frag(w1c, (unsigned)(t * KS1 * 2 + k * 2));
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kp = 0; kp < KP1; ++kp) wsc1[t][kp] = sc_load(rss1, w1sb, (unsigned)((rot * 16 + t) * KP1 + kp));
    __builtin_amdgcn_sched_barrier(0);
  }
  {  // LayerNorm -> e4m3 operand panel + block scales.  Block i of row r = the i-th piece of its 8 lanes.
    const int r = tid >> 3, part = tid & 7;
    float4 v[D / 32];
    if (!(MX_ABL & 8)) {
    load_rows<D>(v, a.x, min(m0 + r, a.M - 1), part);
    layernorm_rows<D>(v, a.ln_g, a.ln_b, part);
    }
#pragma unroll
    for (int i = 0; i < ((MX_ABL & 8) ? 0 : D / 32); ++i) {
      const float am = max8(fmaxf(fmaxf(fabsf(v[i].x), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w))));
      const unsigned sb = mx_scale_byte(am);
      const int c4 = i * 8 + part;  // 4-byte piece index in the row
      *reinterpret_cast<unsigned*>(X8 + poff8<XROWB>(r, c4 * 4)) = mx_pack4(v[i].x, v[i].y, v[i].z, v[i].w, mx_inv_scale(sb));
      if ((i & 7) == part)  // block i: k-step i / 4, lane group i % 4
        reinterpret_cast<unsigned char*>(xsc)[(r * 4 + (i & 3)) * 4 + (i >> 2)] = (unsigned char)sb;
    }
  }
  __syncthreads();
  f32x4 o[2][4];   // output tiles 0, 1 of the wave; tiles 2, 3 (d_model 512) are parked in LDS
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (NH > 1) opark[((wave * 2 + t) * 4 + rt) * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  for (int ci = 0; ci < nchunk; ++ci) {
    const int c = (ci + rot) % nchunk, cn = (c + 1) % nchunk;
    f32x4 h[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float4 bb = *reinterpret_cast<const float4*>(a.b1 + c * 256 + (wave * 2 + t) * 16 + 4 * g);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    const WStream w1c{rs1, w1b + (unsigned)c * 16 * KS1 * 2 * 1024u, lane16};
    const WStream w2c{rs2, w2b + (unsigned)c * 2 * 2 * 1024u, lane16};
    // scale dwords of this chunk's W2 sub-passes: requested now, used after the W1 pass
#pragma unroll
    for (int hf = 0; hf < NH; ++hf)
#pragma unroll
      for (int t = 0; t < 2; ++t) wsc2[hf][t][0] = sc_load(rss2, w2sb, (unsigned)((hf * 2 + t) * kp2 + c));
    pass_mx<2, 2, KS1, XROWB, NTM, XB>(h, X8, xsc, n, g, w1c, KS1 * 2, w2c, ks2 * 2, true, wsc1, wq);
    // hidden = relu(h) as one MX block per (row, wave): the wave's 32 features ARE block `wave` of this 256-chunk
    unsigned char* hb = H8 + (ci & 1) * (HBYTES + SC_BYTES);
    unsigned char* hsc = hb + HBYTES;
#pragma unroll
    for (int rt = 0; rt < ((MX_ABL & 4) ? 0 : 4); ++rt) {
      float am = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) { h[t][rt][q] = fmaxf(h[t][rt][q], 0.f); am = fmaxf(am, h[t][rt][q]); }
      am = rows_max4(am);  // over the 4 lane groups: same token row, the wave's other 24 features
      const unsigned sb = mx_scale_byte(am);
      const float inv = mx_inv_scale(sb);
#pragma unroll
      for (int t = 0; t < 2; ++t)
        *reinterpret_cast<unsigned*>(hb + poff8<HROWB>(rt * 16 + n, (wave * 2 + t) * 16 + 4 * g)) = mx_pack4(h[t][rt][0], h[t][rt][1], h[t][rt][2], h[t][rt][3], inv);
      if (g == 0) hsc[((rt * 16 + n) * 4 + (wave & 3)) * 4 + (wave >> 2)] = (unsigned char)sb;  // block wave: k-step wave / 4, group wave % 4
    }
    // scale dwords of the next chunk's W1 pass
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int kp = 0; kp < KP1; ++kp) wsc1[t][kp] = sc_load(rss1, w1sb, (unsigned)((cn * 16 + t) * KP1 + kp));
    __syncthreads();
    const WStream w1n{rs1, w1b + (unsigned)cn * 16 * KS1 * 2 * 1024u, lane16};
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
      const WStream wcur{rs2, w2c.base + (unsigned)(hf * 2) * ks2 * 2 * 1024u, lane16};
      const bool last = hf + 1 == NH;
      const WStream wnext{last ? rs1 : rs2, last ? w1n.base : wcur.base + 2u * ks2 * 2 * 1024u, lane16};
      if (hf == 0) {
        pass_mx<2, 2, 2, HROWB, NTM, XB>(o, hb, reinterpret_cast<const unsigned*>(hsc), n, g, wcur, ks2 * 2, wnext,
                                         last ? KS1 * 2 : ks2 * 2, !last || ci + 1 < nchunk, wsc2[hf], wq);
      } else {
        f32x4 ob[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) ob[t][rt] = opark[((wave * 2 + t) * 4 + rt) * 64 + lane];
        pass_mx<2, 2, 2, HROWB, NTM, XB>(ob, hb, reinterpret_cast<const unsigned*>(hsc), n, g, wcur, ks2 * 2, wnext,
                                         last ? KS1 * 2 : ks2 * 2, !last || ci + 1 < nchunk, wsc2[hf], wq);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) opark[((wave * 2 + t) * 4 + rt) * 64 + lane] = ob[t][rt];
      }
    }
  }
  float4 b2v[NTO];
#pragma unroll
  for (int t = 0; t < NTO; ++t) b2v[t] = *reinterpret_cast<const float4*>(a.b2 + (wave * NTO + t) * 16 + 4 * g);
  f32x4 ob[2][4];   // the parked tiles come back into registers before their LDS region becomes part of the staging tile
  if constexpr (NH > 1) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) ob[t][rt] = opark[((wave * 2 + t) * 4 + rt) * 64 + lane];
  }
  __syncthreads();
  float* ys = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int hf = 0; hf < NH; ++hf)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const float4 bb = b2v[hf * 2 + t];
        const f32x4 ov = hf == 0 ? o[t][rt] : ob[t][rt];
        *reinterpret_cast<float4*>(ys + (rt * 16 + n) * (D + 4) + (wave * NTO + hf * 2 + t) * 16 + 4 * g) =
            make_float4(ov[0] + bb.x, ov[1] + bb.y, ov[2] + bb.z, ov[3] + bb.w);
      }
  __syncthreads();
  if (!(MX_ABL & 16)) finish_rows<D, FINAL_LN>(ys, a.x, a.out, a.fin_g, a.fin_b, m0, a.M, a.alpha, tid);
}

}  // namespace

extern "C" int csship_ffn_wide(const float* x, float* out, const float* ln_g, const float* ln_b, const uint16_t* w1_packed,
                               const float* b1, const uint16_t* w2_packed, const float* b2, const float* final_ln_g,
                               const float* final_ln_b, int32_t M, int32_t d_model, int32_t d_ffn, float alpha,
                               csship_stream_t stream) {
  if (!x || !out || !ln_g || !ln_b || !w1_packed || !b1 || !w2_packed || !b2) return CSSHIP_ENULL;
  if ((final_ln_g == nullptr) != (final_ln_b == nullptr)) return CSSHIP_ENULL;
  if (M <= 0 || d_ffn <= 0 || (d_ffn % 256) != 0) return CSSHIP_EBADSHAPE;
  if (d_model != 256 && d_model != 512) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(out) || !aligned16(w1_packed) || !aligned16(w2_packed) || !aligned16(b1) || !aligned16(b2))
    return CSSHIP_EALIGN;
  FfnwArgs a{x, out, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w1_packed), b1, reinterpret_cast<const bf16x8*>(w2_packed), b2,
             final_ln_g, final_ln_b, M, d_ffn, alpha};
  dim3 grid((M + BM - 1) / BM), block(512);
  hipStream_t s = as_stream(stream);
  if (d_model == 256) {
    if (final_ln_g) hipLaunchKernelGGL((ffnw_kernel<256, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ffnw_kernel<256, false>), grid, block, 0, s, a);
  } else {
    if (final_ln_g) hipLaunchKernelGGL((ffnw_kernel<512, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ffnw_kernel<512, false>), grid, block, 0, s, a);
  }
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_ln_linear_wide(const float* x, const float* ln_g, const float* ln_b, const uint16_t* w_il8_packed,
                                     const float* bias_il8, uint16_t* out_bf16, int32_t M, int32_t d_model, int32_t N,
                                     csship_stream_t stream) {
  if (!x || !ln_g || !ln_b || !w_il8_packed || !bias_il8 || !out_bf16) return CSSHIP_ENULL;
  if (M <= 0 || N <= 0 || (N % 256) != 0) return CSSHIP_EBADSHAPE;
  if (d_model != 256 && d_model != 512) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(w_il8_packed) || !aligned16(bias_il8) || !aligned16(out_bf16)) return CSSHIP_EALIGN;
  LinwArgs a{x, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w_il8_packed), bias_il8, nullptr, out_bf16, M, N};
  dim3 grid((M + BM - 1) / BM), block(512);
  const bool full = (M % BM) == 0 && N <= 2048 && aligned16(a.bias);
  if (d_model == 256 && full) hipLaunchKernelGGL((linw_kernel<256, true, true>), grid, block, 0, as_stream(stream), a);
  else if (d_model == 256) hipLaunchKernelGGL((linw_kernel<256, true, false>), grid, block, 0, as_stream(stream), a);
  else if (full) hipLaunchKernelGGL((linw_kernel<512, true, true>), grid, block, 0, as_stream(stream), a);
  else hipLaunchKernelGGL((linw_kernel<512, true, false>), grid, block, 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_linear_res_wide(const uint16_t* a_bf16, const uint16_t* w_packed, const float* bias, const float* res,
                                      float* out, int32_t M, int32_t d_model, csship_stream_t stream) {
  if (!a_bf16 || !w_packed || !bias || !res || !out) return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (d_model != 256 && d_model != 512) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(a_bf16) || !aligned16(w_packed) || !aligned16(bias) || !aligned16(res) || !aligned16(out)) return CSSHIP_EALIGN;
  LinwArgs a{a_bf16, nullptr, nullptr, reinterpret_cast<const bf16x8*>(w_packed), bias, res, out, M, d_model};
  dim3 grid((M + BM - 1) / BM), block(512);
  const bool full = (M % BM) == 0 && d_model <= 2048 && aligned16(a.bias);
  if (d_model == 256 && full) hipLaunchKernelGGL((linw_kernel<256, false, true>), grid, block, 0, as_stream(stream), a);
  else if (d_model == 256) hipLaunchKernelGGL((linw_kernel<256, false, false>), grid, block, 0, as_stream(stream), a);
  else if (full) hipLaunchKernelGGL((linw_kernel<512, false, true>), grid, block, 0, as_stream(stream), a);
  else hipLaunchKernelGGL((linw_kernel<512, false, false>), grid, block, 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_ffn_mx(const float* x, float* out, const float* ln_g, const float* ln_b, const uint8_t* w1_mx,
                             const uint32_t* w1_scales, const float* b1, const uint8_t* w2_mx, const uint32_t* w2_scales,
                             const float* b2, const float* final_ln_g, const float* final_ln_b, int32_t M, int32_t d_model,
                             int32_t d_ffn, float alpha, csship_stream_t stream) {
  if (!x || !out || !ln_g || !ln_b || !w1_mx || !w1_scales || !b1 || !w2_mx || !w2_scales || !b2) return CSSHIP_ENULL;
  if ((final_ln_g == nullptr) != (final_ln_b == nullptr)) return CSSHIP_ENULL;
  if (M <= 0 || d_ffn <= 0 || (d_ffn % 256) != 0) return CSSHIP_EBADSHAPE;
  if (d_model != 256 && d_model != 512) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(out) || !aligned16(w1_mx) || !aligned16(w2_mx) || !aligned16(b1) || !aligned16(b2))
    return CSSHIP_EALIGN;
  FfnMxArgs a{x, out, ln_g, ln_b, reinterpret_cast<const uint4*>(w1_mx), w1_scales, b1, reinterpret_cast<const uint4*>(w2_mx),
              w2_scales, b2, final_ln_g, final_ln_b, M, d_ffn, alpha};
  dim3 grid((M + BM - 1) / BM), block(512);
  hipStream_t s = as_stream(stream);
  if (d_model == 256) {
    if (final_ln_g) hipLaunchKernelGGL((ffnmx_kernel<256, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ffnmx_kernel<256, false>), grid, block, 0, s, a);
  } else {
    if (final_ln_g) hipLaunchKernelGGL((ffnmx_kernel<512, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ffnmx_kernel<512, false>), grid, block, 0, s, a);
  }
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}
