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

// acc[t][rt] += W(tile0+t)[16 x 256] . panel^T[256 x 16 rows of tile rt]
//   w: this lane's pointer at block (tile0, k-step 0); block (t, ks) is at w[(t*tstride + ks)*64].
//   wq: weight fragments of k-steps 0 and 1, prefetched by the caller; on return it holds k-steps
//   0 and 1 of the NEXT stream (wn / nstride), so the global-load pipeline never drains.
template <int NT>
DEV void mma_pass(f32x4 (&acc)[NT][4], const unsigned char* panel, int n, int g, int nt, const bf16x8* w, int tstride,
                  const bf16x8* wn, int nstride, int nnt, bf16x8 (&wq)[2][NT]) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 wc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wc[t] = wq[ks & 1][t];
    if (ks + 2 < KS) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) wq[ks & 1][t] = w[(size_t)(t * tstride + ks + 2) * 64];
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nnt) wq[ks & 1][t] = wn[(size_t)(t * nstride + ks + 2 - KS) * 64];
    }
    bf16x8 xa[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) xa[rt] = *reinterpret_cast<const bf16x8*>(panel + panel_off(rt * 16 + n, ks * 4 + g));
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[t], xa[rt], acc[t][rt], 0, 0, 0);
      }
  }
}

template <int NT>
DEV void prefetch2(bf16x8 (&wq)[2][NT], const bf16x8* w, int tstride, int nt) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nt) {
      wq[0][t] = w[(size_t)(t * tstride) * 64];
      wq[1][t] = w[(size_t)(t * tstride + 1) * 64];
    }
}

// ---- panel prologues (all NW waves cooperate; rows >= M are clamped, never stored later)
template <int NW>
DEV void load_panel_ln(unsigned char* panel, const float* x, int m0, int M, const float* gam, const float* bet, int wave,
                       int lane) {
  const float4 gm = *reinterpret_cast<const float4*>(gam + lane * 4);
  const float4 bt = *reinterpret_cast<const float4*>(bet + lane * 4);
  for (int r = wave; r < BM; r += NW) {
    const int row = min(m0 + r, M - 1);
    float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * DM + lane * 4);
    const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.f / DM);
    v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
    const float rstd = rsqrtf(wave_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.f / DM) + 1e-5f);
    const uint2 pk = make_uint2(pack_bf2(v.x * rstd * gm.x + bt.x, v.y * rstd * gm.y + bt.y),
                                pack_bf2(v.z * rstd * gm.z + bt.z, v.w * rstd * gm.w + bt.w));
    *reinterpret_cast<uint2*>(panel + panel_off(r, lane >> 1) + (lane & 1) * 8) = pk;
  }
}
template <int NW>
DEV void load_panel_f32(unsigned char* panel, const float* x, int m0, int M, int wave, int lane) {
  for (int r = wave; r < BM; r += NW) {
    const int row = min(m0 + r, M - 1);
    const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * DM + lane * 4);
    *reinterpret_cast<uint2*>(panel + panel_off(r, lane >> 1) + (lane & 1) * 8) =
        make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
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

// ---- per-row LayerNorm over the 256 features held as v[t][rt][r] by the NW waves of the group.
// Lane (g, n): rows rt*16 + n, features (wave*NT + t)*16 + 4g + r.
template <int NW, int NT>
DEV void row_layernorm(f32x4 (&v)[NT][4], float* red /* [NW][64] */, int wave, int g, int n, const float* gam,
                       const float* bet, int f0) {
  float mean[4], rstd[4];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = pass == 0 ? v[t][rt][r] : v[t][rt][r] - mean[rt];
          s += pass == 0 ? d : d * d;
        }
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (g == 0) red[wave * 64 + rt * 16 + n] = s;
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += red[w * 64 + rt * 16 + n];
      s *= (1.f / DM);
      if (pass == 0) mean[rt] = s; else rstd[rt] = rsqrtf(s + 1e-5f);
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 gm = *reinterpret_cast<const float4*>(gam + f0 + t * 16 + 4 * g);
    const float4 bt = *reinterpret_cast<const float4*>(bet + f0 + t * 16 + 4 * g);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      v[t][rt][0] = (v[t][rt][0] - mean[rt]) * rstd[rt] * gm.x + bt.x;
      v[t][rt][1] = (v[t][rt][1] - mean[rt]) * rstd[rt] * gm.y + bt.y;
      v[t][rt][2] = (v[t][rt][2] - mean[rt]) * rstd[rt] * gm.z + bt.z;
      v[t][rt][3] = (v[t][rt][3] - mean[rt]) * rstd[rt] * gm.w + bt.w;
    }
  }
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
};

template <int NW, bool FINAL_LN>
__global__ __launch_bounds__(NW * 64) void ffn_kernel(FfnArgs a) {
  constexpr int NT = 16 / NW;  // 16-feature tiles per wave per 256-feature pass
  __shared__ __attribute__((aligned(16))) unsigned char Xs[PANEL_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char Hs[2][PANEL_BYTES];
  __shared__ float red[NW * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM;
  const int nchunk = a.FF / DM, ks2 = a.FF / 32;

  // weight streams of this wave: W1 chunk c -> hidden tiles c*16 + wave*NT + t (tile stride KS blocks);
  // W2 chunk c -> output tiles wave*NT + t, k-steps 8c.. (tile stride FF/32 blocks)
  const bf16x8* w1s = a.w1 + (size_t)(wave * NT) * KS * 64 + lane;
  const bf16x8* w2s = a.w2 + (size_t)(wave * NT) * ks2 * 64 + lane;
  bf16x8 wq[2][NT];
  prefetch2<NT>(wq, w1s, KS, NT);  // in flight during the LN prologue

  load_panel_ln<NW>(Xs, a.x, m0, a.M, a.ln_g, a.ln_b, wave, lane);
  __syncthreads();

  f32x4 o[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) o[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int c = 0; c < nchunk; ++c) {
    f32x4 h[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) h[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8* w1c = w1s + (size_t)c * 16 * KS * 64;
    const bf16x8* w2c = w2s + (size_t)c * KS * 64;
    mma_pass<NT>(h, Xs, n, g, NT, w1c, KS, w2c, ks2, NT, wq);
    // hidden = relu(h + b1) -> bf16 -> Hs[c&1][row][wave's 16*NT columns]
    unsigned char* hs = Hs[c & 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int hc = (wave * NT + t) * 16 + 4 * g;  // column within the chunk
      const float4 bb = *reinterpret_cast<const float4*>(a.b1 + c * DM + hc);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const uint2 pk = make_uint2(pack_bf2(fmaxf(h[t][rt][0] + bb.x, 0.f), fmaxf(h[t][rt][1] + bb.y, 0.f)),
                                    pack_bf2(fmaxf(h[t][rt][2] + bb.z, 0.f), fmaxf(h[t][rt][3] + bb.w, 0.f)));
        *reinterpret_cast<uint2*>(hs + panel_off(rt * 16 + n, hc >> 3) + (hc & 7) * 2) = pk;
      }
    }
    __syncthreads();
    const bf16x8* w1n = (c + 1 < nchunk) ? w1c + (size_t)16 * KS * 64 : w1s;  // last: harmless re-read
    mma_pass<NT>(o, hs, n, g, NT, w2c, ks2, w1n, KS, NT, wq);
  }

  // out = x + alpha * (o + b2)  [-> LayerNorm]
  const int f0 = wave * NT * 16;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const float4 bb = *reinterpret_cast<const float4*>(a.b2 + f0 + t * 16 + 4 * g);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int row = min(m0 + rt * 16 + n, a.M - 1);
      const float4 xr = *reinterpret_cast<const float4*>(a.x + (size_t)row * DM + f0 + t * 16 + 4 * g);
      o[t][rt][0] = xr.x + a.alpha * (o[t][rt][0] + bb.x);
      o[t][rt][1] = xr.y + a.alpha * (o[t][rt][1] + bb.y);
      o[t][rt][2] = xr.z + a.alpha * (o[t][rt][2] + bb.z);
      o[t][rt][3] = xr.w + a.alpha * (o[t][rt][3] + bb.w);
    }
  }
  if constexpr (FINAL_LN) row_layernorm<NW, NT>(o, red, wave, g, n, a.fin_g, a.fin_b, f0);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const int row = m0 + rt * 16 + n;
      if (row < a.M)
        *reinterpret_cast<float4*>(a.out + (size_t)row * DM + f0 + t * 16 + 4 * g) =
            make_float4(o[t][rt][0], o[t][rt][1], o[t][rt][2], o[t][rt][3]);
    }
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
};

template <int NW, int PRO, int EPI>
__global__ __launch_bounds__(NW * 64) void panel_linear_kernel(LinArgs a) {
  constexpr int NT = 16 / NW;
  __shared__ __attribute__((aligned(16))) unsigned char Xs[PANEL_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, n = lane & 15;
  const int m0 = blockIdx.x * BM;
  const int ntiles = (a.N + 15) / 16;
  const int tpw = (ntiles + NW - 1) / NW;  // contiguous tile range per wave
  const int tbeg = wave * tpw, tend = min(tbeg + tpw, ntiles);

  bf16x8 wq[2][NT];
  const bf16x8* wbase = a.w + lane;
  {
    const int nt0 = max(0, min(NT, tend - tbeg));
    prefetch2<NT>(wq, wbase + (size_t)tbeg * KS * 64, KS, nt0);
  }
  if constexpr (PRO == CSSHIP_PRO_LN) load_panel_ln<NW>(Xs, reinterpret_cast<const float*>(a.a), m0, a.M, a.ln_g, a.ln_b, wave, lane);
  else if constexpr (PRO == CSSHIP_PRO_F32) load_panel_f32<NW>(Xs, reinterpret_cast<const float*>(a.a), m0, a.M, wave, lane);
  else load_panel_bf16<NW>(Xs, reinterpret_cast<const unsigned short*>(a.a), m0, a.M, tid);
  __syncthreads();

  for (int t0 = tbeg; t0 < tend; t0 += NT) {
    const int nt = min(NT, tend - t0);
    const int nnt = max(0, min(NT, tend - (t0 + NT)));
    f32x4 acc[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8* wc = wbase + (size_t)t0 * KS * 64;
    mma_pass<NT>(acc, Xs, n, g, nt, wc, KS, wc + (size_t)NT * KS * 64, KS, nnt, wq);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t >= nt) continue;
      const int f = (t0 + t) * 16 + 4 * g;  // 4 consecutive features f..f+3
      float bb[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bb[r] = (a.bias && f + r < a.N) ? a.bias[f + r] : 0.f;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int row = m0 + rt * 16 + n;
        if (row >= a.M) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[t][rt][r] + bb[r];
        if constexpr ((EPI & CSSHIP_EPI_RESIDUAL) != 0) {  // N == ldr-aligned 256-wide residual stream
          const float4 xr = *reinterpret_cast<const float4*>(a.res + (size_t)row * a.ldr + f);
          v[0] = xr.x + a.alpha * v[0]; v[1] = xr.y + a.alpha * v[1];
          v[2] = xr.z + a.alpha * v[2]; v[3] = xr.w + a.alpha * v[3];
        }
        if constexpr ((EPI & CSSHIP_EPI_RELU) != 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        if constexpr ((EPI & CSSHIP_EPI_OUT_BF16) != 0) {
          unsigned short* o = reinterpret_cast<unsigned short*>(a.out) + (size_t)row * a.ldo + f;
          if (f + 3 < a.N && (a.ldo & 3) == 0) {
            *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (f + r < a.N) o[r] = f2bf(v[r]);
          }
        } else {
          float* o = reinterpret_cast<float*>(a.out) + (size_t)row * a.ldo + f;
          if (f + 3 < a.N && (a.ldo & 3) == 0) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (f + r < a.N) o[r] = v[r];
          }
        }
      }
    }
  }
}

constexpr int NWAVES = 8;

template <int PRO, int EPI>
int launch_linear(const LinArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((panel_linear_kernel<NWAVES, PRO, EPI>), dim3((a.M + BM - 1) / BM), dim3(NWAVES * 64), 0, s, a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

}  // namespace

extern "C" int csship_ffn_fused(const float* x, float* out, const float* ln_g, const float* ln_b,
                                const uint16_t* w1_packed, const float* b1, const uint16_t* w2_packed,
                                const float* b2, const float* final_ln_g, const float* final_ln_b, int32_t M,
                                int32_t d_model, int32_t d_ffn, float alpha, csship_stream_t stream) {
  if (!x || !out || !ln_g || !ln_b || !w1_packed || !b1 || !w2_packed || !b2) return CSSHIP_ENULL;
  if ((final_ln_g == nullptr) != (final_ln_b == nullptr)) return CSSHIP_ENULL;
  if (M <= 0) return CSSHIP_EBADSHAPE;
  if (d_model != DM || d_ffn <= 0 || (d_ffn % DM) != 0) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(out) || !aligned16(w1_packed) || !aligned16(w2_packed) || !aligned16(ln_g) ||
      !aligned16(ln_b) || !aligned16(b1) || !aligned16(b2) || (final_ln_g && (!aligned16(final_ln_g) || !aligned16(final_ln_b))))
    return CSSHIP_EALIGN;
  FfnArgs a{x, out, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w1_packed), b1,
            reinterpret_cast<const bf16x8*>(w2_packed), b2, final_ln_g, final_ln_b, M, d_ffn, alpha};
  dim3 grid((M + BM - 1) / BM), block(NWAVES * 64);
  if (final_ln_g) hipLaunchKernelGGL((ffn_kernel<NWAVES, true>), grid, block, 0, as_stream(stream), a);
  else hipLaunchKernelGGL((ffn_kernel<NWAVES, false>), grid, block, 0, as_stream(stream), a);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_panel_linear(const void* a, int32_t prologue, const float* ln_g, const float* ln_b,
                                   const uint16_t* w_packed, const float* bias, const float* residual, int64_t ldr,
                                   float alpha, void* out, int64_t ldo, int32_t M, int32_t N, int32_t d_model,
                                   int32_t epilogue, csship_stream_t stream) {
  if (!a || !w_packed || !out) return CSSHIP_ENULL;
  if (M <= 0 || N <= 0 || ldo < N) return CSSHIP_EBADSHAPE;
  if (d_model != DM) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(a) || !aligned16(w_packed) || !aligned16(out)) return CSSHIP_EALIGN;
  if (prologue == CSSHIP_PRO_LN && (!ln_g || !ln_b || !aligned16(ln_g) || !aligned16(ln_b))) return CSSHIP_ENULL;
  if ((epilogue & CSSHIP_EPI_RESIDUAL) && (!residual || !aligned16(residual) || (ldr & 3) != 0 || ldr < N || (N & 3) != 0))
    return CSSHIP_EALIGN;
  LinArgs la{a, ln_g, ln_b, reinterpret_cast<const bf16x8*>(w_packed), bias, residual, out, M, N, ldo, ldr, alpha};
  hipStream_t s = as_stream(stream);
  const int key = prologue * 16 + epilogue;
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
