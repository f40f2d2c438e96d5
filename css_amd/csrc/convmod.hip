// Conformer convolution module + residual, fused (css/models/conformer.py:289-303, 357):
//   y = x + pw2( relu( BN( dwconv_k( glu( LN(x) ) ) ) ) )
// The two "pointwise convs" of the reference are Conv2d(1,2,1) / Conv2d(1,1,1), i.e. scalar
// gains (parameter shapes (2,1,1,1) and (1,1,1,1)), so the module has no GEMM: it is a
// bandwidth-bound stencil.  One workgroup = (batch b, TT frames, all D channels):
//   phase 1  each wave normalises whole rows (LN over D, wave reduction) for the TT + k - 1
//            frames of the tile + halo, applies the scalar GLU and parks them in LDS;
//   phase 2  thread c owns channel c: k-tap depthwise convolution from LDS columns
//            (conflict free), BatchNorm, ReLU, scalar affine, residual add, coalesced store.
// bn_mode 1 (training forward, batch statistics) splits after the convolution: pass A stores
// the pre-BN activations and per-workgroup partial sums, a tiny reduce kernel forms the
// per-channel scale/shift deterministically (no atomics), pass B finishes.
#include <stdlib.h>

#include "common.hpp"

namespace {

struct ConvArgs {
  const float* x;
  float* y;
  csship_conv_weights w;
  const float* g;  // FROMG: glu(LN(x)) rows [B,T,D]
  float* zbuf;     // train: pre-BN activations [B,T,D]
  float* partial;  // train: [nblk][2][D]; MODE 2: [nblk][KS+1][D]
  const float* dc; // MODE 2: gradient wrt the depthwise-conv output [B,T,D]
  int B, T;
  unsigned long long* stamps;
};

template <int D, int TT, int KS, int MODE, bool FROMG>
__global__ __launch_bounds__(2 * D) void convmod_kernel(ConvArgs a) {
  constexpr int ROWS = TT + KS - 1, NTHR = 2 * D;
  // frames in front of the tile that the taps reach: (KS-1)/2 for the centred module; the causal module
  // (conformer.py:265-273, 297-298) looks KS-1 frames back and none ahead - and its input gradient (MODE 3: correlation
  // with the flipped taps) the other way round.  Everything below indexes the tile relatively: only this mapping differs.
  const int HALO = a.w.causal ? (MODE == 3 ? 0 : KS - 1) : (KS - 1) / 2;
  __shared__ __attribute__((aligned(16))) float G[ROWS][D];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  stamp(a.stamps, 0);
  const float w0 = a.w.pw1_w[0], w1 = a.w.pw1_w[1], b0 = a.w.pw1_b[0], b1 = a.w.pw1_b[1];

  // phase 1: LPR = D/32 lanes per row (each lane 32 channels as 8 float4 pieces, the lanes of a row cover
  // whole 128-byte lines per load); LayerNorm statistics = in-lane sums + log2(LPR) shuffles.
  constexpr int LPR = D / 32, RPB = NTHR / LPR;  // rows per batch of the workgroup (64 for D = 256)
  const int part = tid % LPR;
  if constexpr (FROMG) {  // g = glu(LayerNorm(x)) was produced by csship_outproj_glu: plain tile + halo load
    for (int i = tid; i < ROWS * (D / 4); i += NTHR) {
      const int row = i / (D / 4), c4 = i % (D / 4);
      const int t = t0 - HALO + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (t >= 0 && t < a.T) v = *reinterpret_cast<const float4*>(a.g + ((size_t)b * a.T + t) * D + c4 * 4);
      *reinterpret_cast<float4*>(&G[row][c4 * 4]) = v;
    }
  } else
  for (int r0 = 0; r0 < ROWS; r0 += RPB) {
    const int row = r0 + tid / LPR;
    if (row >= ROWS) break;
    const int t = t0 - HALO + row;
    const bool inside = t >= 0 && t < a.T;
    const float* xr = a.x + ((size_t)b * a.T + (inside ? t : 0)) * D;
    float4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (i * LPR + part) * 4);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    static_assert(LPR == 8 || LPR == 16, "row reductions are written for 8 or 16 lanes per row");
    s = sum8(s);
    if constexpr (LPR == 16) s += dpp_mov<DPP_MIRROR>(s);
    const float mean = s * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    ss = sum8(ss);
    if constexpr (LPR == 16) ss += dpp_mov<DPP_MIRROR>(ss);
    const float rstd = rsqrtf(ss * (1.f / D) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = (i * LPR + part) * 4;
      const float4 gm = *reinterpret_cast<const float4*>(a.w.ln_g + c);
      const float4 bt = *reinterpret_cast<const float4*>(a.w.ln_b + c);
      float e[4] = {v[i].x * rstd * gm.x + bt.x, v[i].y * rstd * gm.y + bt.y, v[i].z * rstd * gm.z + bt.z,
                    v[i].w * rstd * gm.w + bt.w};
      // scalar GLU (w0 e + b0) * sigmoid(w1 e + b1); v_rcp_f32 (1 ulp) instead of an IEEE divide
#pragma unroll
      for (int q = 0; q < 4; ++q)
        e[q] = inside ? (w0 * e[q] + b0) * __builtin_amdgcn_rcpf(1.f + __expf(-(w1 * e[q] + b1))) : 0.f;
      *reinterpret_cast<float4*>(&G[row][c]) = make_float4(e[0], e[1], e[2], e[3]);
    }
  }

  // phase 2: two threads per channel, each TT/2 consecutive frames
  const int c = tid % D, half = tid / D;
  constexpr int TH = TT / 2, OB = 8;
  const int tb0 = half * TH;
  float xres[TH];  // residual rows, fetched while phase 1 of the other waves finishes
  if constexpr (MODE == 0) {
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int t = min(t0 + tb0 + j, a.T - 1);
      xres[j] = a.x[((size_t)b * a.T + t) * D + c];
    }
  }
  if constexpr (MODE == 2) {
    // weight-gradient pass: d dw_w[k][c] = sum_t dc[t][c] * g[t + k - HALO][c] over this half tile, from registers
    __syncthreads();
    float gv[TH + KS - 1], dcv[TH];
#pragma unroll
    for (int u = 0; u < TH + KS - 1; ++u) gv[u] = G[tb0 + u][c];
    float sdc = 0.f;
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int t = t0 + tb0 + j;
      dcv[j] = t < a.T ? a.dc[((size_t)b * a.T + t) * D + c] : 0.f;
      sdc += dcv[j];
    }
    // One partial per workgroup: the second half tile hands its 34 sums to the first through the (now dead) staging
    // tile - 416 slabs of 34 KB instead of 832 for the slab sum behind this pass (29 -> 14.5 MB)
    float sk[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < TH; ++j) sacc += dcv[j] * gv[j + k];
      sk[k] = sacc;
    }
    __syncthreads();            // every thread has taken its rows of G into registers
    float* const X = &G[0][0];  // [KS + 1][D]
    if (half == 1) {
#pragma unroll
      for (int k = 0; k < KS; ++k) X[k * D + c] = sk[k];
      X[KS * D + c] = sdc;
    }
    __syncthreads();
    if (half == 0) {
      const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
      float* out = a.partial + blk * (KS + 1) * D + c;
#pragma unroll
      for (int k = 0; k < KS; ++k) out[k * D] = sk[k] + X[k * D + c];
      out[KS * D] = sdc + X[KS * D + c];
    }
    return;
  }
  float wk[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) wk[k] = a.w.dw_w[(MODE == 3 ? KS - 1 - k : k) * D + c];  // [k][D]; MODE 3: flipped taps
  const float cb = MODE == 3 ? 0.f : a.w.dw_b[c];
  float scale = 0.f, shift = 0.f, pw2w = 0.f, pw2b = 0.f;
  if constexpr (MODE == 0) {
    scale = a.w.bn_g[c] * rsqrtf(a.w.bn_var[c] + 1e-5f);
    shift = a.w.bn_b[c] - a.w.bn_mean[c] * scale;
    pw2w = a.w.pw2_w[0];
    pw2b = a.w.pw2_b[0];
  }
  __syncthreads();
  stamp(a.stamps, 1);
  float lsum = 0.f, lsq = 0.f;
#pragma unroll
  for (int blk = 0; blk < TH / OB; ++blk) {
    const int tb = tb0 + blk * OB;
    float acc[OB];
#pragma unroll
    for (int j = 0; j < OB; ++j) acc[j] = cb;
#pragma unroll
    for (int u = 0; u < OB + KS - 1; ++u) {
      const float gv = G[tb + u][c];
#pragma unroll
      for (int j = 0; j < OB; ++j) {
        const int k = u - j;
        if (k >= 0 && k < KS) acc[j] += wk[k] * gv;
      }
    }
#pragma unroll
    for (int j = 0; j < OB; ++j) {
      const int t = t0 + tb + j;
      if (t >= a.T) continue;
      const size_t o = ((size_t)b * a.T + t) * D + c;
      if constexpr (MODE == 0) {
        a.y[o] = xres[blk * OB + j] + fmaxf(acc[j] * scale + shift, 0.f) * pw2w + pw2b;
      } else if constexpr (MODE == 3) {
        a.y[o] = acc[j];  // dg = correlation of dc with the flipped taps
      } else {
        a.zbuf[o] = acc[j];
        lsum += acc[j];
        lsq += acc[j] * acc[j];
      }
    }
  }
  stamp(a.stamps, 3);
  if constexpr (MODE == 1) {
    const size_t blk = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + half;
    a.partial[(blk * 2 + 0) * D + c] = lsum;
    a.partial[(blk * 2 + 1) * D + c] = lsq;
  }
}

// per-channel batch statistics -> scale/shift (biased variance, as BatchNorm1d normalises with).
// Workgroup = 64 channels x 16 waves; wave w sums partials w, w+16, ... in double, then one LDS reduction.
__global__ __launch_bounds__(1024) void bn_reduce_kernel(const float* __restrict__ partial, int nblk, int D, double count,
                                                         const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                         float* __restrict__ scale_shift) {
  __shared__ double red[2][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double s = 0.0, q = 0.0;
  if (c < D) {
    // four partials per lane in flight (a dependent chain of nblk/16 loads took 19 us); out-of-range ones re-read the
    // last partial and are masked, so no load sits behind a branch
    double sa[4] = {0.0, 0.0, 0.0, 0.0}, qa[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = w; i < nblk; i += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ii = i + 16 * u;
        const size_t o = (size_t)min(ii, nblk - 1) * 2 * D + c;
        const float a = partial[o], b = partial[o + D];
        sa[u] += ii < nblk ? (double)a : 0.0;
        qa[u] += ii < nblk ? (double)b : 0.0;
      }
    }
    s = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    q = (qa[0] + qa[1]) + (qa[2] + qa[3]);
  }
  red[0][w][lane] = s;
  red[1][w][lane] = q;
  __syncthreads();
  if (w == 0 && c < D) {
    s = q = 0.0;
    for (int i = 0; i < 16; ++i) {
      s += red[0][i][lane];
      q += red[1][i][lane];
    }
    const double mu = s / count;
    const double var = fmax(q / count - mu * mu, 0.0);
    const float sc = bn_g[c] * (float)(1.0 / sqrt(var + 1e-5));
    scale_shift[c] = sc;
    scale_shift[D + c] = bn_b[c] - (float)mu * sc;
    scale_shift[2 * D + c] = (float)mu;   // batch mean / biased variance: running statistics and the backward
    scale_shift[3 * D + c] = (float)var;
  }
}

// The same statistics with the work spread over D / 8 workgroups (the form above runs on D / 64 = 4 CUs and is a chain of
// dependent partial loads: 10.5 us per launch, 16 launches per training step).  Workgroup = 8 channels x 2 statistics;
// thread t takes partials t, t + 256, ... (two 32-byte loads each, all independent), parks its 16 sums in LDS as
// doubles, and the sums of the 256 threads are added in a fixed two-level order.
__global__ __launch_bounds__(256) void bn_reduce8_kernel(const float* __restrict__ partial, int nblk, int D, double count,
                                                         const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                         float* __restrict__ scale_shift) {
  __shared__ double red[256][17];   // [thread][8 sums | 8 sums of squares] (+1: the column reads below stride 17)
  __shared__ double lvl[16][16];
  const int t = threadIdx.x, c0 = blockIdx.x * 8;
  double dacc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) dacc[u] = 0.0;
  for (int i = t; i < nblk; i += 256) {
    const float* p = partial + (size_t)i * 2 * D + c0;
    const float4 a0 = *reinterpret_cast<const float4*>(p), a1 = *reinterpret_cast<const float4*>(p + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p + D), b1 = *reinterpret_cast<const float4*>(p + D + 4);
    const float v[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int u = 0; u < 16; ++u) dacc[u] += (double)v[u];
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) red[t][u] = dacc[u];
  __syncthreads();
  {  // level 1: thread (seg = t >> 4, u = t & 15) adds the 16 threads of its segment, in order
    const int seg = t >> 4, u = t & 15;
    double s = 0.0;
    for (int k = 0; k < 16; ++k) s += red[seg * 16 + k][u];
    lvl[seg][u] = s;
  }
  __syncthreads();
  if (t < 8) {  // level 2: the 16 segments, in order; thread t finishes channel c0 + t
    double s = 0.0, q = 0.0;
    for (int k = 0; k < 16; ++k) { s += lvl[k][t]; q += lvl[k][8 + t]; }
    const int c = c0 + t;
    const double mu = s / count;
    const double var = fmax(q / count - mu * mu, 0.0);
    const float sc = bn_g[c] * (float)(1.0 / sqrt(var + 1e-5));
    scale_shift[c] = sc;
    scale_shift[D + c] = bn_b[c] - (float)mu * sc;
    scale_shift[2 * D + c] = (float)mu;
    scale_shift[3 * D + c] = (float)var;
  }
}

// SyncBN split of bn_reduce_kernel: (1) partials -> per-channel sums (f64: sum z | sum z^2 over this rank's rows), which the
// caller all-reduces across ranks; (2) global sums + global row count -> scale / shift / mean / var
__global__ __launch_bounds__(1024) void bn_sums_kernel(const float* __restrict__ partial, int nblk, int D,
                                                       double* __restrict__ sums) {
  __shared__ double red[2][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  double s = 0.0, q = 0.0;
  if (c < D)
    for (int i = w; i < nblk; i += 16) {
      s += (double)partial[(size_t)i * 2 * D + c];
      q += (double)partial[(size_t)i * 2 * D + D + c];
    }
  red[0][w][lane] = s;
  red[1][w][lane] = q;
  __syncthreads();
  if (w == 0 && c < D) {
    s = q = 0.0;
    for (int i = 0; i < 16; ++i) { s += red[0][i][lane]; q += red[1][i][lane]; }
    sums[c] = s;
    sums[D + c] = q;
  }
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ sums, double count, int D,
                                                          const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                          float* __restrict__ scale_shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  const double mu = sums[c] / count;
  const double var = fmax(sums[D + c] / count - mu * mu, 0.0);
  const float sc = bn_g[c] * (float)(1.0 / sqrt(var + 1e-5));
  scale_shift[c] = sc;
  scale_shift[D + c] = bn_b[c] - (float)mu * sc;
  scale_shift[2 * D + c] = (float)mu;
  scale_shift[3 * D + c] = (float)var;
}

// bn_mode 2: scale / shift / mean / var from the RUNNING statistics, in the layout bn_reduce_kernel produces, so the
// backward (and bn_apply) read one format
__global__ __launch_bounds__(256) void bn_from_running_kernel(const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                              const float* __restrict__ mean, const float* __restrict__ var,
                                                              int D, float* __restrict__ scale_shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D) return;
  const float sc = bn_g[c] * rsqrtf(var[c] + 1e-5f);
  scale_shift[c] = sc;
  scale_shift[D + c] = bn_b[c] - mean[c] * sc;
  scale_shift[2 * D + c] = mean[c];
  scale_shift[3 * D + c] = var[c];
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                       const float* __restrict__ scale_shift,
                                                       const float* __restrict__ pw2_w, const float* __restrict__ pw2_b,
                                                       float* __restrict__ y, size_t n4, int D, csship_dropout_cfg dcfg) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const Drop dp = drop_make(dcfg.seed, dcfg.stream, dcfg.p);  // training: the module's final dropout (conformer.py:302)
  const int c = (int)((i * 4) % D);
  const float4 xv = reinterpret_cast<const float4*>(x)[i];
  const float4 zv = reinterpret_cast<const float4*>(z)[i];
  const float4 sc = *reinterpret_cast<const float4*>(scale_shift + c);
  const float4 sh = *reinterpret_cast<const float4*>(scale_shift + D + c);
  const float pw = pw2_w[0], pb = pw2_b[0];
  const float4 br = drop_apply4(dp, (unsigned)(i * 4),
                                make_float4(fmaxf(zv.x * sc.x + sh.x, 0.f) * pw + pb, fmaxf(zv.y * sc.y + sh.y, 0.f) * pw + pb,
                                            fmaxf(zv.z * sc.z + sh.z, 0.f) * pw + pb, fmaxf(zv.w * sc.w + sh.w, 0.f) * pw + pb));
  reinterpret_cast<float4*>(y)[i] = make_float4(xv.x + br.x, xv.y + br.y, xv.z + br.z, xv.w + br.w);
}

// phase 0: the whole module; SyncBN (bn_mode 1 only): phase 1 = convolution + this rank's channel sums (f64 [2D] behind the
// scale/shift block of the workspace), phase 2 = statistics from the (all-reduced) sums and `total_rows`, then apply
template <int D, int TT>
int run(const ConvArgs& a0, int bn_mode, void* workspace, hipStream_t s,
        csship_dropout_cfg dcfg = csship_dropout_cfg{nullptr, 0u, 0.f}, int phase = 0, double total_rows = 0.0) {
  ConvArgs a = a0;
  dim3 grid((a.T + TT - 1) / TT, a.B);
  if (bn_mode == 0) {
    if (a.g != nullptr) hipLaunchKernelGGL((convmod_kernel<D, TT, 33, 0, true>), grid, dim3(2 * D), 0, s, a);
    else hipLaunchKernelGGL((convmod_kernel<D, TT, 33, 0, false>), grid, dim3(2 * D), 0, s, a);
    CSSHIP_CHECK_LAUNCH();
    return CSSHIP_OK;
  }
  if (!workspace) return CSSHIP_EWORKSPACE;
  const size_t n = (size_t)a.B * a.T * D;
  const int nblk = grid.x * grid.y * 2;  // two half-tiles per workgroup
  a.zbuf = reinterpret_cast<float*>(workspace);
  a.partial = a.zbuf + n;
  double* sums = reinterpret_cast<double*>(a.partial + (size_t)nblk * 2 * D);  // f64 [2D] (SyncBN); statistics stay LAST
  float* scale_shift = a.partial + (size_t)nblk * 2 * D + 4 * D;
  if (phase != 2) {
    hipLaunchKernelGGL((convmod_kernel<D, TT, 33, 1, false>), grid, dim3(2 * D), 0, s, a);
    CSSHIP_CHECK_LAUNCH();
  }
  if (phase == 1) {
    hipLaunchKernelGGL(bn_sums_kernel, dim3((D + 63) / 64), dim3(1024), 0, s, a.partial, nblk, D, sums);
    CSSHIP_CHECK_LAUNCH();
    return CSSHIP_OK;
  }
  if (phase == 2)
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((D + 255) / 256), dim3(256), 0, s, sums, total_rows, D, a.w.bn_g, a.w.bn_b,
                       scale_shift);
  else if (bn_mode == 2)
    hipLaunchKernelGGL(bn_from_running_kernel, dim3((D + 255) / 256), dim3(256), 0, s, a.w.bn_g, a.w.bn_b, a.w.bn_mean,
                       a.w.bn_var, D, scale_shift);
  else if ((D & 7) == 0 && !getenv("CSSHIP_BN_REDUCE_OLD"))
    hipLaunchKernelGGL(bn_reduce8_kernel, dim3(D / 8), dim3(256), 0, s, a.partial, nblk, D,
                       (double)a.B * (double)a.T, a.w.bn_g, a.w.bn_b, scale_shift);
  else
    hipLaunchKernelGGL(bn_reduce_kernel, dim3((D + 63) / 64), dim3(1024), 0, s, a.partial, nblk, D,
                       (double)a.B * (double)a.T, a.w.bn_g, a.w.bn_b, scale_shift);
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, a.x, a.zbuf, scale_shift,
                     a.w.pw2_w, a.w.pw2_b, a.y, n / 4, D, dcfg);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

constexpr int TT256 = 32, TT512 = 32;

// ---------------------------------------------------------------------------------------------
// Backward (training forward = bn_mode 1; z = pre-BN activations and stats = [sc | sh | mean | var] are the
// forward's workspace).  With zh = z*sc + sh, r = relu(zh), y = x + r*w3 + b3:
//   pass A   per channel S1 = sum dzh, S2 = sum dzh * zn  (dzh = dy*w3*[zh > 0], zn = (z - mean)*rstd); scalars
//            d w3 = sum dy*r, d b3 = sum dy                        -> partials, fixed-order reduce
//   pass A'  dc = sc * (dzh - S1/n - zn * S2/n)                    (BatchNorm backward with batch statistics)
//   pass B1  dg = dc (*) flipped taps                              (convmod_kernel MODE 3)
//   pass B2  d dw_w, d dw_b partials from g (recomputed) and dc    (convmod_kernel MODE 2)
//   pass C   row kernel: GLU backward + LayerNorm backward + residual -> dx, d ln, d pw1
constexpr int BWD_ROWS_C = 16;   // = BWD_ROWS (the launch's rows per workgroup): the unrolled form of the statistics pass
template <int D>
__global__ __launch_bounds__(D) void conv_bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                              const float* __restrict__ stats,
                                                              const float* __restrict__ pw2_w, int64_t M,
                                                              int rows_per_wg, float* __restrict__ part,
                                                              csship_dropout_cfg dcfg) {
  __shared__ float red[2][D / 64];
  const int c = threadIdx.x;
  const Drop dp = drop_make(dcfg.seed, dcfg.stream, dcfg.p);  // the forward's output dropout: d(branch) = dy o mask / (1 - p)
  const float sc = stats[c], sh = stats[D + c], mu = stats[2 * D + c], rstd = rsqrtf(stats[3 * D + c] + 1e-5f);
  const float w3 = pw2_w[0];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
  float s1 = 0.f, s2 = 0.f, dw3 = 0.f, db3 = 0.f;
  if (rows_per_wg == BWD_ROWS_C) {
    // all of the workgroup's rows requested at once (clamped addresses, rows beyond M masked): with the loads inside a loop
    // of run-time length every row was a memory round trip of its own (13.2 us per launch for 26 MB = 2 TB/s); same
    // summation order
    float dv[BWD_ROWS_C], zv[BWD_ROWS_C];
#pragma unroll
    for (int i = 0; i < BWD_ROWS_C; ++i) {
      const int64_t r = min(r0 + i, M - 1);
      dv[i] = dy[r * D + c];
      zv[i] = z[r * D + c];
    }
#pragma unroll
    for (int i = 0; i < BWD_ROWS_C; ++i) {
      const int64_t r = r0 + i;
      float d = r < M ? dv[i] : 0.f;
      const float zz = zv[i];
      if (dp.thresh != 0u) d = drop_keep1(dp, (unsigned)(min(r, M - 1) * D + c)) ? d * dp.scale : 0.f;
      const float zh = zz * sc + sh;
      const float dzh = zh > 0.f ? d * w3 : 0.f;
      s1 += dzh;
      s2 += dzh * (zz - mu) * rstd;
      dw3 += d * fmaxf(zh, 0.f);
      db3 += d;
    }
  } else
  for (int64_t r = r0; r < r1; ++r) {
    float d = dy[r * D + c];
    const float zz = z[r * D + c];
    if (dp.thresh != 0u) d = drop_keep1(dp, (unsigned)(r * D + c)) ? d * dp.scale : 0.f;
    const float zh = zz * sc + sh;
    const float dzh = zh > 0.f ? d * w3 : 0.f;
    s1 += dzh;
    s2 += dzh * (zz - mu) * rstd;
    dw3 += d * fmaxf(zh, 0.f);
    db3 += d;
  }
  float* out = part + (int64_t)blockIdx.x * (2 * D + 2);
  out[c] = s1;
  out[D + c] = s2;
  dw3 = wave_sum(dw3);
  db3 = wave_sum(db3);
  if ((c & 63) == 0) { red[0][c >> 6] = dw3; red[1][c >> 6] = db3; }
  __syncthreads();
  if (c == 0) {
    float a0 = 0.f, a1 = 0.f;
    for (int i = 0; i < D / 64; ++i) { a0 += red[0][i]; a1 += red[1][i]; }
    out[2 * D] = a0;
    out[2 * D + 1] = a1;
  }
}

__global__ __launch_bounds__(256) void conv_bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                const float* __restrict__ stats,
                                                                const float* __restrict__ sums,
                                                                const float* __restrict__ pw2_w, float inv_n, size_t n4,
                                                                int D, float* __restrict__ dc, csship_dropout_cfg dcfg) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % D);
  const Drop dp = drop_make(dcfg.seed, dcfg.stream, dcfg.p);
  const float4 d = drop_apply4(dp, (unsigned)(i * 4), reinterpret_cast<const float4*>(dy)[i]);
  const float4 zz = reinterpret_cast<const float4*>(z)[i];
  const float w3 = pw2_w[0];
  const float dv[4] = {d.x, d.y, d.z, d.w}, zv[4] = {zz.x, zz.y, zz.z, zz.w};
  // per-channel constants as six 16-byte loads (c is a multiple of 4; they were 24 dword loads per thread beside the two
  // data loads: the kernel was bound by the rate at which a CU takes wave memory instructions, 3 TB/s)
  const float4 sc4 = *reinterpret_cast<const float4*>(stats + c), sh4 = *reinterpret_cast<const float4*>(stats + D + c);
  const float4 mu4 = *reinterpret_cast<const float4*>(stats + 2 * D + c), va4 = *reinterpret_cast<const float4*>(stats + 3 * D + c);
  const float4 s14 = *reinterpret_cast<const float4*>(sums + c), s24 = *reinterpret_cast<const float4*>(sums + D + c);
  const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w}, muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
  const float vav[4] = {va4.x, va4.y, va4.z, va4.w}, s1v[4] = {s14.x, s14.y, s14.z, s14.w}, s2v[4] = {s24.x, s24.y, s24.z, s24.w};
  float o[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float sc = scv[q], sh = shv[q], mu = muv[q];
    const float rstd = rsqrtf(vav[q] + 1e-5f);
    const float zh = zv[q] * sc + sh;
    const float dzh = zh > 0.f ? dv[q] * w3 : 0.f;
    o[q] = sc * (dzh - s1v[q] * inv_n - (zv[q] - mu) * rstd * s2v[q] * inv_n);
  }
  reinterpret_cast<float4*>(dc)[i] = make_float4(o[0], o[1], o[2], o[3]);
}

// optional second form of the module's input gradient for the operator behind it in the backward pass (the attention
// output projection: its weight and input gradients multiply with the DROPPED gradient, as bf16 MFMA operands):
// out = bf16(mask / (1 - p) o dx), element index as csship_dropout_apply on the dense [B*T, D] rows; p = 0: plain bf16(dx)
struct NextOut {
  const unsigned* seed;
  unsigned stream;
  float p;
  unsigned short* out;
};

// pass C: one wave per row.  a = LN(x); g = (w0 a + b0) * sigmoid(w1 a + b1); given dg: da, then LayerNorm backward,
// dx = dy + ...; partials [nwg][2D + 4]: d ln_g | d ln_b | d w0, d w1, d b0, d b1
template <int VPL>
__global__ __launch_bounds__(256) void glu_ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dg,
                                                         const float* __restrict__ dy, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ pw1_w,
                                                         const float* __restrict__ pw1_b, int64_t M, int rows_per_wg,
                                                         float* __restrict__ dx, float* __restrict__ part,
                                                         NextOut nx = NextOut{}) {
  constexpr int D = 256 * VPL;
  __shared__ float red[4][2 * D + 4];
  const Drop ndrop = drop_make(nx.seed, nx.stream, nx.p);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float w0 = pw1_w[0], w1 = pw1_w[1], b0 = pw1_b[0], b1 = pw1_b[1];
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
  float gm[VPL][4], bt[VPL][4], ag[VPL][4], ab[VPL][4];
  float sw0 = 0.f, sw1 = 0.f, sb0 = 0.f, sb1 = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float4 g4 = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
    const float4 b4 = *reinterpret_cast<const float4*>(beta + (i * 64 + lane) * 4);
    gm[i][0] = g4.x; gm[i][1] = g4.y; gm[i][2] = g4.z; gm[i][3] = g4.w;
    bt[i][0] = b4.x; bt[i][1] = b4.y; bt[i][2] = b4.z; bt[i][3] = b4.w;
#pragma unroll
    for (int q = 0; q < 4; ++q) ag[i][q] = ab[i][q] = 0.f;
  }
  for (int64_t row = r0 + w; row < r1; row += 4) {
    float v[VPL][4], d[VPL][4];
    float4 dyv[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float4 xv = *reinterpret_cast<const float4*>(x + row * D + (i * 64 + lane) * 4);
      const float4 gv = *reinterpret_cast<const float4*>(dg + row * D + (i * 64 + lane) * 4);
      v[i][0] = xv.x; v[i][1] = xv.y; v[i][2] = xv.z; v[i][3] = xv.w;
      d[i][0] = gv.x; d[i][1] = gv.y; d[i][2] = gv.z; d[i][3] = gv.w;
      s += (xv.x + xv.y) + (xv.z + xv.w);
      dyv[i] = *reinterpret_cast<const float4*>(dy + row * D + (i * 64 + lane) * 4);   // with the row, not behind its four reductions
    }
    const float mean = wave_sum(s) * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[i][q] -= mean;
        ss += v[i][q] * v[i][q];
      }
    const float rstd = rsqrtf(wave_sum(ss) * (1.f / D) + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xh = v[i][q] * rstd;
        const float a = xh * gm[i][q] + bt[i][q];
        const float u = w0 * a + b0;
        const float sg = 1.f / (1.f + __expf(-(w1 * a + b1)));
        const float du = d[i][q] * sg;
        const float dv = d[i][q] * u * sg * (1.f - sg);
        sw0 += du * a; sb0 += du; sw1 += dv * a; sb1 += dv;
        const float da = du * w0 + dv * w1;
        ag[i][q] += da * xh;
        ab[i][q] += da;
        const float gg = da * gm[i][q];
        v[i][q] = xh;
        d[i][q] = gg;
        s1 += gg;
        s2 += gg * xh;
      }
    s1 = wave_sum(s1) * (1.f / D);
    s2 = wave_sum(s2) * (1.f / D);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int64_t o = row * D + (i * 64 + lane) * 4;
      const float4 a = dyv[i];
      float4 r;
      r.x = a.x + rstd * (d[i][0] - s1 - v[i][0] * s2);
      r.y = a.y + rstd * (d[i][1] - s1 - v[i][1] * s2);
      r.z = a.z + rstd * (d[i][2] - s1 - v[i][2] * s2);
      r.w = a.w + rstd * (d[i][3] - s1 - v[i][3] * s2);
      *reinterpret_cast<float4*>(dx + o) = r;
      if (nx.out != nullptr) {  // what the NEXT backward operator multiplies with: mask / (1 - p) o dx, bf16
        const float4 rd = drop_apply4(ndrop, (unsigned)o, r);
        *reinterpret_cast<uint2*>(nx.out + o) = make_uint2(pack_bf2(rd.x, rd.y), pack_bf2(rd.z, rd.w));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      red[w][(i * 64 + lane) * 4 + q] = ag[i][q];
      red[w][D + (i * 64 + lane) * 4 + q] = ab[i][q];
    }
  sw0 = wave_sum(sw0); sw1 = wave_sum(sw1); sb0 = wave_sum(sb0); sb1 = wave_sum(sb1);
  if (lane == 0) { red[w][2 * D] = sw0; red[w][2 * D + 1] = sw1; red[w][2 * D + 2] = sb0; red[w][2 * D + 3] = sb1; }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D + 4; c += 256)
    part[(int64_t)blockIdx.x * (2 * D + 4) + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// ---- passes A', B1, B2 and C as ONE kernel on the forward's tile structure (round 6).  The four launches above move dc
// (written once, read twice with halo) and dg (written, read) through memory and cost 9 + 12 + 16 + 14 us per block for
// ~4 us of traffic each; here a workgroup = (batch b, TT frames, all D channels) keeps both in LDS / registers:
//   0  dc rows of the tile + halo from dy, z and the channel constants (the apply pass, recomputed over the halo) -> LDS
//   1  thread (channel c, half tile): dg = dc (*) flipped taps -> scratch rows in memory (L2); its own 16 dc values stay in registers
//   2  g = glu(LayerNorm(x)) rows of the tile + halo -> the same LDS rows (the forward's phase 1, as MODE 2 recomputes it)
//   3  d dw_w[k][c], d dw_b[c] sums of the half tile from registers; the second half hands its 34 sums over through LDS
//   4  pass C row-wise, dg read back from its scratch rows (NTHR / TT lanes per row): GLU backward + LayerNorm backward + residual -> dx
//      (+ the next operator's bf16 operand), column sums of d ln_g | d ln_b through two LDS tiles, d pw1 scalars
// One LDS buffer of TT + KS + 1 = 66 rows serves every step (67.6 KB at D = 256: two workgroups per CU).
// Partials: part_b [tile][(KS + 1) D] (as MODE 2 leaves them), part_c [tile][2D + 4] (glu_ln_bwd_kernel's layout, one slab per
// tile instead of one per 16 rows).
struct ConvBwdArgs {
  const float *x, *dy, *z, *stats, *sums;
  csship_conv_weights w;
  float inv_n;   // 1 / rows of the batch statistics; 0: running statistics (constants of the graph)
  float* dx;
  float* dg;       // [B, T, D] scratch: written in step 1, read back (from L2) by the same workgroup's pass C
  float *part_b, *part_c;
  int B, T;
  csship_dropout_cfg dcfg;
  NextOut nx;
  unsigned long long* stamps;   // diagnostic (csship_debug_stamps): phase boundaries of every workgroup
};

template <int LR>
DEV float sum_row_lanes(float v) {  // over aligned groups of LR = 16 / 32 lanes, result in all of them
  static_assert(LR == 8 || LR == 16 || LR == 32, "pass C rows are 8, 16 or 32 lanes");
  v = sum8(v);
  if constexpr (LR >= 16) v += dpp_mov<DPP_MIRROR>(v);
  if constexpr (LR == 32) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const unsigned a0 = r[0], a1 = r[1];
    v = __uint_as_float(a0) + __uint_as_float(a1);
  }
  return v;
}

// TT = 32 (the default): 67.6 KB of LDS and <= 128 registers, two workgroups per CU (any T, any B).  TT = 50 (d_model 256 only,
// opt-in): 102 KB and up to 256 registers, ONE workgroup per CU - meant for launches whose 32-frame tiles outnumber the CUs while
// their 50-frame tiles do not (32 x 400 frames: 416 tiles leave 160 CUs with two workgroups and 96 with one - 37 us; 256 tiles are
// one equal round) and measured slower (see run_bwd).  Everything below is written in TT; the row phases round it up to whole
// lane groups.
template <int D, int TT, int KS>
DEV void conv_bwd_fused_body(const ConvBwdArgs& a, float (*G)[D], float (*red)[4]) {
  constexpr int ROWS = TT + KS - 1, NTHR = 2 * D, C4 = D / 4;
  static_assert(NTHR % C4 == 0 && TT % 2 == 0, "whole float4 columns per thread in step 0; two half tiles");
  const int tid = threadIdx.x;
  const int b = blockIdx.y, t0 = blockIdx.x * TT;
  const int H3 = a.w.causal ? 0 : (KS - 1) / 2;        // rows of the dc buffer in front of the tile (input gradient: flipped taps)
  const int H2 = a.w.causal ? KS - 1 : (KS - 1) / 2;   // rows of the g buffer in front of the tile (the forward's halo)

  // the x rows of step 2's first row batch (this thread's 32 channels of row tid / LPR) are requested inside step 0, behind its
  // last dy / z loads: their latency passes behind the rest of step 0 and step 1
  constexpr int LPR = D / 32, RPB = NTHR / LPR;   // 64 rows per batch
  static_assert(LPR == 8 || LPR == 16, "row reductions are written for 8 or 16 lanes per row");
  const int xrow = tid / LPR, xpart = tid % LPR;
  stamp(a.stamps, 0);
  constexpr bool PRE = D == 256;   // (D = 512: 1024 threads hold 128 registers each, the rows would spill: requested in step 2)
  float4 xv[8];
  auto x_row_ptr = [&](int row) {
    const int t = t0 - H2 + row;
    return a.x + ((size_t)b * a.T + (t >= 0 && t < a.T ? t : 0)) * D;
  };

  // ---- step 0: dc = sc * (dzh - S1 / n - zn * S2 / n), dzh = dy o mask / (1 - p) * w3 * [zh > 0]  (conv_bn_bwd_apply_kernel)
  {
    const Drop dp = drop_make(a.dcfg.seed, a.dcfg.stream, a.dcfg.p);
    const int c4 = tid % C4, c = c4 * 4;
    const float4 sc4 = *reinterpret_cast<const float4*>(a.stats + c), sh4 = *reinterpret_cast<const float4*>(a.stats + D + c);
    const float4 mu4 = *reinterpret_cast<const float4*>(a.stats + 2 * D + c), va4 = *reinterpret_cast<const float4*>(a.stats + 3 * D + c);
    const float4 s14 = *reinterpret_cast<const float4*>(a.sums + c), s24 = *reinterpret_cast<const float4*>(a.sums + D + c);
    const float w3 = a.w.pw2_w[0], inv_n = a.inv_n;
    const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w}, muv[4] = {mu4.x, mu4.y, mu4.z, mu4.w};
    const float rsv[4] = {rsqrtf(va4.x + 1e-5f), rsqrtf(va4.y + 1e-5f), rsqrtf(va4.z + 1e-5f), rsqrtf(va4.w + 1e-5f)};
    const float s1v[4] = {s14.x, s14.y, s14.z, s14.w}, s2v[4] = {s24.x, s24.y, s24.z, s24.w};
    // dc = sc * (dzh - S1 / n - (z - mu) * rstd * S2 / n) = k3 * d * [zh > 0] + (c1 * z + c0) with per-channel k3 = sc * w3,
    // c1 = -sc * rstd * S2 / n, c0 = -sc * S1 / n - c1 * mu: three multiply-adds, a compare and a select per element
    float k3v[4], c1v[4], c0v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      k3v[q] = scv[q] * w3;
      c1v[q] = -scv[q] * rsv[q] * s2v[q] * inv_n;
      c0v[q] = -scv[q] * s1v[q] * inv_n - c1v[q] * muv[q];
    }
    // rows in flight per thread: TT = 32 - two batches of 4 x (dy, z) under the 128-register budget; TT = 50 - all 11 at once
    constexpr int RSTEP = NTHR / C4, NIT = (ROWS + RSTEP - 1) / RSTEP, NB = TT == 32 ? 4 : NIT;
    constexpr int NBAT = (NIT + NB - 1) / NB;
#pragma unroll
    for (int bi = 0; bi < NBAT; ++bi) {
      const int k0 = bi * NB;
      float4 dv[NB], zv[NB];
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int row = tid / C4 + (k0 + k) * RSTEP, t = t0 - H3 + row;
        const size_t i = ((size_t)b * a.T + (row < ROWS && t >= 0 && t < a.T ? t : 0)) * C4 + c4;
        dv[k] = reinterpret_cast<const float4*>(a.dy)[i];
        zv[k] = reinterpret_cast<const float4*>(a.z)[i];
      }
      if constexpr (PRE) {
        if (bi + 1 == NBAT) {   // behind the last dy / z requests (loads return in order: in front of them they held step 0 back)
          const float* const xr = x_row_ptr(xrow);
#pragma unroll
          for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const float4*>(xr + (i * LPR + xpart) * 4);
        }
      }
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int row = tid / C4 + (k0 + k) * RSTEP, t = t0 - H3 + row;
        const bool inside = t >= 0 && t < a.T;
        const size_t i = ((size_t)b * a.T + (row < ROWS && inside ? t : 0)) * C4 + c4;
        const float4 d4 = drop_apply4(dp, (unsigned)(i * 4), dv[k]);
        const float d[4] = {d4.x, d4.y, d4.z, d4.w}, zz[4] = {zv[k].x, zv[k].y, zv[k].z, zv[k].w};
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float zh = zz[q] * scv[q] + shv[q];
          const float lin = c1v[q] * zz[q] + c0v[q];
          o[q] = inside ? (zh > 0.f ? k3v[q] * d[q] + lin : lin) : 0.f;
        }
        if (ROWS % RSTEP == 0 || row < ROWS) *reinterpret_cast<float4*>(&G[row][c]) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  __syncthreads();
  stamp(a.stamps, 1);

  // ---- step 1: dg of this thread's channel and half tile (convmod_kernel MODE 3), its own dc values kept
  const int c = tid % D, half = tid / D;
  constexpr int TH = TT / 2;
  const int tb0 = half * TH;
  float dcv[TH];
  {
    float wk[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) wk[k] = a.w.dw_w[(KS - 1 - k) * D + c];  // [k][D], flipped
    constexpr int TQ = TH % 8 == 0 ? 8 : 5;   // passes of 8 (or 5) outputs: 40 dc values + 33 taps live beside the prefetched x rows
    static_assert(TH % TQ == 0, "whole output passes");
#pragma unroll
    for (int j0 = 0; j0 < TH; j0 += TQ) {
      float dcol[TQ + KS - 1];
#pragma unroll
      for (int u = 0; u < TQ + KS - 1; ++u) dcol[u] = G[tb0 + j0 + u][c];
#pragma unroll
      for (int j = 0; j < TQ; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) acc += wk[k] * dcol[j + k];
        if (t0 + tb0 + j0 + j < a.T) a.dg[((size_t)b * a.T + t0 + tb0 + j0 + j) * D + c] = acc;
      }
    }
#pragma unroll
    for (int j = 0; j < TH; ++j) dcv[j] = G[tb0 + j + (a.w.causal ? 0 : (KS - 1) / 2)][c];
  }
  __syncthreads();  // every thread has its dc column in registers
  stamp(a.stamps, 2);

  // ---- step 2: g = glu(LayerNorm(x)) rows (tile + halo) into the same rows: convmod_kernel's phase 1, 64 rows per batch (the
  // first batch on the prefetched rows)
  {
    const float w0 = a.w.pw1_w[0], w1 = a.w.pw1_w[1], b0 = a.w.pw1_b[0], b1 = a.w.pw1_b[1];
#pragma unroll
    for (int r0 = 0; r0 < ROWS; r0 += RPB) {
      const int row = r0 + xrow;
      if (ROWS % RPB != 0 && row >= ROWS) break;
      const int t = t0 - H2 + row;
      const bool xinside = t >= 0 && t < a.T;
      if (!PRE || r0 > 0) {
        const float* const xr = x_row_ptr(row);
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const float4*>(xr + (i * LPR + xpart) * 4);
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
      s = sum8(s);
      if constexpr (LPR == 16) s += dpp_mov<DPP_MIRROR>(s);
      const float mean = s * (1.f / D);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
        ss += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
      }
      ss = sum8(ss);
      if constexpr (LPR == 16) ss += dpp_mov<DPP_MIRROR>(ss);
      const float rstd = rsqrtf(ss * (1.f / D) + 1e-5f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int cc = (i * LPR + xpart) * 4;
        const float4 gm = *reinterpret_cast<const float4*>(a.w.ln_g + cc);
        const float4 bt = *reinterpret_cast<const float4*>(a.w.ln_b + cc);
        float e[4] = {xv[i].x * rstd * gm.x + bt.x, xv[i].y * rstd * gm.y + bt.y, xv[i].z * rstd * gm.z + bt.z,
                      xv[i].w * rstd * gm.w + bt.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
          e[q] = xinside ? (w0 * e[q] + b0) * __builtin_amdgcn_rcpf(1.f + __expf(-(w1 * e[q] + b1))) : 0.f;
        *reinterpret_cast<float4*>(&G[row][cc]) = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  }
  __syncthreads();
  stamp(a.stamps, 3);

  // ---- step 3: d dw_w[k][c] = sum_t dc[t][c] * g[t + k - halo][c], d dw_b[c] = sum_t dc[t][c]  (convmod_kernel MODE 2)
  float sk[KS], sdc = 0.f;
  {
    float gv[TH + KS - 1];
#pragma unroll
    for (int u = 0; u < TH + KS - 1; ++u) gv[u] = G[tb0 + u][c];
#pragma unroll
    for (int j = 0; j < TH; ++j) sdc += dcv[j];
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      float sacc = 0.f;
#pragma unroll
      for (int j = 0; j < TH; ++j) sacc += dcv[j] * gv[j + k];
      sk[k] = sacc;
    }
  }
  __syncthreads();  // every thread has taken its g rows
  stamp(a.stamps, 4);

  // ---- step 4: the second half tile hands its 34 sums to the first through rows [0, KS + 1); pass C's d ln_g terms go to rows
  // [CA0, CA0 + TT), its d ln_b terms to rows [0, TT)
  float* const X = &G[0][0];
  constexpr int CA0 = KS + 1 > TT ? KS + 1 : TT;   // behind the hand-over block AND behind the d ln_b rows
  if (half == 1) {
#pragma unroll
    for (int k = 0; k < KS; ++k) X[k * D + c] = sk[k];
    X[KS * D + c] = sdc;
  }
  __syncthreads();
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  if (half == 0) {
    float* out = a.part_b + blk * (KS + 1) * D + c;
#pragma unroll
    for (int k = 0; k < KS; ++k) out[k * D] = sk[k] + X[k * D + c];
    out[KS * D] = sdc + X[KS * D + c];
  }
  __syncthreads();  // the hand-over block has been read: rows [0, TT) take pass C's d ln_b terms
  stamp(a.stamps, 5);

  // ---- pass C (glu_ln_bwd_kernel): LR lanes per row (the largest power of two with NTHR / LR >= TT rows), D / LR channels per
  // lane as float4 pieces
  constexpr int LR = NTHR / TT >= 32 ? 32 : NTHR / TT >= 16 ? 16 : 8, NP = D / (4 * LR);
  static_assert(NTHR / LR >= TT && NP * 4 * LR == D, "every tile row has its lane group");
  const int row = tid / LR, part = tid % LR;
  const int t = t0 + row;
  const bool rowok = NTHR / LR == TT || row < TT;
  const bool valid = rowok && t < a.T;
  const size_t rbase = ((size_t)b * a.T + (valid ? t : a.T - 1)) * D;
  float xh[NP][4], dd[NP][4];
  float4 dyv[NP];
  float sw0 = 0.f, sw1 = 0.f, sb0 = 0.f, sb1 = 0.f, rstd;
  {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int cc = (i * LR + part) * 4;
      const float4 xq = *reinterpret_cast<const float4*>(a.x + rbase + cc);
      dyv[i] = *reinterpret_cast<const float4*>(a.dy + rbase + cc);
      const float4 gv = *reinterpret_cast<const float4*>(a.dg + rbase + cc);   // (this workgroup's step 1, barriers ago)
      xh[i][0] = xq.x; xh[i][1] = xq.y; xh[i][2] = xq.z; xh[i][3] = xq.w;
      dd[i][0] = gv.x; dd[i][1] = gv.y; dd[i][2] = gv.z; dd[i][3] = gv.w;
      s += (xq.x + xq.y) + (xq.z + xq.w);
    }
    const float mean = sum_row_lanes<LR>(s) * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        xh[i][q] -= mean;
        ss += xh[i][q] * xh[i][q];
      }
    rstd = rsqrtf(sum_row_lanes<LR>(ss) * (1.f / D) + 1e-5f);
  }
  float s1 = 0.f, s2 = 0.f;
  {
    const float w0 = a.w.pw1_w[0], w1 = a.w.pw1_w[1], b0 = a.w.pw1_b[0], b1 = a.w.pw1_b[1];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int cc = (i * LR + part) * 4;
      const float4 g4 = *reinterpret_cast<const float4*>(a.w.ln_g + cc);
      const float4 b4 = *reinterpret_cast<const float4*>(a.w.ln_b + cc);
      const float gm[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
      float caq[4], cbq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float h = xh[i][q] * rstd;
        const float av = h * gm[q] + bt[q];
        const float u = w0 * av + b0;
        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-(w1 * av + b1)));   // (v_rcp_f32, 1 ulp, as the forward's GLU)
        const float dgq = valid ? dd[i][q] : 0.f;
        const float du = dgq * sg;
        const float dv = dgq * u * sg * (1.f - sg);
        sw0 += du * av; sb0 += du; sw1 += dv * av; sb1 += dv;
        const float da = du * w0 + dv * w1;
        caq[q] = da * h;
        cbq[q] = da;
        const float gg = da * gm[q];
        xh[i][q] = h;
        dd[i][q] = gg;
        s1 += gg;
        s2 += gg * h;
      }
      if (rowok) {
        *reinterpret_cast<float4*>(&G[CA0 + row][cc]) = make_float4(caq[0], caq[1], caq[2], caq[3]);
        *reinterpret_cast<float4*>(&G[row][cc]) = make_float4(cbq[0], cbq[1], cbq[2], cbq[3]);
      }
    }
  }
  s1 = sum_row_lanes<LR>(s1) * (1.f / D);
  s2 = sum_row_lanes<LR>(s2) * (1.f / D);
  if (valid) {
    const Drop ndrop = drop_make(a.nx.seed, a.nx.stream, a.nx.p);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const size_t o = rbase + (i * LR + part) * 4;
      const float4 y4 = dyv[i];
      float4 r;
      r.x = y4.x + rstd * (dd[i][0] - s1 - xh[i][0] * s2);
      r.y = y4.y + rstd * (dd[i][1] - s1 - xh[i][1] * s2);
      r.z = y4.z + rstd * (dd[i][2] - s1 - xh[i][2] * s2);
      r.w = y4.w + rstd * (dd[i][3] - s1 - xh[i][3] * s2);
      *reinterpret_cast<float4*>(a.dx + o) = r;
      if (a.nx.out != nullptr) {  // what the NEXT backward operator multiplies with: mask / (1 - p) o dx, bf16
        const float4 rd = drop_apply4(ndrop, (unsigned)o, r);
        *reinterpret_cast<uint2*>(a.nx.out + o) = make_uint2(pack_bf2(rd.x, rd.y), pack_bf2(rd.z, rd.w));
      }
    }
  }
  stamp(a.stamps, 6);
  sw0 = wave_sum(sw0); sw1 = wave_sum(sw1); sb0 = wave_sum(sb0); sb1 = wave_sum(sb1);
  if ((tid & 63) == 0) { red[tid >> 6][0] = sw0; red[tid >> 6][1] = sw1; red[tid >> 6][2] = sb0; red[tid >> 6][3] = sb1; }
  __syncthreads();
  {
    float acc = 0.f;  // thread (c, half): column c of d ln_g (half 0) / d ln_b (half 1), rows in order
#pragma unroll 8
    for (int rr = 0; rr < TT; ++rr) acc += G[(half == 0 ? CA0 : 0) + rr][c];
    float* out = a.part_c + blk * (2 * D + 4);
    out[half * D + c] = acc;
    if (tid < 4) {
      float sc = 0.f;
#pragma unroll
      for (int wv = 0; wv < NTHR / 64; ++wv) sc += red[wv][tid];
      out[2 * D + tid] = sc;
    }
  }
  stamp(a.stamps, 7);
}

template <int TT, int KS>
constexpr int conv_bwd_lds_rows() { return (KS + 1 > TT ? KS + 1 : TT) + TT > TT + KS - 1 ? (KS + 1 > TT ? KS + 1 : TT) + TT : TT + KS - 1; }

template <int D, int TT, int KS>
__global__ __launch_bounds__(2 * D) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_bwd_fused_kernel(ConvBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float G[conv_bwd_lds_rows<TT, KS>()][D];
  __shared__ float red[2 * D / 64][4];
  conv_bwd_fused_body<D, TT, KS>(a, G, red);
}

template <int D, int TT, int KS>   // one workgroup per CU: the whole register file
__global__ __launch_bounds__(2 * D) void conv_bwd_fused1_kernel(ConvBwdArgs a) {
  __shared__ __attribute__((aligned(16))) float G[conv_bwd_lds_rows<TT, KS>()][D];
  __shared__ float red[2 * D / 64][4];
  conv_bwd_fused_body<D, TT, KS>(a, G, red);
}

// rows per workgroup of the row-wise backward passes: short row loops, many waves in flight (these passes are chains of
// loads and wave reductions, not bandwidth: 64 rows per workgroup left less than one wave per SIMD at 12 800 rows)
constexpr int BWD_ROWS = 16;
static_assert(BWD_ROWS == BWD_ROWS_C, "conv_bn_bwd_stats_kernel's unrolled form");

struct BwdLayout {  // carve of the backward workspace (floats)
  size_t dc, dg, part, part_c, sums, total;
  int nwg_rows, nblk;
};
BwdLayout bwd_layout(int B, int T, int D, int TT) {
  BwdLayout L{};
  const size_t n = (size_t)B * T * D;
  L.nwg_rows = (int)(((size_t)B * T + BWD_ROWS - 1) / BWD_ROWS);
  L.nblk = ((T + TT - 1) / TT) * B * 2;
  const size_t p = (size_t)L.nwg_rows * (2 * D + 4) > (size_t)L.nblk * 34 * D ? (size_t)L.nwg_rows * (2 * D + 4)
                                                                             : (size_t)L.nblk * 34 * D;
  // part: the partials of pass A, then of pass B; part_c: pass C's own; sums: S1 | S2 for the apply pass
  const size_t pc = ((size_t)L.nwg_rows * (2 * D + 4) + 3) / 4 * 4;
  L.dc = 0; L.dg = n; L.part = 2 * n; L.part_c = 2 * n + (p + 3) / 4 * 4; L.sums = L.part_c + pc; L.total = L.sums + 2 * (size_t)D;
  return L;
}

template <int D, int TT>
// phase 0: everything; SyncBN: phase 1 = pass A (+ this rank's S1 | S2 copied to the workspace's sums block, which the
// caller all-reduces), phase 2 = the rest with those sums and the GLOBAL row count; phase 3 = the rest with the sums pass A's
// stand-in left in `grads` (csship_convmod_bwd_after_stats)
int run_bwd(const float* x, const float* dy, const csship_conv_weights& w, int B, int T, const float* fwd_ws, float* dx,
            float* grads, float* ws, int bn_eval, hipStream_t s, csship_dropout_cfg dcfg, int phase = 0,
            double total_rows = 0.0, NextOut nx = NextOut{}) {
  const BwdLayout L = bwd_layout(B, T, D, TT);
  const int64_t M = (int64_t)B * T;
  const size_t n = (size_t)M * D;
  const float* z = fwd_ws;
  const float* stats = fwd_ws + convmod_fwd_stats_offset(B, T, D);
  float *dc = ws + L.dc, *dg = ws + L.dg, *part = ws + L.part, *part_c = ws + L.part_c;
  // grads layout (CSSHIP_CONV_GRADS_*): [bn_b D | bn_g D | pw2_w 1 | pw2_b 1 | pad 2][dw_w 33*D | dw_b D][ln_g D | ln_b D | pw1_w 2 | pw1_b 2]
  float* gA = grads;
  float* gB = grads + 2 * D + 4;
  float* gC = gB + 34 * D;
  float* sums = ws + L.sums;
  int rc = 0;
  if (phase != 2 && phase != 3) {   // (3: pass A and its reduction were run by the operator in front - csship_ffn_bwd_conv - into gA)
    // pass A
    hipLaunchKernelGGL((conv_bn_bwd_stats_kernel<D>), dim3(L.nwg_rows), dim3(D), 0, s, dy, z, stats, w.pw2_w, M, BWD_ROWS, part, dcfg);
    CSSHIP_CHECK_LAUNCH();
    rc = launch_slab_reduce(part, L.nwg_rows, 2 * D + 2, 2 * D + 2, 2 * D + 2, 1.f, nullptr, nullptr, 0, gA, s);
    if (rc) return rc;
    if (phase == 1) {
      if (hipMemcpyAsync(sums, gA, 2 * (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return CSSHIP_ELAUNCH;
      return CSSHIP_OK;
    }
  }
  const float inv_n = bn_eval ? 0.f : 1.f / (float)(phase == 2 ? total_rows : (double)M);
  static const bool unfused = [] { const char* e = getenv("CSSHIP_CONV_BWD_UNFUSED"); return e != nullptr && atoi(e) != 0; }();
  if (!unfused) {
    // passes A', B1, B2, C as one launch (conv_bwd_fused_kernel).  Both reductions behind it yield parameter gradients only:
    // they may wait for the block's flush
    // 50-frame tiles, one workgroup per CU: for launches whose 32-frame tiles are more than one round of the chip (416 at 32 x 400)
    // while the 50-frame ones are exactly one - built, correct, and slower:
    constexpr int TT1 = 50;
    static const int tt_env = [] { const char* e = getenv("CSSHIP_CONV_BWD_TT"); return e ? atoi(e) : 0; }();   // A/B: 32 / 50
    const int nt32 = (T + TT - 1) / TT, nt50 = (T + TT1 - 1) / TT1;
    // (measured, 32 x 400 frames: 36.4 us per workgroup - 8 waves alone on a CU leave every phase latency-bound, the row pass 9 us
    // against 3.8 - and 7.66 against 7.61 ms per training step: opt-in, CSSHIP_CONV_BWD_TT=50)
    const bool wide_tiles = D == 256 && TT == 32 && tt_env == TT1;
    const int ntx = wide_tiles ? nt50 : nt32, ntile = ntx * B;
    float* part_b = part;   // (pass A's partials there were consumed by its reduction, which is never deferred)
    ConvBwdArgs fa{x, dy, z, stats, phase == 2 ? sums : gA, w, inv_n, dx, dg, part_b, part_c, B, T, dcfg, nx, g_csship_stamps};
    if constexpr (D == 256) {
      if (wide_tiles) hipLaunchKernelGGL((conv_bwd_fused1_kernel<D, TT1, 33>), dim3(ntx, B), dim3(2 * D), 0, s, fa);
      else hipLaunchKernelGGL((conv_bwd_fused_kernel<D, TT, 33>), dim3(ntx, B), dim3(2 * D), 0, s, fa);
    } else {
      hipLaunchKernelGGL((conv_bwd_fused_kernel<D, TT, 33>), dim3(ntx, B), dim3(2 * D), 0, s, fa);
    }
    CSSHIP_CHECK_LAUNCH();
    rc = launch_slab_reduce(part_b, ntile, (int64_t)34 * D, 34 * D, (int64_t)34 * D, 1.f, nullptr, nullptr, 0, gB, s, true);
    if (rc) return rc;
    return launch_slab_reduce(part_c, ntile, 2 * D + 4, 2 * D + 4, 2 * D + 4, 1.f, nullptr, nullptr, 0, gC, s, true);
  }
  // pass A'
  // running statistics are constants of the graph: no mean / projection terms (1/n -> 0)
  hipLaunchKernelGGL(conv_bn_bwd_apply_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, dy, z, stats,
                     phase == 2 ? sums : gA, w.pw2_w, bn_eval ? 0.f : 1.f / (float)(phase == 2 ? total_rows : (double)M), n / 4,
                     D, dc, dcfg);
  CSSHIP_CHECK_LAUNCH();
  // pass B1 / B2 on the forward's tile structure
  dim3 grid((T + TT - 1) / TT, B);
  ConvArgs a{x, dg, w, dc, nullptr, nullptr, nullptr, B, T, nullptr};
  hipLaunchKernelGGL((convmod_kernel<D, TT, 33, 3, true>), grid, dim3(2 * D), 0, s, a);
  CSSHIP_CHECK_LAUNCH();
  ConvArgs a2{x, nullptr, w, nullptr, nullptr, part, dc, B, T, nullptr};
  hipLaunchKernelGGL((convmod_kernel<D, TT, 33, 2, false>), grid, dim3(2 * D), 0, s, a2);
  CSSHIP_CHECK_LAUNCH();
  rc = launch_slab_reduce(part, L.nblk / 2, (int64_t)34 * D, 34 * D, (int64_t)34 * D, 1.f, nullptr, nullptr, 0, gB, s);   // (one partial per tile: the halves are added in the kernel)
  if (rc) return rc;
  // pass C
  hipLaunchKernelGGL((glu_ln_bwd_kernel<D / 256>), dim3(L.nwg_rows), dim3(256), 0, s, x, dg, dy, w.ln_g, w.ln_b, w.pw1_w,
                     w.pw1_b, M, BWD_ROWS, dx, part_c, nx);
  CSSHIP_CHECK_LAUNCH();
  // (parameter gradients only: may join the block's deferred reductions, csship_reduce_defer)
  return launch_slab_reduce(part_c, L.nwg_rows, 2 * D + 4, 2 * D + 4, 2 * D + 4, 1.f, nullptr, nullptr, 0, gC, s, true);
}

}  // namespace

size_t convmod_fwd_stats_offset(int B, int T, int D) {   // (as run_bwd finds them: z, the tile partials, the f64 sums, then the statistics)
  const int TT = D == 256 ? TT256 : TT512;
  const size_t fwd_nblk = (size_t)((T + TT - 1) / TT) * B * 2;
  return (size_t)B * T * D + fwd_nblk * 2 * D + 4 * (size_t)D;
}

extern "C" size_t csship_convmod_workspace(int32_t B, int32_t T, int32_t D) {
  const int TT = D == 256 ? TT256 : TT512;
  const size_t nblk = (size_t)((T + TT - 1) / TT) * B * 2;
  return ((size_t)B * T * D + nblk * 2 * D + 4 * (size_t)D + 4 * (size_t)D) * sizeof(float);  // f64 [2D] sums (SyncBN), then stats
}

extern "C" size_t csship_convmod_sums_offset(int32_t B, int32_t T, int32_t D) {
  const int TT = D == 256 ? TT256 : TT512;
  const size_t nblk = (size_t)((T + TT - 1) / TT) * B * 2;
  return ((size_t)B * T * D + nblk * 2 * D) * sizeof(float);
}

extern "C" int csship_convmod(const float* x, float* y, const csship_conv_weights* w, int32_t B, int32_t T,
                              int32_t D, int32_t kernel_size, int32_t bn_mode, void* workspace,
                              csship_stream_t stream) {
  return csship_convmod_drop(x, y, w, B, T, D, kernel_size, bn_mode, workspace, nullptr, stream);
}

extern "C" int csship_convmod_drop(const float* x, float* y, const csship_conv_weights* w, int32_t B, int32_t T,
                                   int32_t D, int32_t kernel_size, int32_t bn_mode, void* workspace,
                                   const csship_dropout_cfg* drop, csship_stream_t stream) {
  if (!x || !y || !w) return CSSHIP_ENULL;
  const bool dropping = drop && drop->seed && drop->p > 0.f;
  if (dropping && (bn_mode == 0 || (int64_t)B * T * D > 0xffffffffLL)) return CSSHIP_EBADSHAPE;  // training forward only
  const csship_dropout_cfg dc = dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f};
  if (!w->ln_g || !w->ln_b || !w->pw1_w || !w->pw1_b || !w->dw_w || !w->dw_b || !w->bn_g || !w->bn_b ||
      !w->pw2_w || !w->pw2_b || (bn_mode != 1 && (!w->bn_mean || !w->bn_var)))
    return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  if (kernel_size != 33 || (D != 256 && D != 512) || bn_mode < 0 || bn_mode > 2) return CSSHIP_EUNSUPPORTED;
  if (x == y) return CSSHIP_EBADSHAPE;  // neighbouring tiles read the halo rows: not in-place safe
  if (!aligned16(x) || !aligned16(y) || !aligned16(w->ln_g) || !aligned16(w->ln_b) || (workspace && !aligned16(workspace)))
    return CSSHIP_EALIGN;
  ConvArgs a{x, y, *w, nullptr, nullptr, nullptr, nullptr, B, T, g_csship_stamps};
  if (D == 256) return run<256, TT256>(a, bn_mode, workspace, as_stream(stream), dc);
  return run<512, TT512>(a, bn_mode, workspace, as_stream(stream), dc);
}

extern "C" int csship_outproj_conv_train(const uint16_t* attn_out, const uint16_t* wo_packed, const float* bo, const float* x,
                                         float* x2_out, float* y, const csship_conv_weights* w, int32_t B, int32_t T,
                                         int32_t D, int32_t kernel_size, const csship_dropout_cfg* drop_proj,
                                         const csship_dropout_cfg* drop_conv, void* workspace, csship_stream_t stream) {
  if (!attn_out || !wo_packed || !bo || !x || !x2_out || !w || !workspace) return CSSHIP_ENULL;   // (y NULL: no apply pass)
  if (w->causal) return CSSHIP_EUNSUPPORTED;
  if (!w->ln_g || !w->ln_b || !w->pw1_w || !w->pw1_b || !w->dw_w || !w->dw_b || !w->bn_g || !w->bn_b || !w->pw2_w || !w->pw2_b)
    return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535 || (int64_t)B * T * D > 0xffffffffLL) return CSSHIP_EBADSHAPE;
  if (kernel_size != 33 || D != 256) return CSSHIP_EUNSUPPORTED;
  if (x == x2_out || x2_out == y) return CSSHIP_EBADSHAPE;
  if (!aligned16(attn_out) || !aligned16(wo_packed) || !aligned16(bo) || !aligned16(x) || !aligned16(x2_out) || (y && !aligned16(y)) ||
      !aligned16(w->ln_g) || !aligned16(w->ln_b) || !aligned16(workspace))
    return CSSHIP_EALIGN;
  const csship_dropout_cfg none{nullptr, 0u, 0.f};
  auto cfg = [&](const csship_dropout_cfg* c) { return (c && c->seed && c->p > 0.f) ? *c : none; };
  hipStream_t s = as_stream(stream);
  // the workspace csship_convmod(bn_mode 1) would leave (z | partials | f64 sums | scale, shift, mean, var), so that
  // csship_convmod_bwd* reads it unchanged: the statistics sit behind the partial block of the 32-frame tiling
  const size_t n = (size_t)B * T * D;
  const size_t nblk32 = (size_t)((T + TT256 - 1) / TT256) * B * 2;
  float* zbuf = reinterpret_cast<float*>(workspace);
  float* partial = zbuf + n;
  float* scale_shift = partial + nblk32 * 2 * D + 4 * D;
  int nblk = 0;
  TimelineScope tl(CSSHIP_TL_OUTPROJ_CONV, s);   // three launches: the fused kernel, BatchNorm reduce, BatchNorm apply
  const int rc = launch_outproj_conv_train(attn_out, wo_packed, bo, x, x2_out, *w, B, T, cfg(drop_proj), zbuf, partial, &nblk, s);
  if (rc) return rc;
  if ((size_t)nblk > nblk32) return CSSHIP_EWORKSPACE;   // (50-frame tiles: never more blocks than the 32-frame tiling)
  hipLaunchKernelGGL(bn_reduce8_kernel, dim3(D / 8), dim3(256), 0, s, partial, nblk, D, (double)B * (double)T, w->bn_g, w->bn_b,
                     scale_shift);
  CSSHIP_CHECK_LAUNCH();
  if (y == nullptr) return CSSHIP_OK;   // the consumer applies the statistics (csship_encoder_chain_train, bn_* fields)
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x2_out, zbuf, scale_shift, w->pw2_w,
                     w->pw2_b, y, n / 4, D, cfg(drop_conv));
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

extern "C" int csship_convmod_stats(const float* x, const csship_conv_weights* w, int32_t B, int32_t T, int32_t D,
                                    int32_t kernel_size, void* workspace, csship_stream_t stream) {
  if (!x || !w || !workspace) return CSSHIP_ENULL;
  if (!w->ln_g || !w->ln_b || !w->pw1_w || !w->pw1_b || !w->dw_w || !w->dw_b) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  if (kernel_size != 33 || (D != 256 && D != 512)) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(workspace)) return CSSHIP_EALIGN;
  ConvArgs a{x, nullptr, *w, nullptr, nullptr, nullptr, nullptr, B, T, nullptr};
  const csship_dropout_cfg none{nullptr, 0u, 0.f};
  if (D == 256) return run<256, TT256>(a, 1, workspace, as_stream(stream), none, 1);
  return run<512, TT512>(a, 1, workspace, as_stream(stream), none, 1);
}

extern "C" int csship_convmod_finish(const float* x, float* y, const csship_conv_weights* w, int32_t B, int32_t T, int32_t D,
                                     double total_rows, void* workspace, const csship_dropout_cfg* drop,
                                     csship_stream_t stream) {
  if (!x || !y || !w || !workspace) return CSSHIP_ENULL;
  if (!w->bn_g || !w->bn_b || !w->pw2_w || !w->pw2_b) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535 || !(total_rows >= (double)B * T)) return CSSHIP_EBADSHAPE;
  if (D != 256 && D != 512) return CSSHIP_EUNSUPPORTED;
  if (x == y) return CSSHIP_EBADSHAPE;
  if (!aligned16(x) || !aligned16(y) || !aligned16(workspace)) return CSSHIP_EALIGN;
  const bool dropping = drop && drop->seed && drop->p > 0.f;
  if (dropping && (int64_t)B * T * D > 0xffffffffLL) return CSSHIP_EBADSHAPE;
  ConvArgs a{x, y, *w, nullptr, nullptr, nullptr, nullptr, B, T, nullptr};
  const csship_dropout_cfg dc = dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f};
  if (D == 256) return run<256, TT256>(a, 1, workspace, as_stream(stream), dc, 2, total_rows);
  return run<512, TT512>(a, 1, workspace, as_stream(stream), dc, 2, total_rows);
}

extern "C" int csship_conv_from_glu(const float* x, const float* g, float* y, const csship_conv_weights* w, int32_t B,
                                    int32_t T, int32_t D, int32_t kernel_size, csship_stream_t stream) {
  if (!x || !g || !y || !w) return CSSHIP_ENULL;
  if (!w->dw_w || !w->dw_b || !w->bn_g || !w->bn_b || !w->bn_mean || !w->bn_var || !w->pw2_w || !w->pw2_b) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  if (kernel_size != 33 || (D != 256 && D != 512)) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(g) || !aligned16(y) || g == y) return CSSHIP_EALIGN;
  ConvArgs a{x, y, *w, g, nullptr, nullptr, nullptr, B, T, g_csship_stamps};
  if (D == 256) return run<256, TT256>(a, 0, nullptr, as_stream(stream));
  return run<512, TT512>(a, 0, nullptr, as_stream(stream));
}

extern "C" size_t csship_convmod_bwd_workspace(int32_t B, int32_t T, int32_t D) {
  return bwd_layout(B, T, D, D == 256 ? TT256 : TT512).total * sizeof(float);
}

extern "C" int csship_convmod_bwd(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                                  int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx,
                                  float* grads, int32_t bn_eval, void* workspace, size_t workspace_bytes,
                                  csship_stream_t stream) {
  return csship_convmod_bwd_drop(x, dy, w, B, T, D, kernel_size, fwd_workspace, dx, grads, bn_eval, nullptr, workspace,
                                 workspace_bytes, stream);
}

static int convmod_bwd_impl(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T, int32_t D,
                            int32_t kernel_size, const void* fwd_workspace, float* dx, float* grads, int32_t bn_eval,
                            const csship_dropout_cfg* drop, void* workspace, size_t workspace_bytes, csship_stream_t stream,
                            int phase, double total_rows, const csship_dropout_cfg* next_drop = nullptr,
                            uint16_t* dx_next_bf16 = nullptr);

extern "C" int csship_convmod_bwd_next(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                                       int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx,
                                       float* grads, int32_t bn_eval, const csship_dropout_cfg* drop,
                                       const csship_dropout_cfg* next_drop, uint16_t* dx_next_bf16, void* workspace,
                                       size_t workspace_bytes, csship_stream_t stream) {
  if (!dx_next_bf16 || !aligned8(dx_next_bf16)) return dx_next_bf16 ? CSSHIP_EALIGN : CSSHIP_ENULL;
  return convmod_bwd_impl(x, dy, w, B, T, D, kernel_size, fwd_workspace, dx, grads, bn_eval, drop, workspace, workspace_bytes,
                          stream, 0, 0.0, next_drop, dx_next_bf16);
}

extern "C" int csship_convmod_bwd_after_stats(const float* x, const float* dy, const csship_conv_weights* w, int32_t B,
                                              int32_t T, int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx,
                                              float* grads, int32_t bn_eval, const csship_dropout_cfg* drop,
                                              const csship_dropout_cfg* next_drop, uint16_t* dx_next_bf16, void* workspace,
                                              size_t workspace_bytes, csship_stream_t stream) {
  if (dx_next_bf16 && !aligned8(dx_next_bf16)) return CSSHIP_EALIGN;
  return convmod_bwd_impl(x, dy, w, B, T, D, kernel_size, fwd_workspace, dx, grads, bn_eval, drop, workspace, workspace_bytes,
                          stream, 3, 0.0, next_drop, dx_next_bf16);
}

extern "C" int csship_convmod_bwd_drop(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                                       int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx,
                                       float* grads, int32_t bn_eval, const csship_dropout_cfg* drop, void* workspace,
                                       size_t workspace_bytes, csship_stream_t stream) {
  return convmod_bwd_impl(x, dy, w, B, T, D, kernel_size, fwd_workspace, dx, grads, bn_eval, drop, workspace, workspace_bytes,
                          stream, 0, 0.0);
}

extern "C" size_t csship_convmod_bwd_sums_offset(int32_t B, int32_t T, int32_t D) {
  return bwd_layout(B, T, D, D == 256 ? TT256 : TT512).sums * sizeof(float);
}

extern "C" int csship_convmod_bwd_stats(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                                        int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx, float* grads,
                                        const csship_dropout_cfg* drop, void* workspace, size_t workspace_bytes,
                                        csship_stream_t stream) {
  return convmod_bwd_impl(x, dy, w, B, T, D, kernel_size, fwd_workspace, dx, grads, 0, drop, workspace, workspace_bytes, stream,
                          1, 0.0);
}

extern "C" int csship_convmod_bwd_finish(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T,
                                         int32_t D, int32_t kernel_size, const void* fwd_workspace, float* dx, float* grads,
                                         double total_rows, const csship_dropout_cfg* drop, void* workspace,
                                         size_t workspace_bytes, csship_stream_t stream) {
  if (!(total_rows >= (double)B * T)) return CSSHIP_EBADSHAPE;
  return convmod_bwd_impl(x, dy, w, B, T, D, kernel_size, fwd_workspace, dx, grads, 0, drop, workspace, workspace_bytes, stream,
                          2, total_rows);
}

static int convmod_bwd_impl(const float* x, const float* dy, const csship_conv_weights* w, int32_t B, int32_t T, int32_t D,
                            int32_t kernel_size, const void* fwd_workspace, float* dx, float* grads, int32_t bn_eval,
                            const csship_dropout_cfg* drop, void* workspace, size_t workspace_bytes, csship_stream_t stream,
                            int phase, double total_rows, const csship_dropout_cfg* next_drop, uint16_t* dx_next_bf16) {
  const bool ndropping = next_drop && next_drop->seed && next_drop->p > 0.f;
  if (dx_next_bf16 && (int64_t)B * T * D > 0xffffffffLL) return CSSHIP_EBADSHAPE;
  const NextOut nx{ndropping ? next_drop->seed : nullptr, ndropping ? next_drop->stream : 0u, ndropping ? next_drop->p : 0.f,
                   dx_next_bf16};
  const bool dropping = drop && drop->seed && drop->p > 0.f;
  if (dropping && (int64_t)B * T * D > 0xffffffffLL) return CSSHIP_EBADSHAPE;
  const csship_dropout_cfg dc = dropping ? *drop : csship_dropout_cfg{nullptr, 0u, 0.f};
  if (!x || !dy || !w || !fwd_workspace || !dx || !grads || !workspace) return CSSHIP_ENULL;
  if (!w->ln_g || !w->ln_b || !w->pw1_w || !w->pw1_b || !w->dw_w || !w->dw_b || !w->pw2_w) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  if (kernel_size != 33 || (D != 256 && D != 512)) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(dy) || !aligned16(dx) || !aligned16(fwd_workspace) || !aligned16(workspace))
    return CSSHIP_EALIGN;
  if (workspace_bytes < csship_convmod_bwd_workspace(B, T, D)) return CSSHIP_EWORKSPACE;
  const float* fw = reinterpret_cast<const float*>(fwd_workspace);
  float* ws = reinterpret_cast<float*>(workspace);
  if (D == 256) return run_bwd<256, TT256>(x, dy, *w, B, T, fw, dx, grads, ws, bn_eval, as_stream(stream), dc, phase, total_rows, nx);
  return run_bwd<512, TT512>(x, dy, *w, B, T, fw, dx, grads, ws, bn_eval, as_stream(stream), dc, phase, total_rows, nx);
}
