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
#include "common.hpp"

namespace {

struct ConvArgs {
  const float* x;
  float* y;
  csship_conv_weights w;
  const float* g;  // FROMG: glu(LN(x)) rows [B,T,D]
  float* zbuf;     // train: pre-BN activations [B,T,D]
  float* partial;  // train: [nblk][2][D]
  int B, T;
  unsigned long long* stamps;
};

template <int D, int TT, int KS, int MODE, bool FROMG>
__global__ __launch_bounds__(2 * D) void convmod_kernel(ConvArgs a) {
  constexpr int HALO = (KS - 1) / 2, ROWS = TT + KS - 1, NTHR = 2 * D;
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
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o);
    const float mean = s * (1.f / D);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) ss += __shfl_xor(ss, o);
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
  float wk[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) wk[k] = a.w.dw_w[k * D + c];  // [k][D]: lanes read consecutive floats
  const float cb = a.w.dw_b[c];
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
  if (c < D)
    for (int i = w; i < nblk; i += 16) {
      s += (double)partial[((size_t)i * 2 + 0) * D + c];
      q += (double)partial[((size_t)i * 2 + 1) * D + c];
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
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ z,
                                                       const float* __restrict__ scale_shift,
                                                       const float* __restrict__ pw2_w, const float* __restrict__ pw2_b,
                                                       float* __restrict__ y, size_t n4, int D) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const int c = (int)((i * 4) % D);
  const float4 xv = reinterpret_cast<const float4*>(x)[i];
  const float4 zv = reinterpret_cast<const float4*>(z)[i];
  const float4 sc = *reinterpret_cast<const float4*>(scale_shift + c);
  const float4 sh = *reinterpret_cast<const float4*>(scale_shift + D + c);
  const float pw = pw2_w[0], pb = pw2_b[0];
  float4 o;
  o.x = xv.x + fmaxf(zv.x * sc.x + sh.x, 0.f) * pw + pb;
  o.y = xv.y + fmaxf(zv.y * sc.y + sh.y, 0.f) * pw + pb;
  o.z = xv.z + fmaxf(zv.z * sc.z + sh.z, 0.f) * pw + pb;
  o.w = xv.w + fmaxf(zv.w * sc.w + sh.w, 0.f) * pw + pb;
  reinterpret_cast<float4*>(y)[i] = o;
}

template <int D, int TT>
int run(const ConvArgs& a0, int bn_mode, void* workspace, hipStream_t s) {
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
  float* scale_shift = a.partial + (size_t)nblk * 2 * D;
  hipLaunchKernelGGL((convmod_kernel<D, TT, 33, 1, false>), grid, dim3(2 * D), 0, s, a);
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_reduce_kernel, dim3((D + 63) / 64), dim3(1024), 0, s, a.partial, nblk, D,
                     (double)a.B * (double)a.T, a.w.bn_g, a.w.bn_b, scale_shift);
  CSSHIP_CHECK_LAUNCH();
  hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, a.x, a.zbuf, scale_shift,
                     a.w.pw2_w, a.w.pw2_b, a.y, n / 4, D);
  CSSHIP_CHECK_LAUNCH();
  return CSSHIP_OK;
}

constexpr int TT256 = 32, TT512 = 32;

}  // namespace

extern "C" size_t csship_convmod_workspace(int32_t B, int32_t T, int32_t D) {
  const int TT = D == 256 ? TT256 : TT512;
  const size_t nblk = (size_t)((T + TT - 1) / TT) * B * 2;
  return ((size_t)B * T * D + nblk * 2 * D + 2 * (size_t)D) * sizeof(float);
}

extern "C" int csship_convmod(const float* x, float* y, const csship_conv_weights* w, int32_t B, int32_t T,
                              int32_t D, int32_t kernel_size, int32_t bn_mode, void* workspace,
                              csship_stream_t stream) {
  if (!x || !y || !w) return CSSHIP_ENULL;
  if (!w->ln_g || !w->ln_b || !w->pw1_w || !w->pw1_b || !w->dw_w || !w->dw_b || !w->bn_g || !w->bn_b ||
      !w->pw2_w || !w->pw2_b || (bn_mode == 0 && (!w->bn_mean || !w->bn_var)))
    return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  if (kernel_size != 33 || (D != 256 && D != 512) || (bn_mode != 0 && bn_mode != 1)) return CSSHIP_EUNSUPPORTED;
  if (x == y) return CSSHIP_EBADSHAPE;  // neighbouring tiles read the halo rows: not in-place safe
  if (!aligned16(x) || !aligned16(y) || !aligned16(w->ln_g) || !aligned16(w->ln_b) || (workspace && !aligned16(workspace)))
    return CSSHIP_EALIGN;
  ConvArgs a{x, y, *w, nullptr, nullptr, nullptr, B, T, g_csship_stamps};
  if (D == 256) return run<256, TT256>(a, bn_mode, workspace, as_stream(stream));
  return run<512, TT512>(a, bn_mode, workspace, as_stream(stream));
}

extern "C" int csship_conv_from_glu(const float* x, const float* g, float* y, const csship_conv_weights* w, int32_t B,
                                    int32_t T, int32_t D, int32_t kernel_size, csship_stream_t stream) {
  if (!x || !g || !y || !w) return CSSHIP_ENULL;
  if (!w->dw_w || !w->dw_b || !w->bn_g || !w->bn_b || !w->bn_mean || !w->bn_var || !w->pw2_w || !w->pw2_b) return CSSHIP_ENULL;
  if (B <= 0 || T <= 0 || B > 65535) return CSSHIP_EBADSHAPE;
  if (kernel_size != 33 || (D != 256 && D != 512)) return CSSHIP_EUNSUPPORTED;
  if (!aligned16(x) || !aligned16(g) || !aligned16(y) || g == y) return CSSHIP_EALIGN;
  ConvArgs a{x, y, *w, g, nullptr, nullptr, B, T, g_csship_stamps};
  if (D == 256) return run<256, TT256>(a, 0, nullptr, as_stream(stream));
  return run<512, TT512>(a, 0, nullptr, as_stream(stream));
}
